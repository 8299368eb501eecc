"""cProfile of the Python orchestrator at one rank: where the host spends the step (waits included)."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from summersph_amd import ic
from summersph_amd.dist import DistSim, HipBackend

gas, sinks = ic.split_rows(ic.keplerian_disc(1000000, seed=202, nngb=85.0))
sim = DistSim(HipBackend(0), gas, sinks, np.zeros(0))
d = sim.run(3, 1e-2); sim.be.synchronize()
pr = cProfile.Profile(); pr.enable()
d = sim.run(20, d); sim.be.synchronize()
pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(22)
