"""Cost of the orchestrators with ONE rank (no messages): DistSim (Python, torch.distributed) and the native step loop of
libsummersph_halo.so against sph_run on the same workload."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch
from summersph_amd import capi, ic
from summersph_amd.dist import DistSim, HipBackend

n = 1000000
gas, sinks = ic.split_rows(ic.keplerian_disc(n, seed=202, nngb=85.0))
ctx = capi.Context(device=0); ctx.upload(gas); ctx.set_sinks(sinks)
dt, t = ctx.run(2, 1e-2, 0.0); ctx.synchronize()
t0 = time.perf_counter(); dt, t = ctx.run(30, dt, t); ctx.synchronize(); t1 = time.perf_counter()
print("sph_run       ms/step", (t1 - t0) / 30 * 1e3)
ctx.close()
sim = DistSim(HipBackend(0), gas, sinks, np.zeros(0))
d = sim.run(2, 1e-2); sim.be.synchronize()
t0 = time.perf_counter(); d = sim.run(30, d); sim.be.synchronize(); torch.cuda.synchronize(); t1 = time.perf_counter()
print("DistSim (P=1) ms/step", (t1 - t0) / 30 * 1e3, "final dt", d, dt)
sim.profile = True
d = sim.run(10, d); sim.be.synchronize()
print("phases, ms/step (synchronised at phase boundaries):", {k: round(v / 10 * 1e3, 3) for k, v in sim.phase_s.items()})

from summersph_amd import halo
c2 = capi.Context(device=0); c2.set_sinks(sinks)
h = halo.Halo.inproc(c2, halo.Hub(1), 0, 1)
h.set_slabs(np.zeros(0), 32); h.upload(gas)
d2, t2 = h.run(2, 1e-2, 0.0); c2.synchronize()
t0 = time.perf_counter(); d2, t2 = h.run(30, d2, t2); c2.synchronize(); t1 = time.perf_counter()
print("native halo (P=1) ms/step", (t1 - t0) / 30 * 1e3, "final dt", d2)
