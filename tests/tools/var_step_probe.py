"""wall time per step of the variable-h path, kernel timers off (A/B of SPH_NO_H_REFRESH)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from summersph_amd import capi, ic
gas, sinks = ic.split_rows(ic.keplerian_disc_var(1000000, seed=303))
ctx = capi.Context(device=0, variable=True)
ctx.upload(gas); ctx.set_sinks(sinks)
dt, t = ctx.run(3, 1e-2, 0.0); ctx.synchronize()
for k in range(3):
    t0 = time.perf_counter(); dt, t = ctx.run(4, dt, t); ctx.synchronize(); t1 = time.perf_counter()
    print("ms/step", (t1 - t0) / 4 * 1e3, "builds", ctx.stats().grid_builds, ctx.stats().nlist_builds, flush=True)
