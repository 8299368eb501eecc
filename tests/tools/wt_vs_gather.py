"""The whole-tile kernels (density_wt, forces_q) against the direct-gather kernels (SPH_FLAG_NO_WHOLE_TILE) on the bench
disc: kernel times from the library's own HIP events, and the differences field by field (density bitwise, forces at
summation-order level).
   python tests/tools/wt_vs_gather.py [n] [launches]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from summersph_amd import capi, ic

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
gas, sinks = ic.split_rows(ic.keplerian_disc(n, seed=202, nngb=85.0))
rng = np.random.default_rng(11)
gas["vx"] = gas["vx"] + rng.normal(0.0, 0.05, n)
gas["alpha"] = np.full(n, 0.3)
capi.load()
res = {}
for name, flags in (("wt", 0), ("gathers", capi.FLAG_NO_WHOLE_TILE)):
    ctx = capi.Context(device=0, flags=flags)
    ctx.upload(gas); ctx.set_sinks(sinks)
    ctx.density(); ctx.forces()
    res[name] = {f: ctx.field(f) for f in "rho P c ax ay az du dalpha".split()}
    ctx.timing(True); ctx.timing_reset()
    for _ in range(reps):
        ctx.density(); ctx.forces()
    ctx.synchronize()
    d, f = ctx.timing_get("density"), ctx.timing_get("forces")
    st = ctx.stats()
    print(f"{name}: density {d[0] / d[1]:.4f} ms forces {f[0] / f[1]:.4f} ms fit d {st.tile_fit_pct} f {st.tile_fit_pct_forces}")
    ctx.close()
for k in res["wt"]:
    a, b = res["wt"][k], res["gathers"][k]
    nd = int(np.sum(a != b))
    print(f"{k}: differing {nd} of {a.size}, max rel {np.max(np.abs(a - b)) / np.max(np.abs(b)):.2e}")
