"""A/B of the whole-tile pair kernels (default) against the direct-gather kernels (SPH_FLAG_NO_WHOLE_TILE): evaluates the
bench disc with both, compares bitwise and prints the kernel times.
   python tests/tools/wt_probe.py [n]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from summersph_amd import capi, ic

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
gas, sinks = ic.split_rows(ic.keplerian_disc(n, seed=202, nngb=85.0))
rng = np.random.default_rng(11)
gas["vx"] = gas["vx"] + rng.normal(0.0, 0.05, n)
gas["alpha"] = np.full(n, 0.3)
capi.load()
res = {}
for name, flags in (("whole-tile", 0), ("gathers", capi.FLAG_NO_WHOLE_TILE)):
    ctx = capi.Context(device=0, flags=flags)
    ctx.upload(gas); ctx.set_sinks(sinks)
    ctx.density(); ctx.forces()
    res[name] = {f: ctx.field(f) for f in "rho P c ax ay az du dalpha".split()}
    ctx.timing(True); ctx.timing_reset()
    for _ in range(10):
        ctx.density(); ctx.forces()
    ctx.synchronize()
    d, f = ctx.timing_get("density"), ctx.timing_get("forces")
    print(f"{name:11s}: density {d[0] / d[1]:.4f} ms, forces {f[0] / f[1]:.4f} ms, tile_fit_pct {ctx.stats().tile_fit_pct}")
    ctx.close()
print("bitwise equal:", all(np.array_equal(res["whole-tile"][k], res["gathers"][k]) for k in res["gathers"]))
