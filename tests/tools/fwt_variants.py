"""A/B of the forces_wt variants (tile record length, workgroup size, dw table in LDS or recomputed; tiled.hip
FWT_VARIANTS) on the bench disc: one process per variant (SPH_FWT_VARIANT is read once), kernel times from the
library's own HIP events, results compared bitwise with the direct-gather kernels.
   python tests/tools/fwt_variants.py [n] [variants, e.g. 0,1,2]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CHILD = r'''
import sys, json, numpy as np
sys.path.insert(0, %r)
from summersph_amd import capi, ic
n = int(sys.argv[1])
gas, sinks = ic.split_rows(ic.keplerian_disc(n, seed=202, nngb=85.0))
rng = np.random.default_rng(11)
gas["vx"] = gas["vx"] + rng.normal(0.0, 0.05, n)
gas["alpha"] = np.full(n, 0.3)
capi.load()
res = {}
out = {}
for name, flags in (("wt", 0), ("gathers", capi.FLAG_NO_WHOLE_TILE)):
    ctx = capi.Context(device=0, flags=flags)
    ctx.upload(gas); ctx.set_sinks(sinks)
    ctx.density(); ctx.forces()
    res[name] = {f: ctx.field(f) for f in "rho P c ax ay az du dalpha".split()}
    ctx.timing(True); ctx.timing_reset()
    for _ in range(20):
        ctx.density(); ctx.forces()
    ctx.synchronize()
    d, f = ctx.timing_get("density"), ctx.timing_get("forces")
    st = ctx.stats()
    out[name] = {"density_ms": d[0] / d[1], "forces_ms": f[0] / f[1], "fit_d": st.tile_fit_pct, "fit_f": st.tile_fit_pct_forces}
    ctx.close()
out["bitwise"] = all(np.array_equal(res["wt"][k], res["gathers"][k]) for k in res["gathers"])
print("RESULT " + json.dumps(out))
''' % ROOT

n = sys.argv[1] if len(sys.argv) > 1 else "1000000"
variants = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else list(range(8))
for v in variants:
    r = subprocess.run([sys.executable, "-c", CHILD, n], capture_output=True, text=True, env={**os.environ, "SPH_FWT_VARIANT": str(v)}, timeout=600)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    if r.returncode != 0 or not line:
        print(f"variant {v}: FAILED rc={r.returncode}\n{r.stdout[-2000:]}\n{r.stderr[-2000:]}", flush=True)
        continue
    o = json.loads(line[0][7:])
    print(f"variant {v}: forces_wt {o['wt']['forces_ms']:.4f} ms (fit {o['wt']['fit_f']}%), density_wt {o['wt']['density_ms']:.4f} ms (fit {o['wt']['fit_d']}%), "
          f"gathers: forces {o['gathers']['forces_ms']:.4f} density {o['gathers']['density_ms']:.4f}; bitwise equal: {o['bitwise']}", flush=True)
