"""ms per step and per kernel group of the fixed-h loop for a disc of N particles (A/B of kernel choices by environment switch):
   python tests/tools/step_time.py N [steps] [tag]"""
import json, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from summersph_amd import capi, ic
n = int(sys.argv[1]); steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40; tag = sys.argv[3] if len(sys.argv) > 3 else ""
gas, sinks = ic.split_rows(ic.keplerian_disc(n, seed=214, nngb=85.0))
ctx = capi.Context(device=0)
ctx.upload(gas); ctx.set_sinks(sinks)
dt, t = ctx.run(5, 1e-2, 0.0)
ctx.synchronize()
t0 = time.perf_counter()
dt, t = ctx.run(steps, dt, t)
ctx.synchronize()
el = time.perf_counter() - t0
ctx.timing(True); ctx.timing_reset()
ctx.run(10, dt, t); ctx.synchronize()
ctx.timing(False)
kt = {k: round(ctx.timing_get(k)[0] / 10, 4) for k in capi.KERNELS}
st = ctx.stats()
print(json.dumps({"tag": tag, "n": n, "ms_per_step": round(el / steps * 1e3, 4), "Mps": round(n * steps / el / 1e6, 1),
                  "fit": [st.tile_fit_pct, st.tile_fit_pct_forces], "kernels": {k: v for k, v in kt.items() if v > 0}}))
