import sys, os, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from summersph_amd import capi, ic
n = 200000
gas, sinks = ic.split_rows(ic.keplerian_disc(n, seed=212, nngb=float(sys.argv[1]) if len(sys.argv) > 1 else 340.0))
rng = np.random.default_rng(11); gas["vx"] = gas["vx"] + rng.normal(0, 0.05, n); gas["alpha"] = np.full(n, 0.3)
res = {}
for name, flags in (("wt", 0), ("gathers", capi.FLAG_NO_WHOLE_TILE)):
    ctx = capi.Context(device=0, flags=flags); ctx.upload(gas); ctx.set_sinks(sinks)
    ctx.density(); ctx.forces()
    res[name] = {f: ctx.field(f) for f in "rho ax du dalpha".split()}
    ctx.timing(True); ctx.timing_reset()
    for _ in range(10): ctx.density(); ctx.forces()
    ctx.synchronize()
    d, f = ctx.timing_get("density"), ctx.timing_get("forces"); st = ctx.stats()
    print(f"{name}: density {d[0]/d[1]:.4f} forces {f[0]/f[1]:.4f} ms fit d {st.tile_fit_pct} f {st.tile_fit_pct_forces} mean {st.nlist_mean:.1f} lane_eff {st.lane_efficiency_forces:.3f}")
    ctx.close()
for k in res["wt"]:
    a, b = res["wt"][k], res["gathers"][k]
    print(k, float(np.max(np.abs(a - b)) / np.max(np.abs(b))))
