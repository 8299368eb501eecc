"""variable h over many steps with and without the re-flag pass: the neighbour sets must stay those of a build all the way
(state equal to rounding, identical dt decisions).   python tests/tools/reflag_long_run.py [n=100000] [steps=200]"""
import os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from summersph_amd import capi, ic
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
gas, sinks = ic.split_rows(ic.keplerian_disc_var(n, seed=71))
out = {}
for tag, flags in (("build", capi.FLAG_NO_REFLAG), ("reflag", 0)):
    ctx = capi.Context(device=0, variable=True, flags=capi.FLAG_VARIABLE_H | flags)
    ctx.upload(gas); ctx.set_sinks(sinks)
    dts, t = [1e-2], 0.0
    for _ in range(steps):
        dt, t = ctx.run(1, dts[-1], t); dts.append(dt)
    st = ctx.stats()
    out[tag] = dict(dts=dts, t=t, builds=st.nlist_builds, reflags=st.nlist_reflags, **{f: ctx.field(f) for f in "x vx u h rho alpha".split()})
    ctx.close()
print("steps", steps, "builds/reflags:", out["build"]["builds"], out["build"]["reflags"], "|", out["reflag"]["builds"], out["reflag"]["reflags"])
print("identical dt sequence:", out["build"]["dts"] == out["reflag"]["dts"], "t", out["build"]["t"], out["reflag"]["t"])
for f in "x vx u h rho alpha".split():
    a, b = out["build"][f], out["reflag"][f]
    print(f, "max rel deviation", float(np.max(np.abs(a - b)) / np.max(np.abs(a))))
