"""many steps of the native multi-GPU loop (ranks = threads on one GPU) against one context: protocol soak test
(capacity changes, migrations, lists regrowing).   python tests/tools/halo_long_run.py [ranks=2] [n=100000] [steps=400]"""
import os, sys, threading, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from summersph_amd import capi, halo, ic
from summersph_amd.dist import slab_bounds
P = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 400
gas, sinks = ic.split_rows(ic.keplerian_disc(n, seed=61, nngb=85.0))
rng = np.random.default_rng(3); gas["vx"] = gas["vx"] + rng.normal(0, 0.03, n); gas["alpha"] = np.full(n, 0.2)
ctx = capi.Context(device=0); ctx.upload(gas); ctx.set_sinks(sinks)
dt1, t1 = ctx.run(steps, 1e-2, 0.0)
ref = {f: ctx.field(f) for f in ("x", "y", "vx", "u", "alpha")}; ctx.close()
hub = halo.Hub(P); bounds = slab_bounds(gas["x"], P); owner = np.searchsorted(bounds, gas["x"], side="right")
out, errs = [None] * P, []
def worker(rank):
    try:
        c = capi.Context(device=0); h = halo.Halo.inproc(c, hub, rank, P)
        sel = owner == rank; mine = {k: v[sel] for k, v in gas.items()}; mine["gid"] = np.nonzero(sel)[0]
        c.set_sinks(sinks); h.set_slabs(bounds, 8); h.upload(mine)
        d, t = h.run(steps, 1e-2, 0.0)
        s = h.stats()
        out[rank] = dict(dt=d, t=t, st=h.download(), mig=s.migrated, ex=s.exchanges, waits=s.host_waits, n=h.n_owned)
        h.close(); c.close()
    except Exception as e:
        errs.append((rank, repr(e)))
th = [threading.Thread(target=worker, args=(r,)) for r in range(P)]
[x.start() for x in th]; [x.join() for x in th]
print("errors:", errs)
order = np.argsort(np.concatenate([o["st"]["gid"] for o in out]))
print("steps", steps, "dt", dt1, [o["dt"] for o in out], "t", t1, [o["t"] for o in out])
print("owned", [o["n"] for o in out], "migrated", out[0]["mig"], "p2p rounds", out[0]["ex"], "host waits", out[0]["waits"])
for f in ref:
    m = np.concatenate([o["st"][f] for o in out])[order]
    print(f, "max rel deviation", float(np.max(np.abs(m - ref[f])) / np.max(np.abs(ref[f]))))
