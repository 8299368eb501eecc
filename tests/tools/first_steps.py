"""ms of each of the first K steps after an upload, twice on the same context (the second time after a re-upload): what the
first pass over a trajectory pays that a repeat does not.   python tests/tools/first_steps.py [N] [K]"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from summersph_amd import capi, ic
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 30
gas, sinks = ic.split_rows(ic.keplerian_disc(n, seed=214, nngb=85.0))
ctx = capi.Context(device=0)
for rep in range(2):
    ctx.upload(gas); ctx.set_sinks(sinks)
    dt, t, ms = 1e-2, 0.0, []
    for k in range(K):
        ctx.synchronize(); t0 = time.perf_counter()
        dt, t = ctx.run(1, dt, t)
        ctx.synchronize(); ms.append(round((time.perf_counter() - t0) * 1e3, 3))
    st = ctx.stats()
    print("pass", rep, "cap", st.nlist_capacity, "max", st.nlist_max, "syncs", st.host_syncs, ms, flush=True)
