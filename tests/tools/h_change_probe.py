"""how much does h change per step in the variable-h bench workload?  (sizes the margin of a re-flag pass)"""
import sys, os, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from summersph_amd import capi, ic
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
ctx = bench.make_single_ctx(capi, ic, torch, True, n, 85.0, 0, 0)
dt, t = 1e-2, 0.0
h0 = ctx.field("h")
for s in range(12):
    dt, t = ctx.run(1, dt, t)
    h1 = ctx.field("h")
    # the order of ctx.field is the caller's order (orig), so rows compare particle by particle
    rel = h1 / h0
    print(s, "dt %.4g" % dt, "max grow %.5f max shrink %.5f  p99.9 |rel-1| %.2e  mean |rel-1| %.2e" % (rel.max(), rel.min(), np.quantile(np.abs(rel - 1), 0.999), np.abs(rel - 1).mean()), flush=True)
    h0 = h1
