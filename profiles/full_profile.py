"""Small driver for rocprofv3: a few steps of the full simulate() path (self-gravity + accretion + cull) at 1e6 particles."""
import sys
sys.path.insert(0, "/root/repo")
import torch  # noqa: F401
from summersph_amd import capi, ic
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
gas, sinks = ic.split_rows(ic.keplerian_disc(n, seed=202, nngb=85.0))
ctx = capi.Context(device=0, flags=capi.FLAG_SELF_GRAVITY | capi.FLAG_ACCRETE_CULL)
ctx.upload(gas); ctx.set_sinks(sinks)
dt, t = ctx.run(4, 1e-2, 0.0)
ctx.synchronize()
print("done", dt, t, ctx.n)
