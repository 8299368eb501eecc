import os, sys, time, numpy as np
sys.path.insert(0, "/root/repo")
import torch
from summersph_amd import capi, ic
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
rows = ic.keplerian_disc(n, seed=202, nngb=85.0)
gas, sinks = ic.split_rows(rows)
ctx = capi.Context(device=0, flags=capi.FLAG_SELF_GRAVITY)
ctx.upload(gas); ctx.set_sinks(sinks)
ctx.density(); ctx.forces()
a = ctx.field("ax").copy()
ctx.timing(True); ctx.timing_reset()
for _ in range(3):
    ctx.kick(0.0); ctx.density(); ctx.forces()
ctx.synchronize()
ms, cnt = ctx.timing_get("gravity")
print("gravity ms/launch", ms / cnt, "launches", cnt)
np.save(os.environ.get("OUT", "/tmp/ax.npy"), a)
