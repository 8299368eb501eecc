"""one sph_run of K steps of simulate() as the reference runs it (Barnes-Hut self-gravity, accretion, cull) on an N-particle disc:
   python tests/tools/full_loop_steps.py N K      (profiles/hiptrace.sh counts the host synchronisations of this run)"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from summersph_amd import capi, ic
n = int(sys.argv[1]); k = int(sys.argv[2])
gas, sinks = ic.split_rows(ic.keplerian_disc(n, seed=202, nngb=85.0))
ctx = capi.Context(device=0, flags=capi.FLAG_SELF_GRAVITY | capi.FLAG_ACCRETE_CULL)
ctx.upload(gas); ctx.set_sinks(sinks)
s0 = ctx.stats().host_syncs
t0 = time.perf_counter()
dt, t = ctx.run(k, 1e-2, 0.0)
el = time.perf_counter() - t0
print(f"{k} steps, {ctx.n} particles left, {el / k * 1e3:.3f} ms/step, host synchronisations inside the build / accretion paths per step: "
      f"{(ctx.stats().host_syncs - s0) / k:.2f}")
