"""Sanity run: 300 steps of the reference's whole loop body (self-gravity, accretion, cull) on a 100k disc; prints the
particle count, dt, energy and list statistics every 50 steps."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from summersph_amd import capi, ic

gas, sinks = ic.split_rows(ic.keplerian_disc(100000, seed=5, r_in=3.0))
ctx = capi.Context(device=0, flags=capi.FLAG_SELF_GRAVITY | capi.FLAG_ACCRETE_CULL)
ctx.upload(gas); ctx.set_sinks(sinks)
dt, t = 1e-2, 0.0
for k in range(6):
    dt, t = ctx.run(50, dt, t)
    m = ctx.field("m")
    e = float(np.sum(m * (0.5 * (ctx.field("vx") ** 2 + ctx.field("vy") ** 2 + ctx.field("vz") ** 2) + ctx.field("u"))))
    st = ctx.stats()
    s = ctx.get_sinks()
    ok = all(np.all(np.isfinite(ctx.field(f))) for f in "x vx u alpha".split())
    print(f"step {50 * (k + 1):4d} n={ctx.n} dt={dt:.5f} t={t:.3f} E_kin+th={e:.6e} sink m={s['m'][0]:.8f} "
          f"list max={st.nlist_max} cap={st.nlist_capacity} tile_fit={st.tile_fit_pct}% finite={ok}", flush=True)
