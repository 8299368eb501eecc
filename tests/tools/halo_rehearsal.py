"""P ranks of the native step loop (libsummersph_halo.so) as threads sharing the one GPU of a test box: wall time per step
against the same particles in one context (sph_run).  The ranks' kernels share the device, so the interesting number is
(P-rank ms/step) - (one-context ms/step of the same total particle count): what the decomposition costs.

    python tests/tools/halo_rehearsal.py [ranks=2] [particles per rank=1000000] [steps=10] [ic=disc|ring] [sph|full]

full = the loop the reference runs: Barnes-Hut self-gravity (the other ranks' sources as a locally essential tree;
SPH_HALO_REPLICATED=1: every particle of every rank), accretion and cull.
"""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from summersph_amd import capi, halo, ic            # noqa: E402
from summersph_amd.dist import slab_bounds          # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n_per = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
kind = sys.argv[4] if len(sys.argv) > 4 else "disc"
full = len(sys.argv) > 5 and sys.argv[5] == "full"
flags = (capi.FLAG_SELF_GRAVITY | capi.FLAG_ACCRETE_CULL) if full else 0
n = P * n_per
rows = ic.thin_ring(n, seed=404) if kind == "ring" else ic.keplerian_disc(n, seed=202, nngb=85.0)
gas, sinks = ic.split_rows(rows)

ctx = capi.Context(device=0, flags=flags)
ctx.upload(gas); ctx.set_sinks(sinks)
dt, t = ctx.run(3, 1e-2, 0.0)
ctx.synchronize()
t0 = time.perf_counter()
dt1, t1 = ctx.run(steps, dt, t)
ctx.synchronize()
one = (time.perf_counter() - t0) / steps * 1e3
ref = {f: ctx.field(f) for f in ("x", "u")}
n_left = ctx.n
ctx.close()

hub = halo.Hub(P)
bounds = slab_bounds(gas["x"], P)
owner = np.searchsorted(bounds, gas["x"], side="right")
res, bar = [None] * P, threading.Barrier(P)


def worker(rank):
    c = capi.Context(device=0, flags=flags)
    h = halo.Halo.inproc(c, hub, rank, P)
    sel = owner == rank
    mine = {k: v[sel] for k, v in gas.items()}
    mine["gid"] = np.nonzero(sel)[0]
    c.set_sinks(sinks)
    h.set_slabs(bounds, 32)
    h.upload(mine)
    d, tt = h.run(3, 1e-2, 0.0)
    c.synchronize(); bar.wait()
    s0 = h.stats()
    w0 = (s0.host_waits, s0.exchanges, s0.collectives, s0.let_sent, s0.let_received, s0.let_updates)
    a = time.perf_counter()
    d, tt = h.run(steps, d, tt)
    c.synchronize(); bar.wait()
    el = time.perf_counter() - a
    s1 = h.stats()
    res[rank] = dict(ms=el / steps * 1e3, dt=d, t=tt, state=h.download(), ghosts=s1.ghosts,
                     waits=(s1.host_waits - w0[0]) / steps, p2p=(s1.exchanges - w0[1]) / steps, coll=(s1.collectives - w0[2]) / steps,
                     let=((s1.let_sent - w0[3]) / max(s1.let_updates - w0[5], 1), (s1.let_received - w0[4]) / max(s1.let_updates - w0[5], 1)),
                     owned=h.n_owned)
    h.close(); c.close()


th = [threading.Thread(target=worker, args=(r,)) for r in range(P)]
[x.start() for x in th]
[x.join() for x in th]
order = np.argsort(np.concatenate([r["state"]["gid"] for r in res]))
if full:
    assert sum(r["owned"] for r in res) == n_left, (sum(r["owned"] for r in res), n_left)
    mode = "replicated sources" if os.environ.get("SPH_HALO_REPLICATED") else "locally essential tree"
    print(f"self-gravity sources ({mode}): records sent / received per rank and source update", [tuple(int(v) for v in r["let"]) for r in res],
          "of", [n_left - r["owned"] for r in res], "particles held by the other ranks")
err = {f: float(np.max(np.abs(np.concatenate([r["state"][f] for r in res])[order] - ref[f])) / np.max(np.abs(ref[f]))) for f in ref}
print(f"{kind} {n} particles: one context {one:.3f} ms/step; {P} ranks (threads, one GPU) {max(r['ms'] for r in res):.3f} ms/step; "
      f"difference {max(r['ms'] for r in res) - one:.3f} ms")
print("per rank and step: host waits", [r["waits"] for r in res], "p2p rounds", [r["p2p"] for r in res], "all-gathers", [r["coll"] for r in res],
      "ghosts", [r["ghosts"] for r in res])
print("dt equal:", all(r["dt"] == dt1 for r in res), "max rel deviation from the one-context run:", err)
