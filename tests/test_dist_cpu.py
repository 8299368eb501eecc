"""Multi-rank orchestration (summersph_amd/dist.py) on CPUs: world_size 2 and 3 under gloo, with the
oracle-backed backend.  The decomposed run must reproduce the single-domain run (and the real
reference's trajectory fixture) particle by particle -- decomposition invariance, SURVEY.md 8(e)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT, load_golden, rel_err

FIELDS = "x y z vx vy vz u alpha".split()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, nsteps, outdir, ic_rows, migrate_every, chunk, gravity=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from dist_backend import OracleBackend
    from summersph_amd import ic
    from summersph_amd.dist import DistSim, slab_bounds
    torch.set_num_threads(1)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    gas, sinks = ic.split_rows(ic_rows)
    bounds = slab_bounds(gas["x"], world)
    owner = np.searchsorted(bounds, gas["x"], side="right")
    sel = owner == rank
    mine = {k: v[sel] for k, v in gas.items()}
    mine["gid"] = np.nonzero(sel)[0]
    sim = DistSim(OracleBackend(gravity=gravity), mine, sinks, bounds, migrate_every=migrate_every)
    dts = [1e-2]
    for _ in range(0, nsteps, chunk):       # chunk > 1: the dt reduction rides on the next step's exchange
        dts.append(sim.run(chunk, dts[-1]))
    st = sim.gather_state()
    s = sim.be.get_sinks()
    np.savez(os.path.join(outdir, f"r{rank}.npz"), dts=np.array(dts), ghosts=sim.stats["ghosts"],
             migrated=sim.stats["migrated"], sx=s["x"], svx=s["vx"], **st)
    dist.barrier()
    dist.destroy_process_group()


def _run(world, nsteps, rows, tmp_path, migrate_every=8, chunk=1, gravity=False):
    mp.spawn(_worker, args=(world, _free_port(), nsteps, str(tmp_path), rows, migrate_every, chunk, gravity), nprocs=world, join=True)
    parts = [dict(np.load(tmp_path / f"r{r}.npz")) for r in range(world)]
    gid = np.concatenate([p["gid"] for p in parts])
    assert np.array_equal(np.sort(gid), np.arange(gid.size))          # every particle owned exactly once
    order = np.argsort(gid)
    merged = {k: np.concatenate([p[k] for p in parts])[order] for k in FIELDS}
    return parts, merged


@pytest.mark.parametrize("world,chunk", [(2, 1), (3, 1), (2, 5)])
def test_decomposition_invariance_vs_reference_fixture(tmp_path, world, chunk):
    """chunk 1: one run() per step (dt reduced at the end of every step); chunk 5: one run(5), the dt candidate of
    each step travels with the first reduction of the next one"""
    g = load_golden("disc3000_traj")
    parts, merged = _run(world, 5, g["ic"], tmp_path, chunk=chunk)
    for p in parts:
        assert list(p["dts"]) == list(g["sph_dt_seq"])[::chunk]       # same dt decisions on every rank
        assert p["ghosts"] > 0
        assert np.max(np.abs(p["sx"] - g["sph_s5_sx"])) <= 1e-12      # replicated sink stays in sync
        assert np.array_equal(p["sx"], parts[0]["sx"]) and np.array_equal(p["svx"], parts[0]["svx"])
    for f in FIELDS:
        assert rel_err(merged[f], g["sph_s5_" + f]) <= 1e-11, f        # vs the REAL reference trajectory


def test_migration_happens_and_is_lossless(tmp_path):
    from summersph_amd import ic
    rows = ic.keplerian_disc(2500, seed=77, r_in=8.0)
    rows[:-1, 3:6] *= 3.0          # fast particles: many cross the slab edges within a few steps
    parts, merged = _run(3, 6, rows, tmp_path, migrate_every=2)
    assert sum(int(p["migrated"]) for p in parts) > 0
    # single-domain oracle run of the same IC
    from oracle import orc
    gas, sinks = ic.split_rows(rows)
    o = orc.Oracle(gas, sinks)
    dts = [1e-2]
    for _ in range(6):
        dts.append(o.step(dts[-1]))
    assert list(parts[0]["dts"]) == dts
    for f in FIELDS:
        assert rel_err(merged[f], getattr(o, f)) <= 1e-11, f


def test_self_gravity_with_replicated_tree_vs_reference_fixture(tmp_path):
    """find_forces with the Barnes-Hut term on 2 ranks: every rank builds the tree of ALL particles (all-gathered
    sources) and walks it for its own; the result is the reference's full trajectory (no accretion happens in it)"""
    g = load_golden("disc3000_traj")
    parts, merged = _run(2, 5, g["ic"], tmp_path, gravity=True, migrate_every=2)
    for p in parts:
        assert list(p["dts"]) == list(g["full_dt_seq"])
    for f in FIELDS:
        assert rel_err(merged[f], g["full_s5_" + f]) <= 1e-11, f


def _worker_acc(rank, world, port, nsteps, outdir, ic_rows):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from dist_backend import OracleBackend
    from summersph_amd import ic
    from summersph_amd.dist import DistSim, slab_bounds
    torch.set_num_threads(1)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    gas, sinks = ic.split_rows(ic_rows)
    bounds = slab_bounds(gas["x"], world)
    sel = np.searchsorted(bounds, gas["x"], side="right") == rank
    mine = {k: v[sel] for k, v in gas.items()}
    mine["gid"] = np.nonzero(sel)[0]
    sim = DistSim(OracleBackend(gravity=True, accrete=True), mine, sinks, bounds, migrate_every=2)
    dts, ns = [1e-2], []
    for _ in range(nsteps):
        dts.append(sim.step(dts[-1]))
        ns.append(sim.n_owned)
    st = sim.gather_state()
    s = sim.be.get_sinks()
    np.savez(os.path.join(outdir, f"r{rank}.npz"), dts=np.array(dts), ns=np.array(ns), sm=s["m"], sx=s["x"], svx=s["vx"], **st)
    dist.barrier()
    dist.destroy_process_group()


def test_accretion_and_cull_across_ranks_vs_reference_fixture(tmp_path):
    """simulate() as it is on 2 ranks: self-gravity with the shared tree, accretion decided on it by every rank for its
    own particles, sink sums all-gathered -- the particle set shrinks exactly as in the real reference"""
    g = load_golden("acc2000_traj")
    mp.spawn(_worker_acc, args=(2, _free_port(), 3, str(tmp_path), g["ic"]), nprocs=2, join=True)
    parts = [dict(np.load(tmp_path / f"r{r}.npz")) for r in range(2)]
    ns = parts[0]["ns"] + parts[1]["ns"]
    assert list(ns) == [int(v) for v in g["full_n_seq"][1:]] and ns[0] == 1996
    gid = np.concatenate([p["gid"] for p in parts])
    order = np.argsort(gid)                    # survivors keep their relative order in the reference (pack)
    assert np.unique(gid).size == gid.size == 1996
    for p in parts:
        assert list(p["dts"]) == list(g["full_dt_seq"])
        assert abs(p["sm"][0] - g["full_s3_sm"][0]) <= 1e-15 and abs(p["sx"][0] - g["full_s3_sx"][0]) <= 1e-12
        assert np.array_equal(p["sm"], parts[0]["sm"]) and np.array_equal(p["svx"], parts[0]["svx"])
    for f in FIELDS:
        assert rel_err(np.concatenate([p[f] for p in parts])[order], g["full_s3_" + f]) <= 1e-11, f


def test_two_sinks_across_ranks(tmp_path):
    """a binary: the sink-sink terms are added once (rank 0), the gas terms summed over ranks"""
    g = load_golden("bin2000_traj")
    parts, merged = _run(3, 3, g["ic"], tmp_path, migrate_every=2)
    for p in parts:
        assert list(p["dts"]) == list(g["sph_dt_seq"])
        assert np.max(np.abs(p["sx"] - g["sph_s3_sx"])) <= 1e-12 and np.array_equal(p["svx"], parts[0]["svx"])
    for f in FIELDS:
        assert rel_err(merged[f], g["sph_s3_" + f]) <= 1e-11, f


def _worker_var(rank, world, port, nsteps, outdir, ic_rows, params):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from dist_backend import OracleVBackend
    from summersph_amd import ic
    from summersph_amd.dist import DistSim, slab_bounds
    torch.set_num_threads(1)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    gas, sinks = ic.split_rows(ic_rows)
    bounds = slab_bounds(gas["x"], world)
    sel = np.searchsorted(bounds, gas["x"], side="right") == rank
    mine = {k: v[sel] for k, v in gas.items()}
    mine["gid"] = np.nonzero(sel)[0]
    gamma, eta, tol, maxlen, scale = (float(v) for v in params)
    sim = DistSim(OracleVBackend(gamma, eta, tol, maxlen, scale), mine, sinks, bounds, migrate_every=2)
    dts = [1e-2]
    for _ in range(nsteps):
        dts.append(sim.step(dts[-1]))
    st = sim.gather_state()
    np.savez(os.path.join(outdir, f"r{rank}.npz"), dts=np.array(dts), ghosts=sim.stats["ghosts"], h=sim.owned[9].cpu().numpy(), **st)
    dist.barrier()
    dist.destroy_process_group()


def test_variable_h_across_ranks_vs_reference_fixture(tmp_path):
    """the variable-h orchestration (h and global particle numbers in the ghost exchange, leaf cells from the octree of all
    particles, rho + Omega refresh, calc_smoothing per rank) on 2 CPU ranks against the real reference's trajectory"""
    g = load_golden("discv3000_traj")
    mp.spawn(_worker_var, args=(2, _free_port(), 5, str(tmp_path), g["ic"], g["params"]), nprocs=2, join=True)
    parts = [dict(np.load(tmp_path / f"r{r}.npz")) for r in range(2)]
    gid = np.concatenate([p["gid"] for p in parts])
    assert np.array_equal(np.sort(gid), np.arange(gid.size))
    order = np.argsort(gid)
    for p in parts:
        assert list(p["dts"]) == list(g["sph_dt_seq"]) and p["ghosts"] > 0
    for f in FIELDS + ["h"]:
        assert rel_err(np.concatenate([p[f] for p in parts])[order], g["sph_s5_" + f]) <= 1e-10, f
