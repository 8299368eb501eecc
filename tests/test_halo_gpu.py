"""The native multi-GPU step loop (libsummersph_halo.so, csrc/halo.hip) on the one GPU of the test box.

* in-process transport: 2 and 3 ranks as threads of this process, each with its own context and its own pair of streams;
  the merged result must match the real reference's trajectory like the single-context run and dist.py's ranks do;
* RCCL transport with ONE rank: communicator set-up, ncclAllGather and grouped ncclSend/ncclRecv (to itself) really run;
  more ranks need more GPUs than this pool has (RCCL refuses two ranks on one device).
"""
import threading

import numpy as np
import pytest

from conftest import load_golden, rel_err
from summersph_amd import capi, halo, ic
from summersph_amd.dist import slab_bounds

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(300)]
FIELDS = "x y z vx vy vz u alpha".split()


def _run_ranks(world, gas, sinks, nsteps, migrate_every=2, collect_root=False):
    hub = halo.Hub(world)
    bounds = slab_bounds(gas["x"], world)
    owner = np.searchsorted(bounds, gas["x"], side="right")
    out, errs = [None] * world, []

    def worker(rank):
        try:
            ctx = capi.Context(device=0)
            h = halo.Halo.inproc(ctx, hub, rank, world)
            sel = owner == rank
            mine = {k: v[sel] for k, v in gas.items()}
            mine["gid"] = np.nonzero(sel)[0]
            ctx.set_sinks(sinks)
            h.set_slabs(bounds, migrate_every)
            h.upload(mine)
            dts, t = [1e-2], 0.0
            for _ in range(nsteps):
                dt, t = h.run(1, dts[-1], t)
                dts.append(dt)
            res = {"dts": dts, "t": t, "state": h.download(), "stats": h.stats(), "sinks": ctx.get_sinks()}
            if collect_root:
                res["root"] = h.gather_root(0, gas["x"].size)
            out[rank] = res
            h.close(); ctx.close()
        except Exception as e:      # noqa: BLE001 -- reported by the test below
            errs.append((rank, repr(e)))

    th = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    for t_ in th:
        t_.start()
    for t_ in th:
        t_.join()
    hub.close()
    assert not errs, errs
    return out


@pytest.mark.parametrize("world", [2, 3])
def test_inproc_ranks_match_reference_fixture(world):
    g = load_golden("disc3000_traj")
    gas, sinks = ic.split_rows(g["ic"])
    parts = _run_ranks(world, gas, sinks, 5, collect_root=True)
    gid = np.concatenate([p["state"]["gid"] for p in parts])
    assert np.array_equal(np.sort(gid), np.arange(gid.size))
    order = np.argsort(gid)
    for p in parts:
        assert list(p["dts"]) == list(g["sph_dt_seq"])
        assert p["stats"].ghosts > 0 and p["stats"].exchanges > 0
        assert np.array_equal(p["sinks"]["x"], parts[0]["sinks"]["x"])
    assert sum(p["stats"].migrations for p in parts) > 0
    for f in FIELDS:
        merged = np.concatenate([p["state"][f] for p in parts])[order]
        assert rel_err(merged, g["sph_s5_" + f]) <= 1e-11, f
    # the collective save: everything on rank 0, in global-number order
    root = parts[0]["root"]
    assert parts[1]["root"] is None
    assert np.array_equal(root["gid"], np.arange(gid.size))
    for f in FIELDS:
        assert np.array_equal(root[f], np.concatenate([p["state"][f] for p in parts])[order]), f


def test_inproc_single_rank_is_sph_run():
    """one rank: no ghosts, no messages -- the loop must give what sph_run gives, bit for bit"""
    g = load_golden("disc3000_traj")
    gas, sinks = ic.split_rows(g["ic"])
    one = _run_ranks(1, gas, sinks, 5)[0]
    ctx = capi.Context(device=0)
    ctx.upload(gas); ctx.set_sinks(sinks)
    dt, t = 1e-2, 0.0
    for _ in range(5):
        dt, t = ctx.run(1, dt, t)
    ref = {f: ctx.field(f) for f in FIELDS}
    assert one["dts"][-1] == dt and one["t"] == t
    order = np.argsort(one["state"]["gid"])
    for f in FIELDS:
        assert np.array_equal(one["state"][f][order], ref[f]), f
    ctx.close()


def test_rccl_transport_single_rank():
    g = load_golden("disc3000_traj")
    gas, sinks = ic.split_rows(g["ic"])
    ctx = capi.Context(device=0)
    h = halo.Halo.rccl(ctx, halo.unique_id(), 0, 1)
    h.selftest(100000)
    assert h.stats().exchanges == 1 and h.stats().collectives == 1
    ctx.set_sinks(sinks)
    h.set_slabs(np.zeros(0), 2)
    gas = dict(gas); gas["gid"] = np.arange(gas["x"].size)
    h.upload(gas)
    dts, t = [1e-2], 0.0
    for _ in range(5):
        dt, t = h.run(1, dts[-1], t)
        dts.append(dt)
    assert dts == list(g["sph_dt_seq"])
    st = h.download()
    for f in FIELDS:
        assert rel_err(st[f], g["sph_s5_" + f]) <= 1e-12, f
    whole = h.gather_root(0, gas["x"].size)
    assert np.array_equal(whole["x"], st["x"])
    h.close(); ctx.close()


def test_attach_to_a_callers_communicator_and_stream():
    """sph_halo_attach: the ncclComm_t and the second hipStream_t belong to the caller (here: made with RCCL's and HIP's own
    C entry points through ctypes, one rank); the loop must not destroy either"""
    import ctypes as C
    import torch
    g = load_golden("disc3000_traj")
    gas, sinks = ic.split_rows(g["ic"])
    ctx = capi.Context(device=0)
    halo.load()
    rccl = C.CDLL("librccl.so.1")           # the copy already in the process (halo.load)

    class Uid(C.Structure):
        _fields_ = [("b", C.c_char * 128)]

    uid = Uid()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, Uid, C.c_int]
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    stream = torch.cuda.Stream(device=0)
    h = halo.Halo.attach(ctx, comm.value, stream.cuda_stream, 0, 1)
    h.selftest(50000)
    ctx.set_sinks(sinks)
    h.set_slabs(np.zeros(0), 0)
    h.upload(gas)
    dts, t = [1e-2], 0.0
    for _ in range(5):
        dt, t = h.run(1, dts[-1], t)
        dts.append(dt)
    assert dts == list(g["sph_dt_seq"])
    st = h.download()
    for f in FIELDS:
        assert rel_err(st[f], g["sph_s5_" + f]) <= 1e-12, f
    h.close()
    # both are still the caller's: usable after the halo object is gone
    x = torch.ones(8, device="cuda:0")
    with torch.cuda.stream(stream):
        y = (x * 2).sum()
    stream.synchronize()
    assert float(y) == 16.0
    count = C.c_int(0)
    rccl.ncclCommCount.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    assert rccl.ncclCommCount(comm, C.byref(count)) == 0 and count.value == 1
    rccl.ncclCommDestroy.argtypes = [C.c_void_p]
    assert rccl.ncclCommDestroy(comm) == 0
    ctx.close()


def test_ring_two_ranks_viscosity():
    """the viscous, asymmetric workload (BASELINE configs[1] shape): 2 ranks against one context, 4 steps"""
    rows = ic.thin_ring(20000, seed=17)
    gas, sinks = ic.split_rows(rows)
    parts = _run_ranks(2, gas, sinks, 4, migrate_every=2)
    ctx = capi.Context(device=0)
    ctx.upload(gas); ctx.set_sinks(sinks)
    dt, t = 1e-2, 0.0
    for _ in range(4):
        dt, t = ctx.run(1, dt, t)
    order = np.argsort(np.concatenate([p["state"]["gid"] for p in parts]))
    assert parts[0]["dts"][-1] == dt and parts[1]["dts"][-1] == dt
    for f in FIELDS:
        merged = np.concatenate([p["state"][f] for p in parts])[order]
        assert rel_err(merged, ctx.field(f)) <= 1e-11, f
    ctx.close()


def test_four_ranks_many_migrations_and_a_second_upload():
    """4 ranks, ownership re-decided every step (a rotating disc crosses slab edges all the time), 8 steps; then the same halo
    objects take a new particle set (sph_halo_upload again) and run it: the result of each must match a single context"""
    rows = ic.keplerian_disc(24000, seed=41)
    gas, sinks = ic.split_rows(rows)
    rng = np.random.default_rng(5)
    gas2 = {k: v.copy() for k, v in gas.items()}
    gas2["vx"] = gas2["vx"] + rng.normal(0.0, 0.02, gas2["vx"].size)
    world = 4
    hub = halo.Hub(world)
    out, errs = [None] * world, []

    def worker(rank):
        try:
            ctx = capi.Context(device=0)
            h = halo.Halo.inproc(ctx, hub, rank, world)
            res = []
            for g_ in (gas, gas2):
                bounds = slab_bounds(g_["x"], world)
                sel = np.searchsorted(bounds, g_["x"], side="right") == rank
                mine = {k: v[sel] for k, v in g_.items()}
                mine["gid"] = np.nonzero(sel)[0]
                ctx.set_sinks(sinks)
                h.set_slabs(bounds, 1)
                h.upload(mine)
                dt, t = h.run(8, 1e-2, 0.0)
                res.append({"dt": dt, "state": h.download(), "stats": h.stats()})
            out[rank] = res
            h.close(); ctx.close()
        except Exception as e:      # noqa: BLE001
            errs.append((rank, repr(e)))

    th = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    [x.start() for x in th]
    [x.join() for x in th]
    hub.close()
    assert not errs, errs
    for k, g_ in enumerate((gas, gas2)):
        ctx = capi.Context(device=0)
        ctx.upload(g_); ctx.set_sinks(sinks)
        dt, t = ctx.run(8, 1e-2, 0.0)
        gid = np.concatenate([out[r][k]["state"]["gid"] for r in range(world)])
        assert np.array_equal(np.sort(gid), np.arange(gid.size))
        order = np.argsort(gid)
        assert all(out[r][k]["dt"] == dt for r in range(world))
        for f in FIELDS:
            merged = np.concatenate([out[r][k]["state"][f] for r in range(world)])[order]
            assert rel_err(merged, ctx.field(f)) <= 1e-11, (k, f)
        ctx.close()
    assert out[0][1]["stats"].migrations >= 14 and out[0][1]["stats"].migrated > 0


def test_a_rank_that_owns_nothing():
    """three ranks, the third slab lies beyond the disc: the empty rank takes part in every collective and message round"""
    rows = ic.keplerian_disc(9000, seed=45)
    gas, sinks = ic.split_rows(rows)
    world = 3
    hub = halo.Hub(world)
    bounds = np.array([0.0, gas["x"].max() + 100.0])
    owner = np.searchsorted(bounds, gas["x"], side="right")
    out, errs = [None] * world, []

    def worker(rank):
        try:
            ctx = capi.Context(device=0)
            h = halo.Halo.inproc(ctx, hub, rank, world)
            sel = owner == rank
            mine = {k: v[sel] for k, v in gas.items()}
            mine["gid"] = np.nonzero(sel)[0]
            ctx.set_sinks(sinks); h.set_slabs(bounds, 2); h.upload(mine)
            dt, t = h.run(6, 1e-2, 0.0)
            out[rank] = (dt, h.download(), h.stats().ghosts)
            h.close(); ctx.close()
        except Exception as e:      # noqa: BLE001
            errs.append((rank, repr(e)))

    th = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    [x.start() for x in th]
    [x.join() for x in th]
    hub.close()
    assert not errs, errs
    assert out[2][1]["x"].size == 0 and out[2][2] == 0 and out[0][2] > 0
    ctx = capi.Context(device=0)
    ctx.upload(gas); ctx.set_sinks(sinks)
    dt, t = ctx.run(6, 1e-2, 0.0)
    order = np.argsort(np.concatenate([o[1]["gid"] for o in out]))
    assert all(o[0] == dt for o in out)
    for f in FIELDS:
        assert rel_err(np.concatenate([o[1][f] for o in out])[order], ctx.field(f)) <= 1e-11, f
    ctx.close()


def test_fortran_multi_gpu_host_one_rank(tmp_path):
    """run_sph_hip_mg with one rank (RCCL communicator of size 1, id through the file) writes what run_sph_hip ... sph
    writes: same dt decisions, same snapshot, byte for byte"""
    import os
    import subprocess
    from conftest import ROOT
    from summersph_amd import txtio
    host = os.path.join(ROOT, "summersph_amd", "host")
    if not os.path.exists(os.path.join(host, "run_sph_hip_mg")):
        subprocess.run(["make", "-C", host], check=True, stdout=subprocess.DEVNULL)
    g = load_golden("disc3000_traj")
    icf = tmp_path / "ic.txt"
    txtio.write_ic(str(icf), g["ic"])
    a = subprocess.run([os.path.join(host, "run_sph_hip"), str(icf), "5", str(tmp_path / "one.txt"), "sph"],
                       capture_output=True, text=True, cwd=tmp_path, timeout=300)
    assert a.returncode == 0, a.stdout + a.stderr
    b = subprocess.run([os.path.join(host, "run_sph_hip_mg"), "0", "1", str(tmp_path / "id.bin"), str(icf), "5", str(tmp_path / "mg.txt")],
                       capture_output=True, text=True, cwd=tmp_path, timeout=300)
    assert b.returncode == 0, b.stdout + b.stderr
    dts = lambda out: [l for l in out.splitlines() if l.startswith("dt ")]
    assert dts(a.stdout) == dts(b.stdout) and len(dts(b.stdout)) == 6
    assert [float(l.split()[2]) for l in dts(b.stdout)] == list(g["sph_dt_seq"])
    assert (tmp_path / "one.txt").read_bytes() == (tmp_path / "mg.txt").read_bytes()
    # the periodic saves through the collective gather: the same files as the single-GPU host writes
    for sub, cmd in (("s1", [os.path.join(host, "run_sph_hip"), str(icf), "4", str(tmp_path / "s1" / "f.txt"), "sph", "saves", "tend=20"]),
                     ("s2", [os.path.join(host, "run_sph_hip_mg"), "0", "1", str(tmp_path / "id2.bin"), str(icf), "4",
                             str(tmp_path / "s2" / "f.txt"), "saves", "tend=20"])):
        (tmp_path / sub).mkdir()
        r = subprocess.run(cmd, capture_output=True, text=True, cwd=tmp_path / sub, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
    names = sorted(p.name for p in (tmp_path / "s1").glob("save*.txt"))
    assert len(names) >= 2 and names == sorted(p.name for p in (tmp_path / "s2").glob("save*.txt"))
    for nm in names:
        assert (tmp_path / "s1" / nm).read_bytes() == (tmp_path / "s2" / nm).read_bytes(), nm


def test_bench_glue_of_the_native_loop_one_rank():
    """bench.py's NativeSim (what `bench.py --gpus N` runs by default) with a one-rank RCCL communicator and a stub for the
    gloo control group: create -> self-test -> upload -> run -> stats, against sph_run"""
    import sys
    import torch
    from conftest import ROOT
    sys.path.insert(0, ROOT)
    import bench

    class OneRank:                       # what NativeSim.create asks of torch.distributed
        class ReduceOp:
            MIN = None

        @staticmethod
        def broadcast_object_list(objs, src=0):
            return None

        @staticmethod
        def all_reduce(t, op=None):
            return None

    rows = ic.keplerian_disc(20000, seed=43)
    gas, sinks = ic.split_rows(rows)
    mine = dict(gas); mine["gid"] = np.arange(gas["x"].size)
    env = {"capi": capi, "dist": OneRank, "torch": torch, "rank": 0, "world": 1, "local_rank": 0}
    sim = bench.NativeSim.create(env, 0, mine, sinks, np.zeros(0))
    assert sim is not None
    dt = sim.run(3, 1e-2)
    dt = sim.run(4, dt)
    assert sim.n_owned == gas["x"].size and sim.stats["ghosts"] == 0
    ctx = capi.Context(device=0)
    ctx.upload(gas); ctx.set_sinks(sinks)
    d2, t2 = ctx.run(7, 1e-2, 0.0)
    assert dt == d2 and sim.t == t2
    assert np.array_equal(sim.h.download()["x"], ctx.field("x"))
    sim.h.close(); sim.ctx.close(); ctx.close()
