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
    b = subprocess.run([os.path.join(host, "run_sph_hip_mg"), "0", "1", str(tmp_path / "id.bin"), str(icf), "5", str(tmp_path / "mg.txt"), "sph"],
                       capture_output=True, text=True, cwd=tmp_path, timeout=300)
    assert b.returncode == 0, b.stdout + b.stderr
    dts = lambda out: [l for l in out.splitlines() if l.startswith("dt ")]
    assert dts(a.stdout) == dts(b.stdout) and len(dts(b.stdout)) == 6
    assert [float(l.split()[2]) for l in dts(b.stdout)] == list(g["sph_dt_seq"])
    assert (tmp_path / "one.txt").read_bytes() == (tmp_path / "mg.txt").read_bytes()
    # the periodic saves through the collective gather: the same files as the single-GPU host writes
    for sub, cmd in (("s1", [os.path.join(host, "run_sph_hip"), str(icf), "4", str(tmp_path / "s1" / "f.txt"), "sph", "saves", "tend=20"]),
                     ("s2", [os.path.join(host, "run_sph_hip_mg"), "0", "1", str(tmp_path / "id2.bin"), str(icf), "4",
                             str(tmp_path / "s2" / "f.txt"), "sph", "saves", "tend=20"])):
        (tmp_path / sub).mkdir()
        r = subprocess.run(cmd, capture_output=True, text=True, cwd=tmp_path / sub, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
    names = sorted(p.name for p in (tmp_path / "s1").glob("save*.txt"))
    assert len(names) >= 2 and names == sorted(p.name for p in (tmp_path / "s2").glob("save*.txt"))
    for nm in names:
        assert (tmp_path / "s1" / nm).read_bytes() == (tmp_path / "s2" / nm).read_bytes(), nm


def test_fortran_multi_gpu_host_one_rank_full_loop(tmp_path):
    """run_sph_hip_mg WITHOUT `sph` = simulate() as the reference runs it (Barnes-Hut self-gravity, accretion, cull): with one
    rank it writes what run_sph_hip writes, byte for byte, and both reproduce the real reference's dt decisions and particle
    count (2000 -> 1996)"""
    import os
    import subprocess
    from conftest import ROOT
    from summersph_amd import txtio
    host = os.path.join(ROOT, "summersph_amd", "host")
    g = load_golden("acc2000_traj")
    icf = tmp_path / "ic.txt"
    txtio.write_ic(str(icf), g["ic"])
    a = subprocess.run([os.path.join(host, "run_sph_hip"), str(icf), "3", str(tmp_path / "one.txt")],
                       capture_output=True, text=True, cwd=tmp_path, timeout=300)
    assert a.returncode == 0, a.stdout + a.stderr
    b = subprocess.run([os.path.join(host, "run_sph_hip_mg"), "0", "1", str(tmp_path / "id.bin"), str(icf), "3", str(tmp_path / "mg.txt")],
                       capture_output=True, text=True, cwd=tmp_path, timeout=300)
    assert b.returncode == 0, b.stdout + b.stderr
    dts = lambda out: [l for l in out.splitlines() if l.startswith("dt ")]
    assert dts(a.stdout) == dts(b.stdout)
    assert [float(l.split()[2]) for l in dts(b.stdout)] == list(g["full_dt_seq"])
    assert (tmp_path / "one.txt").read_bytes() == (tmp_path / "mg.txt").read_bytes()
    gas, sinks = txtio.read_snapshot(str(tmp_path / "mg.txt"))
    assert gas.shape[0] == int(g["full_n_seq"][-1]) == 1996
    assert "accreted+culled 4" in b.stdout


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


# ---- the loop the reference actually runs: self-gravity, accretion + cull, variable h, sink creation -----------------------

def _run_ranks_ex(world, gas, sinks, nsteps, ctx_kw, migrate_every=2):
    """as _run_ranks for any context flavour: per rank the dt sequence, owned counts per step, final state and sinks"""
    hub = halo.Hub(world)
    bounds = slab_bounds(gas["x"], world)
    owner = np.searchsorted(bounds, gas["x"], side="right")
    out, errs = [None] * world, []

    def worker(rank):
        try:
            ctx = capi.Context(device=0, **ctx_kw)
            h = halo.Halo.inproc(ctx, hub, rank, world)
            sel = owner == rank
            mine = {k: v[sel] for k, v in gas.items()}
            mine["gid"] = np.nonzero(sel)[0]
            ctx.set_sinks(sinks)
            h.set_slabs(bounds, migrate_every)
            h.upload(mine)
            dts, ns, t = [1e-2], [], 0.0
            for _ in range(nsteps):
                dt, t = h.run(1, dts[-1], t)
                dts.append(dt); ns.append(h.n_owned)
            out[rank] = {"dts": dts, "ns": np.array(ns), "t": t, "state": h.download(), "stats": h.stats(), "sinks": ctx.get_sinks()}
            h.close(); ctx.close()
        except Exception as e:      # noqa: BLE001 -- reported by the test below
            errs.append((rank, repr(e)))
            hub_fail.set()

    hub_fail = threading.Event()
    th = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    for t_ in th:
        t_.start()
    for t_ in th:
        t_.join()
    hub.close()
    assert not errs, errs
    return out


def _merged(parts, f):
    order = np.argsort(np.concatenate([p["state"]["gid"] for p in parts]))
    return np.concatenate([p["state"][f] for p in parts])[order]


def _var_kw(g, full):
    gamma, eta, tol, maxlen, scale = (float(v) for v in g["params"])
    kw = dict(variable=True, gamma=gamma, gamma_m1=gamma - 1.0, eta=eta, h_tol=tol, h_max_length=maxlen, dt_scale=scale)
    if full:
        kw["flags"] = capi.FLAG_VARIABLE_H | capi.FLAG_SELF_GRAVITY | capi.FLAG_ACCRETE_CULL | capi.FLAG_SINK_CREATION
    return kw


@pytest.mark.parametrize("world", [2, 3])
def test_native_loop_self_gravity_vs_reference(world):
    """find_forces as the reference has it (Barnes-Hut term on the replicated tree of all-gathered sources) through the native
    loop: the real reference's `full` trajectory of the 3000-particle disc"""
    g = load_golden("disc3000_traj")
    gas, sinks = ic.split_rows(g["ic"])
    parts = _run_ranks_ex(world, gas, sinks, 5, dict(flags=capi.FLAG_SELF_GRAVITY))
    for p in parts:
        assert list(p["dts"]) == list(g["full_dt_seq"])
        assert p["stats"].ghosts > 0
    for f in FIELDS:
        assert rel_err(_merged(parts, f), g["full_s5_" + f]) <= 1e-10, f


@pytest.mark.parametrize("name,world", [("acc2000_traj", 2), ("acc2000_traj", 3), ("bin2000_traj", 2)])
def test_native_loop_accretion_and_cull_vs_reference(name, world):
    """simulate()'s whole loop body on several ranks: shared-tree gravity, accretion (one / two accretors), boundary cull --
    identical dt decisions and particle counts, sinks and state against the real reference"""
    g = load_golden(name)
    gas, sinks = ic.split_rows(g["ic"])
    parts = _run_ranks_ex(world, gas, sinks, 3, dict(flags=capi.FLAG_SELF_GRAVITY | capi.FLAG_ACCRETE_CULL))
    assert list(sum(p["ns"] for p in parts)) == [int(v) for v in g["full_n_seq"][1:]]
    gid = np.concatenate([p["state"]["gid"] for p in parts])
    assert np.unique(gid).size == gid.size == int(g["full_n_seq"][-1])
    assert sum(p["stats"].removed for p in parts) == int(g["full_n_seq"][0] - g["full_n_seq"][-1])
    for p in parts:
        assert list(p["dts"]) == list(g["full_dt_seq"])
        assert np.max(np.abs(p["sinks"]["m"] - g["full_s3_sm"])) <= 1e-15 and np.max(np.abs(p["sinks"]["x"] - g["full_s3_sx"])) <= 1e-12
        assert np.array_equal(p["sinks"]["m"], parts[0]["sinks"]["m"])
    for f in FIELDS:
        assert rel_err(_merged(parts, f), g["full_s3_" + f]) <= 1e-10, f


@pytest.mark.parametrize("variant,world", [("sph", 2), ("sph", 3), ("full", 2)])
def test_native_loop_variable_h_vs_reference(variant, world):
    """"SUMMER_SPH - Variable.f90" through the native loop: per-particle h and global numbers in the ghost payload, the leaf
    boxes of the octree of all particles, rho AND Omega of the ghosts, calc_smoothing per rank; `full` adds self-gravity,
    accretion and cull"""
    g = load_golden("discv3000_traj")
    gas, sinks = ic.split_rows(g["ic"])
    parts = _run_ranks_ex(world, gas, sinks, 5, _var_kw(g, variant == "full"))
    gid = np.concatenate([p["state"]["gid"] for p in parts])
    assert np.array_equal(np.sort(gid), np.arange(gid.size))
    for p in parts:
        assert list(p["dts"]) == list(g[variant + "_dt_seq"])
        assert p["stats"].ghosts > 0
    for f in FIELDS + ["h"]:
        assert rel_err(_merged(parts, f), g[f"{variant}_s5_" + f]) <= 1e-10, f


def test_native_loop_sink_creation_vs_reference():
    """check_sink_creation on 2 ranks: the candidate with the lowest global number wins one all-gather, every rank adds the same
    sink, which then accretes its seed"""
    g = load_golden("sinkcv1500_traj")
    gas, sinks = ic.split_rows(g["ic"])
    parts = _run_ranks_ex(2, gas, sinks, 3, _var_kw(g, True))
    gid = np.concatenate([p["state"]["gid"] for p in parts])
    assert gid.size == 1499
    for p in parts:
        assert list(p["dts"]) == list(g["full_dt_seq"])
        assert p["stats"].sinks_created == 1
        assert p["sinks"]["m"].size == 2 and np.max(np.abs(p["sinks"]["m"] - g["full_s3_sm"]) / g["full_s3_sm"]) <= 1e-14
        assert np.max(np.abs(p["sinks"]["x"] - g["full_s3_sx"])) <= 1e-9
    for f in FIELDS + ["h"]:
        assert rel_err(_merged(parts, f), g["full_s3_" + f]) <= 1e-9, f


def test_native_loop_sink_cull_vs_reference():
    """[V]'s check_bounds drops the sink that starts outside the box -- on every rank alike"""
    g = load_golden("sinkcullv1000_traj")
    gas, sinks = ic.split_rows(g["ic"])
    parts = _run_ranks_ex(2, gas, sinks, 3, _var_kw(g, True))
    for p in parts:
        assert list(p["dts"]) == list(g["full_dt_seq"])
        assert p["sinks"]["m"].size == 1 and p["sinks"]["m"][0] == g["full_s3_sm"][0]
    for f in FIELDS + ["h"]:
        assert rel_err(_merged(parts, f), g["full_s3_" + f]) <= 1e-10, f


@pytest.mark.parametrize("flavour", ["full", "variable_full"])
def test_native_loop_one_rank_is_sph_run_on_the_octree_paths(flavour):
    """one rank, no messages: gravity, accretion + cull (and variable h with sink creation) through the native loop must give
    what sph_run gives, bit for bit -- state, sinks, particle count, dt"""
    if flavour == "full":
        g = load_golden("acc2000_traj"); kw = dict(flags=capi.FLAG_SELF_GRAVITY | capi.FLAG_ACCRETE_CULL); names = FIELDS
    else:
        g = load_golden("sinkcv1500_traj"); kw = _var_kw(g, True); names = FIELDS + ["h"]
    gas, sinks = ic.split_rows(g["ic"])
    one = _run_ranks_ex(1, gas, sinks, 3, kw)[0]
    ctx = capi.Context(device=0, **kw)
    ctx.upload(gas); ctx.set_sinks(sinks)
    dt, t = 1e-2, 0.0
    for _ in range(3):
        dt, t = ctx.run(1, dt, t)
    assert one["dts"][-1] == dt and one["t"] == t and one["state"]["x"].size == ctx.n
    assert one["state"]["x"].size == int(g["full_n_seq"][-1])
    for f in names:
        assert np.array_equal(one["state"][f], ctx.field(f)), f
    s = ctx.get_sinks()
    for k in ("x", "vx", "m"):
        assert np.array_equal(one["sinks"][k], s[k]), k
    # the survivors kept their global numbers: exactly the particles the reference kept
    assert np.all(np.diff(one["state"]["gid"]) > 0)
    ctx.close()


@pytest.mark.parametrize("world", [2, 3])
def test_locally_essential_tree_matches_the_replicated_tree(world, monkeypatch):
    """self-gravity on several ranks: by default every rank ships, per receiver box, the coarsest octree cells that box accepts
    whatever their centre of mass (as {com, mass} pseudo-particles) and single particles elsewhere; SPH_HALO_REPLICATED=1 ships
    every particle to everybody (what dist.py does).  Same accepted sets for every target, sums in another order: identical dt
    decisions, states equal to <= 1e-12 -- and far fewer bytes"""
    rows = ic.keplerian_disc(40_000, seed=47, nngb=85.0)
    gas, sinks = ic.split_rows(rows)
    kw = dict(flags=capi.FLAG_SELF_GRAVITY | capi.FLAG_ACCRETE_CULL)
    let = _run_ranks_ex(world, gas, sinks, 4, kw)
    monkeypatch.setenv("SPH_HALO_REPLICATED", "1")
    rep = _run_ranks_ex(world, gas, sinks, 4, kw)
    n_total = gas["x"].size
    for a, b in zip(let, rep):
        assert a["dts"] == b["dts"] and list(a["ns"]) == list(b["ns"])
        assert a["stats"].let_updates > 0 and b["stats"].let_updates == 0
        # per source update a rank receives far fewer records than the other ranks hold
        others = n_total - a["state"]["x"].size
        assert a["stats"].let_received / a["stats"].let_updates < 0.5 * others
    for f in FIELDS:
        assert rel_err(_merged(let, f), _merged(rep, f)) <= 1e-12, f
    for k in ("x", "vx", "m"):
        assert np.max(np.abs(let[0]["sinks"][k] - rep[0]["sinks"][k])) <= 1e-13
