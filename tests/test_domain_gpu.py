"""The device-side building blocks of the multi-GPU exchange (include/summersph.h, "kept on the device";
csrc/domain.hip) one by one, through the C ABI, against numpy on the same seeded inputs:
owned bounding box, box selection, ghost swap inside the context, rank reductions, dt held on the device."""
import numpy as np
import pytest

from conftest import rel_err
from summersph_amd import ic

pytestmark = pytest.mark.gpu
STATE = "x y z vx vy vz u m alpha".split()


@pytest.fixture(scope="module")
def capi():
    from summersph_amd import capi as m
    m.load()
    return m


@pytest.fixture(scope="module")
def torch():
    import torch as t
    return t


def _disc(n, seed):
    gas, sinks = ic.split_rows(ic.keplerian_disc(n, seed=seed))
    gas["alpha"] = np.full(n, 0.1)
    return gas, sinks


def _split(gas, frac=0.7):
    """owned = the left part of the disc (by x), ghosts = the particles within 2h = 5 of it"""
    x = gas["x"]
    cut = np.quantile(x, frac)
    own = np.nonzero(x <= cut)[0]
    gh = np.nonzero((x > cut) & (x <= cut + 5.0))[0]
    return own, gh, cut


def _upload(capi, torch, gas, idx, reserve=0):
    ctx = capi.Context(device=0)
    if reserve:
        ctx.reserve(reserve)
    dev = [torch.from_numpy(np.ascontiguousarray(gas[k][idx])).cuda() for k in STATE]
    torch.cuda.synchronize()
    ctx.upload_dev(idx.size, [t.data_ptr() for t in dev])
    return ctx


def test_owned_bbox_and_box_selection(capi, torch):
    gas, sinks = _disc(20000, 5)
    own, gh, cut = _split(gas)
    both = np.concatenate([own, gh])
    ctx = _upload(capi, torch, gas, both)
    ctx.set_owned(own.size)
    ctx.set_sinks(sinks)
    pos = np.stack([gas[k][own] for k in "xyz"])
    assert np.array_equal(ctx.owned_bbox(), np.concatenate([pos.min(1), pos.max(1)]))
    ctx.density()                                        # cell-sorted inside now; ids must stay the caller's
    out = torch.empty(6, dtype=torch.float64, device="cuda")
    ctx.owned_bbox(out.data_ptr())
    ctx.synchronize()
    assert np.array_equal(out.cpu().numpy(), np.concatenate([pos.min(1), pos.max(1)]))
    boxes = np.array([[cut - 5.0, -1e9, -1e9, 1e9, 1e9, 1e9],          # the strip a right-hand neighbour needs
                      [-20.0, -20.0, -1.0, 20.0, 20.0, 1.0],
                      [1e8, 1e8, 1e8, 2e8, 2e8, 2e8]])                  # empty
    counts = ctx.select_boxes(boxes)
    for b, box in enumerate(boxes):
        want = np.nonzero(np.all((pos >= box[:3, None]) & (pos <= box[3:, None]), axis=0))[0]
        assert counts[b] == want.size
        ids = torch.empty(int(counts[b]), dtype=torch.int64, device="cuda")
        ctx.selected_ids_dev(b, int(counts[b]), ids.data_ptr())
        ctx.synchronize()
        assert np.array_equal(ids.cpu().numpy(), want)                  # ascending original ids, owned only
    assert counts[2] == 0
    ctx.close()


def test_ghost_swap_matches_fresh_upload(capi, torch):
    """replace the ghosts of a context twice (with a drift in between); each time rho and the rates of the owned
    particles must equal those of a fresh context that was uploaded with the same owned + ghost set"""
    gas, sinks = _disc(20000, 6)
    own, gh, cut = _split(gas)
    ctx = _upload(capi, torch, gas, own, reserve=own.size + 3 * gh.size + 1000)
    ctx.set_sinks(sinks)
    n0 = own.size
    for rnd in range(2):
        g = {k: v.copy() for k, v in gas.items()}
        if rnd == 1:                                     # second round: everybody has moved, other ghosts
            ctx.drift(0.05)
            for a, va in zip("xyz", ("vx", "vy", "vz")):
                g[a] = g[a] + g[va] * 0.05
                g[a][own] = ctx.field(a)[:n0]            # the owned particles' positions as the device drifted them
            gh = np.setdiff1d(np.nonzero((g["x"] > g["x"][own].max()) & (g["x"] <= g["x"][own].max() + 5.0))[0], own)
        ghosts = torch.from_numpy(np.stack([g[k][gh] for k in STATE])).cuda()
        torch.cuda.synchronize()
        ctx.replace_ghosts_dev(gh.size, ghosts.data_ptr())
        assert ctx.n == n0 + gh.size
        ctx.density()
        ref = _upload(capi, torch, g, np.concatenate([own, gh]))
        ref.set_owned(n0)
        ref.set_sinks(sinks)
        ref.density()
        assert rel_err(ctx.field("rho")[:n0], ref.field("rho")[:n0]) <= 1e-14
        for f in STATE:                                  # ids: owned keep theirs, ghosts follow in the order given
            assert np.array_equal(ctx.field(f), ref.field(f)), f
        # ghosts get the rho "their owner sent" (any positive values will do for this comparison), then the EOS refresh
        rho_g = torch.from_numpy(np.mean(ref.field("rho")[:n0]) * (1.0 + 0.1 * np.sin(np.arange(gh.size)))[None, :].copy()).cuda()
        torch.cuda.synchronize()
        for cx in (ctx, ref):
            cx.scatter_fields_dev(["rho"], n0, gh.size, rho_g.data_ptr())
            cx.refresh_eos()
            cx.forces()
        for f in ("ax", "ay", "az", "du", "dalpha"):
            assert rel_err(ctx.field(f)[:n0], ref.field(f)[:n0]) <= 1e-13, f
        ref.close()
    # too many ghosts for the reserved slots: refused, nothing written
    big = torch.zeros((9, 4 * own.size), dtype=torch.float64, device="cuda")
    with pytest.raises(capi.SphError):
        ctx.replace_ghosts_dev(big.shape[1], big.data_ptr())
    ctx.close()


def test_rank_reduction_and_device_dt(capi, torch):
    gas, sinks = _disc(5000, 7)
    ctx = _upload(capi, torch, gas, np.arange(5000))
    ctx.set_sinks(sinks)
    ctx.density(); ctx.forces()
    part = ctx.get_sinks()
    ctx.set_dt(0.01, 2.0)
    ctx.dt_candidate_dev()
    mine = torch.empty(199, dtype=torch.float64, device="cuda")     # SPH_PARTIALS
    ctx.pack_partials_dev(mine.data_ptr())
    ctx.synchronize()
    m = mine.cpu().numpy()
    assert m[0] == part["ax"][0] and m[64] == part["ay"][0] and m[128] == part["az"][0]
    assert m[192] == ctx.dt_candidate()                  # the local part of get_next_timestep
    # three "ranks": mine, a second block, a third one with the smallest dt candidate
    allp = np.stack([m, 2.0 * m, -0.5 * m])
    allp[1, 192] = 10.0 * m[192]; allp[2, 192] = 0.004      # < 0.5 dt: the step halves ([F]:857-858)
    dev = torch.from_numpy(allp).cuda()
    torch.cuda.synchronize()
    # [193, 199): where the particles will be after the coming kick + drift, for dt' in {0.5, 1, 1.5} dt
    a = {k: ctx.field(k) for k in "x vx ax".split()}
    q = np.stack([a["x"] + (a["vx"] + 0.5 * a["ax"] * f * 0.01) * f * 0.01 for f in (0.5, 1.0, 1.5)])
    assert abs(m[193] - q.min()) <= 1e-12 and abs(m[196] - q.max()) <= 1e-12
    ctx.apply_partials_dev(dev.data_ptr(), 3, 199, True)
    s = ctx.get_sinks()
    assert s["ax"][0] == (m[0] + 2.0 * m[0]) + -0.5 * m[0]       # rank order
    dt, t = ctx.get_dt()
    assert (dt, t) == (0.005, 2.01)
    # the kicks and the drift take dt from the device
    v0, x0, a0 = ctx.field("vx"), ctx.field("x"), ctx.field("ax")
    ctx.kick_devdt()
    v1 = ctx.field("vx")
    assert rel_err(v1, v0 + 0.5 * a0 * 0.005) <= 1e-15             # [F]:749 (the device may contract a*dt + v)
    ctx.drift_devdt()
    assert rel_err(ctx.field("x"), x0 + v1 * 0.005) <= 1e-15       # [F]:769
    ctx.close()


def test_selection_without_a_wait(capi, torch):
    """sph_select_boxes_async + sph_gather_selected_dev + sph_selected_counts (the ghost exchange of csrc/halo.hip) against the
    waiting forms: same counts, same ids, same packed rows; a selection that does not fit its room leaves the header alone"""
    gas, sinks = _disc(20000, 11)
    ctx = _upload(capi, torch, gas, np.arange(20000))
    ctx.set_sinks(sinks)
    ctx.density()
    x = gas["x"]
    boxes = np.array([[-1e9, -1e9, -1e9, np.quantile(x, 0.1), 1e9, 1e9],
                      [np.quantile(x, 0.7), -1e9, -1e9, 1e9, 1e9, 1e9],
                      [1e8, 1e8, 1e8, 2e8, 2e8, 2e8]])                     # the last one is empty
    want = ctx.select_boxes(boxes)
    ids_ref, rows_ref = [], []
    names = ["x", "vy", "rho"]
    for b, cnt in enumerate(want):
        ids = torch.empty(int(cnt), dtype=torch.int64, device="cuda")
        ctx.selected_ids_dev(b, int(cnt), ids.data_ptr())
        out = torch.empty((3, int(cnt)), dtype=torch.float64, device="cuda")
        if cnt:
            ctx.gather_fields_dev(names, int(cnt), ids.data_ptr(), out.data_ptr())
        ctx.synchronize()
        ids_ref.append(ids.cpu().numpy()); rows_ref.append(out.cpu().numpy())
    assert want[0] == np.count_nonzero(x <= boxes[0, 3]) and want[2] == 0
    nb = ctx.select_boxes_async(boxes)
    room = [int(want[0]) + 10, int(want[1]) - 1, 0]                       # fits, does not fit, empty
    outs = [torch.full((2 + 3 * max(r, 1),), -7.0, dtype=torch.float64, device="cuda") for r in room]
    for b in range(nb):
        ctx.gather_selected_dev(b, names, room[b], outs[b].data_ptr())
    ctx.synchronize()
    assert np.array_equal(ctx.selected_counts(nb), want)
    got = [o.cpu().numpy() for o in outs]
    assert got[0][0] == want[0] and got[0][1] == 0.0
    assert np.array_equal(got[0][2:2 + 3 * want[0]].reshape(3, -1), rows_ref[0])
    assert got[1][0] == want[1] and np.all(got[1][2:] == -7.0)           # header alone
    assert got[2][0] == 0.0
    ids = torch.empty(int(want[1]), dtype=torch.int64, device="cuda")
    ctx.selected_ids_dev(1, int(want[1]), ids.data_ptr())
    ctx.synchronize()
    assert np.array_equal(ids.cpu().numpy(), ids_ref[1])
    ctx.close()


def test_fused_kick_calls_are_bitwise_the_separate_ones(capi, torch):
    """sph_kick_drift_devdt == sph_kick_devdt + sph_drift_devdt; sph_kick_dt_candidate_dev == sph_kick_devdt +
    sph_dt_candidate_dev (state, sink velocities and the candidate).  All particles owned: ghost rho / rates are whatever
    their owners last sent, which this test does not stage (the ghost protocol is covered by test_halo_gpu / test_dist_gpu)"""
    gas, sinks = _disc(6000, 9)
    res = []
    for fused in (False, True, "split"):
        ctx = _upload(capi, torch, gas, np.arange(6000))
        ctx.set_sinks(sinks)
        ctx.set_dt(0.013, 0.0)
        ctx.density(); ctx.forces()
        if fused:
            ctx.kick_drift_devdt()
        else:
            ctx.kick_devdt(); ctx.drift_devdt()
        ctx.density(); ctx.forces()
        if fused == "split":          # the gas and the sink half of the closing kick as two calls
            ctx.kick_dt_candidate_gas_dev(); ctx.kick_sinks_devdt()
        elif fused:
            ctx.kick_dt_candidate_dev()
        else:
            ctx.kick_devdt(); ctx.dt_candidate_dev()
        out = {k: ctx.field(k) for k in "x y z vx vy vz u alpha".split()}
        mine = torch.empty(199, dtype=torch.float64, device="cuda")
        ctx.pack_partials_dev(mine.data_ptr())
        ctx.synchronize()
        out["cand"] = mine.cpu().numpy()[192:193]
        s = ctx.get_sinks()
        out["sx"], out["svx"] = s["x"], s["vx"]
        res.append(out)
        ctx.close()
    for k in res[0]:
        assert np.array_equal(res[0][k], res[1][k]), k
        assert np.array_equal(res[0][k], res[2][k]), k


def test_context_on_callers_stream(capi, torch):
    """sph_set_stream: the context runs on torch's current stream; device-pointer calls then need no host
    synchronisation to be ordered with torch kernels on that stream"""
    gas, sinks = _disc(8000, 8)
    ctx = capi.Context(device=0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    dev = [torch.from_numpy(gas[k]).cuda() for k in STATE]
    ctx.upload_dev(8000, [t.data_ptr() for t in dev])
    ctx.set_sinks(sinks)
    ctx.density()
    ids = torch.arange(0, 8000, 7, dtype=torch.int64, device="cuda")
    out = torch.empty((2, ids.numel()), dtype=torch.float64, device="cuda")
    ctx.gather_fields_dev(["rho", "x"], ids.numel(), ids.data_ptr(), out.data_ptr())
    doubled = out * 2.0                                   # torch kernel queued behind the gather, same stream
    got = doubled.cpu().numpy()
    assert np.array_equal(got[1], 2.0 * gas["x"][::7])
    assert np.array_equal(got[0], 2.0 * ctx.field("rho")[::7])
    ctx.close()


def test_split_force_evaluation_is_bitwise_the_full_one(capi, torch):
    """sph_forces_part 1 + 2 (interior wavefronts, then the ones near the other GPUs' boxes) == sph_forces"""
    gas, sinks = _disc(30000, 9)
    own, gh, cut = _split(gas)
    both = np.concatenate([own, gh])
    ctxs = []
    for _ in range(2):
        c = _upload(capi, torch, gas, both)
        c.set_owned(own.size)
        c.set_sinks(sinks)
        c.density()
        rho_g = torch.from_numpy(np.full((1, gh.size), 1.2e-4)).cuda()
        torch.cuda.synchronize()
        c.scatter_fields_dev(["rho"], own.size, gh.size, rho_g.data_ptr())
        c.refresh_eos()
        ctxs.append(c)
    full, split = ctxs
    full.forces()
    with pytest.raises(capi.SphError):
        split.forces_part(2)                                   # part 2 before part 1
    gpos = np.stack([gas[k][gh] for k in "xyz"])
    split.set_boundary_boxes(np.concatenate([gpos.min(1), gpos.max(1)])[None, :])
    split.forces_part(1)
    with pytest.raises(capi.SphError):
        split.kick(0.01)                                       # rates are not complete yet
    split.forces_part(2)
    for f in ("ax", "ay", "az", "du", "dalpha"):
        assert np.array_equal(split.field(f)[:own.size], full.field(f)[:own.size]), f
    a, b = full.get_sinks(), split.get_sinks()
    assert np.array_equal(a["ax"], b["ax"])
    # no boxes at all: everything is interior, part 2 has nothing left to do but completes the evaluation
    split.set_boundary_boxes(np.zeros((0, 6)))
    split.forces_part(1); split.forces_part(2)
    assert np.array_equal(split.field("ax")[:own.size], full.field("ax")[:own.size])
    for c in ctxs:
        c.close()
