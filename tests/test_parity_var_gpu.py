"""GPU parity of the VARIABLE-h path (SPH_FLAG_VARIABLE_H) through the C ABI, against fixtures dumped
from the real variable-h reference ("SUMMER_SPH - Variable.f90") and against the CPU oracle.

Tolerances as in test_parity_gpu.py: 1e-13 for a single evaluation (rho, Omega, P, c, rates),
1e-12 for the updated smoothing lengths, identical dt decisions, 1e-10 after 5 steps (the h
iteration feeds rounding noise back into the neighbour sets)."""
import numpy as np
import pytest

from conftest import load_golden, rel_err
from summersph_amd import ic

pytestmark = pytest.mark.gpu
TOL = 1e-13


@pytest.fixture(scope="module")
def capi():
    from summersph_amd import capi as m
    m.load()
    return m


def make_ctx(capi, g, **kw):
    gas, sinks = ic.split_rows(g["ic"])
    gamma, eta, tol, maxlen, scale = g["params"]
    ctx = capi.Context(device=0, variable=True, gamma=gamma, gamma_m1=gamma - 1.0, eta=eta, h_tol=tol,
                       h_max_length=maxlen, dt_scale=scale, **kw)
    ctx.upload(gas)
    ctx.set_sinks(sinks)
    return ctx, gas, sinks


@pytest.mark.parametrize("name", ["discv3000_eval", "discv2000r_eval"])
def test_single_evaluation_vs_reference_fixture(capi, name):
    g = load_golden(name)
    ctx, gas, sinks = make_ctx(capi, g)
    assert np.array_equal(ctx.field("h"), g["h"])
    ctx.density()
    for f in ("rho", "omega", "P", "c"):
        assert rel_err(ctx.field(f), g[f]) <= TOL, f
    ctx.forces()
    for f in ("ax", "ay", "az", "du", "dalpha"):
        assert rel_err(ctx.field(f), g["sph_" + f]) <= TOL, f
    assert ctx.next_dt(1e-2) == g["sph_dt"][0]
    ctx.update_h()
    assert rel_err(ctx.field("h"), g["sph_hnew"]) <= 1e-12
    ctx.close()


def test_trajectory_vs_reference_fixture(capi):
    g = load_golden("discv3000_traj")
    ctx, gas, sinks = make_ctx(capi, g)
    dts, t = [1e-2], 0.0
    for k in range(1, 6):
        dt, t = ctx.step(dts[-1], t)
        dts.append(dt)
        if k in (1, 5):
            p = f"sph_s{k}_"
            for f in "x y z vx vy vz u alpha h".split():
                assert rel_err(ctx.field(f), g[p + f]) <= 1e-10, (k, f)
    assert dts == list(g["sph_dt_seq"])
    ctx.close()


def test_unfused_equals_fused(capi):
    g = load_golden("discv3000_traj")
    a, _, _ = make_ctx(capi, g)
    b, _, _ = make_ctx(capi, g)
    da = 1e-2
    for _ in range(2):
        a.density(); a.forces(); a.kick(da); a.drift(da); a.density(); a.forces(); a.kick(da)
        nd = a.next_dt(da); a.update_h(); da = nd
    db, _ = b.run(2, 1e-2, 0.0)
    assert da == db
    for f in "x vx u h".split():
        assert np.array_equal(a.field(f), b.field(f)), f
    a.close(); b.close()


def test_disc_vs_oracle_larger(capi):
    """20k variable-h disc: single evaluation + h update against the CPU oracle"""
    from oracle import orc, orc_v
    rows = ic.keplerian_disc_var(20000, seed=31)
    gas, sinks = ic.split_rows(rows)
    ctx = capi.Context(device=0, variable=True)
    ctx.upload(gas); ctx.set_sinks(sinks)
    o = orc_v.OracleV(gas, sinks, nthreads=orc.max_threads())
    ctx.density(); ctx.forces(); o.evaluate()
    for f in ("rho", "omega", "ax", "ay", "az", "du", "dalpha"):
        assert rel_err(ctx.field(f), getattr(o, f)) <= TOL, f
    assert ctx.next_dt(1e-2) == o.next_dt(1e-2)
    ctx.update_h(); o.update_h()
    assert rel_err(ctx.field("h"), o.h) <= 1e-12
    ctx.close()


def test_pair_closer_than_the_key_resolution(capi):
    """two particles closer than root_edge / 2^21 share their 63-bit path key; the reference splits on (to depth 1000) and
    gives each a tiny leaf.  A third body placed just outside that tiny box -- but inside the box a level-21 leaf would have
    -- tells the two rules apart: it must NOT count the pair in its density sum.  Checked against the CPU oracle, which
    recurses like the reference."""
    from oracle import orc, orc_v
    rows = ic.keplerian_disc_var(4000, seed=35)
    gas, sinks = ic.split_rows(rows)
    pos = np.stack([gas[k] for k in "xyz"])
    lo, hi = pos.min(1), pos.max(1)
    root_c, root_s = (hi + lo) / 2.0, float((hi - lo).max())
    e21 = root_s / 2.0 ** 21
    j = int(np.argmin(np.abs(gas["x"] - 20.0) + np.abs(gas["y"]) + np.abs(gas["z"])))      # somewhere in the bulk
    # the level-21 cell of j: replay the reference's splits
    c, sz = root_c.copy(), root_s
    for _ in range(21):
        up = pos[:, j] > c
        c = c + np.where(up, 0.25 * sz, -0.25 * sz)
        sz *= 0.5
    off = pos[0, j] - c[0]                               # j inside its cell along x, in (-e21/2, e21/2]
    twin_dx = 0.05 * e21 * (-1.0 if off > 0 else 1.0)    # the twin stays in the same level-21 cell
    hj = float(gas["h"][j])
    sign = -1.0 if off > 0 else 1.0                      # probe on the side where the level-21 box reaches farther than the tiny one
    room = e21 / 2.0 + abs(off)                          # how far that side of the level-21 box lies beyond the tiny one
    probe_x = pos[0, j] + sign * (2.0 * hj + 0.2 * e21 + 0.25 * room)
    add = {k: np.array([gas[k][j], gas[k][j]]) for k in gas}
    add["x"] = np.array([pos[0, j] + twin_dx, probe_x])
    add["h"] = np.array([hj, 1.3 * hj])
    g2 = {k: np.concatenate([gas[k], add[k]]) for k in gas}
    n = g2["x"].size
    assert g2["x"].min() >= lo[0] and g2["x"].max() <= hi[0]              # the root box is unchanged
    ctx = capi.Context(device=0, variable=True)
    ctx.upload(g2); ctx.set_sinks(sinks)
    o = orc_v.OracleV(g2, sinks, nthreads=orc.max_threads())
    ctx.density(); ctx.forces(); o.evaluate()
    # the oracle gave both twins leaves far smaller than a level-21 cell, and the probe does not see them
    assert abs(o.ls[j]) <= e21 / 4 and abs(o.ls[n - 2]) <= e21 / 4
    for f in ("rho", "omega", "ax", "ay", "az", "du"):
        assert rel_err(ctx.field(f), getattr(o, f)) <= TOL, f
    assert abs(ctx.field("rho")[n - 1] - o.rho[n - 1]) <= 1e-13 * o.rho[n - 1]
    ctx.close()


def test_reflagged_list_vs_oracle(capi):
    """after calc_smoothing only h is new: the list of the new lengths is derived in place from the list of the old ones
    (nlist_v_reflag) once h has settled -- the evaluation on it against the CPU oracle, which searches afresh"""
    from oracle import orc, orc_v
    rows = ic.keplerian_disc_var(20000, seed=33)
    gas, sinks = ic.split_rows(rows)
    ctx = capi.Context(device=0, variable=True)
    ctx.upload(gas); ctx.set_sinks(sinks)
    o = orc_v.OracleV(gas, sinks, nthreads=orc.max_threads())
    # the IC's guess of h relaxes: the first updates change it by more than the margin.  A re-flagged list has lost its
    # margin shell, so without a drift in between the evaluations alternate between building and re-flagging: stop after
    # one that built, the next one then re-flags
    for it in range(10):
        r0 = ctx.stats().nlist_reflags
        ctx.density(); ctx.forces()
        built = ctx.stats().nlist_reflags == r0
        ctx.update_h()
        o.evaluate(); o.update_h()
        if built and it >= 4:
            break
    before = ctx.stats().nlist_reflags
    ctx.density(); ctx.forces(); o.evaluate()
    assert ctx.stats().nlist_reflags == before + 1
    assert rel_err(ctx.field("h"), o.h) <= 1e-12
    for f in ("rho", "omega", "ax", "ay", "az", "du", "dalpha"):
        assert rel_err(ctx.field(f), getattr(o, f)) <= TOL, f
    ctx.close()


def test_reflag_equals_build_over_a_trajectory(capi):
    """12 steps of a 60k disc with and without the re-flag pass: identical dt decisions, the same state to rounding (the
    neighbour SETS are the same, the order of a few list entries is not), and the pass really ran"""
    rows = ic.keplerian_disc_var(60000, seed=34)
    gas, sinks = ic.split_rows(rows)
    out = {}
    for tag, flags in (("build", capi.FLAG_NO_REFLAG), ("reflag", 0)):
        ctx = capi.Context(device=0, variable=True, flags=capi.FLAG_VARIABLE_H | flags)
        ctx.upload(gas); ctx.set_sinks(sinks)
        dts, t = [1e-2], 0.0
        for _ in range(12):
            dt, t = ctx.run(1, dts[-1], t)
            dts.append(dt)
        st = ctx.stats()
        out[tag] = dict(dts=dts, reflags=st.nlist_reflags, builds=st.nlist_builds, **{f: ctx.field(f) for f in "x y z vx vy vz u alpha h rho".split()})
        ctx.close()
    assert out["build"]["reflags"] == 0 and out["reflag"]["reflags"] >= 4
    assert out["reflag"]["builds"] + out["reflag"]["reflags"] == out["build"]["builds"]
    assert out["build"]["dts"] == out["reflag"]["dts"]
    for f in "x y z vx vy vz u alpha h rho".split():
        assert rel_err(out["reflag"][f], out["build"][f]) <= 1e-13, f


def test_full_simulate_trajectory_vs_reference_fixture(capi):
    """simulate() of the variable-h reference as it is: find_forces with the gas self-gravity (softening looked up
    with the particle's own h), calc_smoothing, sink accretion ([V]'s L1-distance rule) and the boundary cull"""
    g = load_golden("discv3000_traj")
    ctx, gas, sinks = make_ctx(capi, g, flags=capi.FLAG_VARIABLE_H | capi.FLAG_SELF_GRAVITY | capi.FLAG_ACCRETE_CULL)
    dts, t = [1e-2], 0.0
    for _ in range(5):
        dt, t = ctx.step(dts[-1], t)
        dts.append(dt)
    assert dts == list(g["full_dt_seq"])
    assert ctx.n == int(g["full_n_seq"][-1])
    for f in "x y z vx vy vz u alpha h".split():
        assert rel_err(ctx.field(f), g["full_s5_" + f]) <= 1e-10, f
    ctx.close()


def test_sink_creation_vs_reference_fixture(capi):
    """check_sink_creation ([V]:549-597) inside the loop: a very massive particle outside the disc becomes the seed of a
    second sink in step 1 (mass 1e-11, radius 2h), which accretes its seed in the same step -- as the real reference does"""
    g = load_golden("sinkcv1500_traj")
    ctx, gas, sinks = make_ctx(capi, g, flags=capi.FLAG_VARIABLE_H | capi.FLAG_SELF_GRAVITY | capi.FLAG_ACCRETE_CULL
                               | capi.FLAG_SINK_CREATION)
    dts, t, ns = [1e-2], 0.0, [ctx.n]
    for k in range(1, 4):
        dt, t = ctx.step(dts[-1], t)
        dts.append(dt); ns.append(ctx.n)
        p = f"full_s{k}_"
        s = ctx.get_sinks()
        assert s["m"].size == g[p + "sm"].size == 2
        assert np.max(np.abs(s["m"] - g[p + "sm"]) / g[p + "sm"]) <= 1e-14
        assert np.max(np.abs(s["radius"] - g[p + "srad"])) <= 1e-10
        assert np.max(np.abs(s["x"] - g[p + "sx"])) <= 1e-9 and np.max(np.abs(s["vy"] - g[p + "svy"])) <= 1e-9
        assert ctx.n == g[p + "x"].size
        for f in "x y z vx vy vz u alpha h".split():
            assert rel_err(ctx.field(f), g[p + f]) <= 1e-9, (k, f)
    assert ns == [int(v) for v in g["full_n_seq"]] and ns[1] == 1499
    assert dts == list(g["full_dt_seq"])
    ctx.close()
    # without the flag nothing is created
    ctx, gas, sinks = make_ctx(capi, g, flags=capi.FLAG_VARIABLE_H | capi.FLAG_SELF_GRAVITY | capi.FLAG_ACCRETE_CULL)
    ctx.step(1e-2, 0.0)
    assert ctx.get_sinks()["m"].size == 1
    ctx.close()


def test_sinks_outside_the_box_are_culled(capi):
    """[V]'s check_bounds packs the sinks too ([V]:610-613): the far sink of the file pulls on the gas during step 1 and
    is gone afterwards -- 3 steps against the real reference"""
    g = load_golden("sinkcullv1000_traj")
    ctx, gas, sinks = make_ctx(capi, g, flags=capi.FLAG_VARIABLE_H | capi.FLAG_SELF_GRAVITY | capi.FLAG_ACCRETE_CULL
                               | capi.FLAG_SINK_CREATION)
    assert ctx.get_sinks()["m"].size == 2
    dts, t = [1e-2], 0.0
    for k in range(1, 4):
        dt, t = ctx.step(dts[-1], t)
        dts.append(dt)
        if k in (1, 3):
            p = f"full_s{k}_"
            s = ctx.get_sinks()
            assert s["m"].size == 1 and s["m"][0] == g[p + "sm"][0]
            assert abs(s["x"][0] - g[p + "sx"][0]) <= 1e-12
            for f in "x y z vx vy vz u alpha h".split():
                assert rel_err(ctx.field(f), g[p + f]) <= 1e-10, (k, f)
    assert dts == list(g["full_dt_seq"])
    ctx.close()


def test_upload_of_h_after_an_evaluation_builds_the_list_anew(capi):
    """density, upload_field(h * 1.5), density.  The short path (only h is newer than the grid) must not
    measure the growth of h against h_new -- which holds build lengths only after sph_update_h -- and re-flag a list whose
    margin shell does not cover the uploaded lengths.  Checked against the CPU oracle evaluating the same h, once right
    after an upload (h_new never written) and once after an update_h (h_new = some older lengths)."""
    from oracle import orc, orc_v
    rows = ic.keplerian_disc_var(6000, seed=41)
    gas, sinks = ic.split_rows(rows)
    ctx = capi.Context(device=0, variable=True)
    ctx.upload(gas); ctx.set_sinks(sinks)
    ctx.density(); ctx.forces()
    for round_ in range(2):
        builds0, reflags0 = ctx.stats().nlist_builds, ctx.stats().nlist_reflags
        h15 = ctx.field("h") * 1.5
        ctx.upload_field("h", h15)
        ctx.density(); ctx.forces()
        st = ctx.stats()
        assert st.nlist_builds == builds0 + 1 and st.nlist_reflags == reflags0      # built, not re-flagged
        g2 = dict(gas); g2["h"] = h15
        for k in "x y z vx vy vz u alpha".split():
            g2[k] = ctx.field(k)
        o = orc_v.OracleV(g2, sinks, nthreads=orc.max_threads())
        o.evaluate()
        for f in ("rho", "omega", "ax", "ay", "az", "du", "dalpha"):
            assert rel_err(ctx.field(f), getattr(o, f)) <= TOL, (round_, f)
        # second round: h_new now holds the lengths of an update_h two lists ago
        ctx.update_h(); ctx.density(); ctx.forces()
    ctx.close()
