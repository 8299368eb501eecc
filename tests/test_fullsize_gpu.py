"""The hot path at BASELINE.json's full size (1e6-particle disc, the bench workload) through properties that do not
need an oracle run of that size:

  * an independent brute-force density for a random subset (neighbours from a k-d tree, the reference's table
    look-up restated in numpy)                                                       -- [F]:105-127, 440-457
  * Newton's third law: sum_i m_i a_i + sum_s m_s a_s = 0 (pair terms are antisymmetric, gas <-> sink gravity too)
  * energy: sum_i m_i (v_i . a_i^SPH + du_i) = 0 (pressure and viscosity terms)      -- [F]:381-390
  * invariance under a permutation of the input order (only the summation order may change)
  * fused sph_run == the unfused call sequence, bitwise
The same for the variable-h path (momentum only: its pair rule is asymmetric in selection but every selected pair acts
on both partners, [V]:394-428)."""
import numpy as np
import pytest
from scipy.spatial import cKDTree

from summersph_amd import ic

pytestmark = pytest.mark.gpu
N = 1_000_000
H = 2.5


@pytest.fixture(scope="module")
def capi():
    from summersph_amd import capi as m
    m.load()
    return m


@pytest.fixture(scope="module")
def disc():
    gas, sinks = ic.split_rows(ic.keplerian_disc(N, seed=202, nngb=85.0))     # the bench workload
    rng = np.random.default_rng(11)
    gas["vx"] = gas["vx"] + rng.normal(0.0, 0.05, N)        # some velocity dispersion: viscosity switches on
    gas["vz"] = gas["vz"] + rng.normal(0.0, 0.05, N)
    gas["alpha"] = np.full(N, 0.3)
    return gas, sinks


@pytest.fixture(scope="module")
def evaluated(capi, disc):
    gas, sinks = disc
    ctx = capi.Context(device=0)
    ctx.upload(gas); ctx.set_sinks(sinks)
    ctx.density(); ctx.forces()
    out = {f: ctx.field(f) for f in "rho P c ax ay az du dalpha".split()}
    out["sink"] = ctx.get_sinks()
    out["G"] = ctx.params.G
    ctx.close()
    return out


def _tables(nq=5000):
    q = np.arange(nq + 1) * (2.0 / nq)
    return np.where(q <= 1.0, 1.0 - 1.5 * q ** 2 + 0.75 * q ** 3, 0.25 * (2.0 - q) ** 3)        # [F]:62-75


def test_density_subset_vs_brute_force(disc, evaluated):
    gas, _ = disc
    pos = np.stack([gas["x"], gas["y"], gas["z"]], axis=1)
    tree = cKDTree(pos)
    pick = np.random.default_rng(5).choice(N, 1500, replace=False)
    w, nq = _tables(), 5000
    dq = 2.0 / nq
    worst = 0.0
    for i in pick:
        nb = np.array(tree.query_ball_point(pos[i], 2.0 * H * (1.0 + 1e-12)))
        r = np.sqrt(((pos[nb] - pos[i]) ** 2).sum(1))
        qi = r / H
        ok = qi <= 2.0
        k = np.minimum((qi[ok] / dq).astype(np.int64), nq - 1)
        a = (qi[ok] - k * dq) / dq
        wl = ((1.0 - a) * w[k] + a * w[k + 1]) / (3.14159265359 * H ** 3)                     # [F]:114-125
        rho = np.sum(np.sort(gas["m"][nb][ok] * wl))
        worst = max(worst, abs(rho - evaluated["rho"][i]) / rho)
    assert worst <= 1e-13


def test_newtons_third_law_and_energy(disc, evaluated):
    gas, sinks = disc
    e = evaluated
    m = gas["m"]
    a = np.stack([e["ax"], e["ay"], e["az"]])
    s = e["sink"]
    tot = (m * a).sum(1) + np.array([np.sum(sinks["m"] * s[k]) for k in ("ax", "ay", "az")])
    scale = np.abs(m * a).sum(1).max()
    assert np.max(np.abs(tot)) <= 1e-12 * scale            # ~1e6 rounding errors of relative size 1e-16
    # SPH part of the acceleration = total - sink gravity ([F]:567-576: a_j -= m_s G (x_j - x_s)/|x_j - x_s|^3)
    d = np.stack([gas["x"] - sinks["x"][0], gas["y"] - sinks["y"][0], gas["z"] - sinks["z"][0]])
    ag = -sinks["m"][0] * e["G"] * d / np.sqrt((d ** 2).sum(0)) ** 3
    v = np.stack([gas["vx"], gas["vy"], gas["vz"]])
    work = m * ((v * (a - ag)).sum(0) + e["du"])
    assert abs(work.sum()) <= 1e-11 * np.abs(work).sum()
    assert np.all(e["dalpha"] > -1.0) and np.all(np.isfinite(e["du"]))


def test_permutation_invariance(capi, disc, evaluated):
    gas, sinks = disc
    perm = np.random.default_rng(6).permutation(N)
    g2 = {k: np.ascontiguousarray(v[perm]) for k, v in gas.items() if isinstance(v, np.ndarray) and v.shape == (N,)}
    ctx = capi.Context(device=0)
    ctx.upload(g2); ctx.set_sinks(sinks)
    ctx.density(); ctx.forces()
    for f in ("rho", "ax", "du", "dalpha"):
        got = ctx.field(f)
        ref = evaluated[f][perm]
        assert np.max(np.abs(got - ref)) <= 1e-13 * np.max(np.abs(ref)), f
    ctx.close()


def test_fused_run_equals_unfused_calls(capi, disc):
    gas, sinks = disc
    a = capi.Context(device=0); a.upload(gas); a.set_sinks(sinks)
    b = capi.Context(device=0); b.upload(gas); b.set_sinks(sinks)
    dt_a, t_a = a.run(2, 1e-2, 0.0)
    dt = 1e-2
    for _ in range(2):
        b.density(); b.forces(); b.kick(dt); b.drift(dt); b.density(); b.forces(); b.kick(dt)
        dt = b.next_dt(dt)
    assert dt == dt_a
    for f in "x y z vx vy vz u alpha".split():
        assert np.array_equal(a.field(f), b.field(f)), f
    st = a.stats()
    assert st.density_passes == 4 and st.force_passes == 4 and st.nlist_builds == 3       # upload + one per drift
    a.close(); b.close()


def test_variable_h_momentum_at_full_size(capi):
    gas, sinks = ic.split_rows(ic.keplerian_disc_var(N, seed=303))
    ctx = capi.Context(device=0, variable=True)
    ctx.upload(gas); ctx.upload_field("h", gas["h"]); ctx.set_sinks(sinks)
    ctx.density(); ctx.forces()
    m = gas["m"]
    a = np.stack([ctx.field("ax"), ctx.field("ay"), ctx.field("az")])
    s = ctx.get_sinks()
    tot = (m * a).sum(1) + np.array([np.sum(sinks["m"] * s[k]) for k in ("ax", "ay", "az")])
    assert np.max(np.abs(tot)) <= 1e-12 * np.abs(m * a).sum(1).max()
    om = ctx.field("omega")
    assert np.all(np.isfinite(om)) and np.all(ctx.field("rho") > 0.0)
    ctx.close()
