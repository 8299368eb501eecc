"""Pins the CPU restatement of the Barnes-Hut gas self-gravity, sink accretion and boundary cull
(oracle/sph_oracle_grav.c) against the "full_*" fixtures dumped from the real reference
(find_forces and simulate's loop body as they are).  CPU only."""
import numpy as np
import pytest

from conftest import load_golden, rel_err
from oracle import orc_grav
from summersph_amd import ic


def _oracle(g, nthreads=1):
    gas, sinks = ic.split_rows(g["ic"])
    return orc_grav.OracleFull(gas, sinks, nthreads=nthreads)


@pytest.mark.parametrize("name", ["sod1000_eval", "disc3000_eval", "disc3000ns_eval", "bin2000_eval"])
def test_find_forces_with_self_gravity(name):
    g = load_golden(name)
    o = _oracle(g)
    o.evaluate()
    # the BH term itself (difference full - sph) is tiny next to the sink's pull: check it on its own scale
    for k in "ax ay az".split():
        grav_ref = g["full_" + k] - g["sph_" + k]
        assert rel_err(getattr(o, k), g["full_" + k]) <= 1e-13, k
        assert np.max(np.abs(grav_ref)) > 0
    for k in "du dalpha".split():
        assert rel_err(getattr(o, k), g["full_" + k]) <= 1e-13, k
    assert o.next_dt(1e-2) == g["full_dt"][0]


def test_gravity_term_alone_matches_to_1e10():
    g = load_golden("disc3000_eval")
    gas, sinks = ic.split_rows(g["ic"])
    t = orc_grav.Tree(gas["x"], gas["y"], gas["z"], gas["m"])
    a = [np.zeros(3000) for _ in range(3)]
    orc_grav.gravity(t, gas["x"], gas["y"], gas["z"], *a)
    for k, arr in zip("ax ay az".split(), a):
        ref = g["full_" + k] - g["sph_" + k]          # cancellation leaves ~1e-13 of noise relative to |a|
        assert np.max(np.abs(arr - ref)) <= 1e-12 * np.max(np.abs(g["full_" + k])) , k
        assert np.max(np.abs(arr - ref)) <= 1e-7 * np.max(np.abs(ref)), k


@pytest.mark.parametrize("name", ["sod1000_traj", "disc3000_traj"])
def test_full_trajectory(name):
    g = load_golden(name)
    o = _oracle(g, nthreads=2)
    dts = [1e-2]
    for _ in range(5):
        dts.append(o.step(dts[-1]))
    assert dts == list(g["full_dt_seq"])
    assert o.n == int(g["full_n_seq"][-1])
    for f in "x y z vx vy vz u alpha".split():
        assert rel_err(getattr(o, f), g["full_s5_" + f]) <= 1e-11, f


def test_accretion_and_cull():
    g = load_golden("acc2000_traj")
    o = _oracle(g)
    dts, ns = [1e-2], [o.n]
    for k in range(1, 4):
        dts.append(o.step(dts[-1])); ns.append(o.n)
        p = f"full_s{k}_"
        assert o.n == g[p + "x"].size
        for f in "x y z vx vy vz u m alpha".split():
            assert rel_err(getattr(o, f), g[p + f]) <= 1e-11, (k, f)
        assert np.max(np.abs(o.sm - g[p + "sm"])) <= 1e-15 and np.max(np.abs(o.sx - g[p + "sx"])) <= 1e-12
    assert ns == [int(v) for v in g["full_n_seq"]] and ns[1] < ns[0]
    assert dts == list(g["full_dt_seq"])


def test_binary_two_sinks_full_loop():
    """two sinks (sink-sink forces, either may accrete): the reference's loop body for 3 steps, and its sph variant"""
    g = load_golden("bin2000_traj")
    o = _oracle(g)
    dts, ns = [1e-2], [o.n]
    for k in range(1, 4):
        dts.append(o.step(dts[-1])); ns.append(o.n)
        if k in (1, 3):
            p = f"full_s{k}_"
            assert o.n == g[p + "x"].size
            for f in "x y z vx vy vz u m alpha".split():
                assert rel_err(getattr(o, f), g[p + f]) <= 1e-11, (k, f)
            assert np.max(np.abs(o.sm - g[p + "sm"])) <= 1e-15 and np.max(np.abs(o.sx - g[p + "sx"])) <= 1e-12
            assert np.max(np.abs(o.svy - g[p + "svy"])) <= 1e-12
    assert ns == [int(v) for v in g["full_n_seq"]] and ns[-1] == 1997
    assert dts == list(g["full_dt_seq"])
