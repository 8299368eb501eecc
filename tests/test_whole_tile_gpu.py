"""The whole-tile pair kernels (tiled.hip: neighbours from one LDS tile per workgroup, chosen per list build and per
workgroup) against the direct-gather kernels of pairs.hip (SPH_FLAG_NO_WHOLE_TILE).  density_wt performs the same
operations in the same order: bitwise.  The forces kernel (forces_q) gives every target to four lanes, each with a
quarter of the list, and adds the four partial sums in a fixed tree: every pair term is bitwise the same, the sums differ
by summation order (<= 1e-14 of the field's scale, and reproducible).  Parity with the reference is what
test_parity_gpu.py checks (it runs whichever kernels the context picks)."""
import numpy as np
import pytest

from summersph_amd import ic

pytestmark = pytest.mark.gpu
FIELDS = "rho P c ax ay az du dalpha".split()
BITWISE = ("rho", "P", "c")


def same(a, b, f, tol=1e-14):
    if f in BITWISE:
        return np.array_equal(a, b)
    return float(np.max(np.abs(a - b))) <= tol * float(np.max(np.abs(b)))


@pytest.fixture(scope="module")
def capi():
    from summersph_amd import capi as m
    m.load()
    return m


def evaluate(capi, gas, sinks, flags, steps=0):
    ctx = capi.Context(device=0, flags=flags)
    ctx.upload(gas); ctx.set_sinks(sinks)
    if steps:
        ctx.run(steps, 1e-2, 0.0)
        out = {f: ctx.field(f) for f in "x y z vx vy vz u alpha".split()}
    else:
        ctx.density(); ctx.forces()
        out = {f: ctx.field(f) for f in FIELDS}
    st = ctx.stats()
    ctx.close()
    return out, st


def stirred_disc(n, seed=202):
    gas, sinks = ic.split_rows(ic.keplerian_disc(n, seed=seed, nngb=85.0))
    rng = np.random.default_rng(3)
    gas["vx"] = gas["vx"] + rng.normal(0.0, 0.05, n)          # viscosity switches on
    gas["alpha"] = np.full(n, 0.3)
    return gas, sinks


def test_reference_fixtures_are_evaluated_by_the_whole_tile_kernels(capi):
    """the disc fixtures of the real reference run through density_wt / forces_wt in the default context: the parity
    statement of test_parity_gpu.py is about these kernels"""
    from conftest import load_golden, rel_err
    for name in ("disc3000_eval", "disc3000ns_eval"):
        g = load_golden(name)
        gas, sinks = ic.split_rows(g["ic"])
        ctx = capi.Context(device=0)
        ctx.upload(gas); ctx.set_sinks(sinks)
        ctx.density(); ctx.forces()
        assert ctx.stats().tile_fit_pct >= 90
        for f in ("rho", "P", "c"):
            assert rel_err(ctx.field(f), g[f]) <= 1e-13, (name, f)
        for f in ("ax", "ay", "az", "du", "dalpha"):
            assert rel_err(ctx.field(f), g["sph_" + f]) <= 1e-13, (name, f)
        ctx.close()


def test_thin_disc_runs_from_the_tile_and_matches_bitwise(capi):
    gas, sinks = stirred_disc(200_000)
    a, sa = evaluate(capi, gas, sinks, 0)
    b, sb = evaluate(capi, gas, sinks, capi.FLAG_NO_WHOLE_TILE)
    assert sa.tile_fit_pct >= 90 and sa.tile_fit_pct_forces >= 90 and sb.tile_fit_pct == -1
    for f in FIELDS:
        assert same(a[f], b[f], f), f
    a2, _ = evaluate(capi, gas, sinks, 0)                      # reproducible: a second context gives the same bits
    for f in FIELDS:
        assert np.array_equal(a[f], a2[f]), f


def test_ragged_size_and_trajectory(capi):
    gas, sinks = stirred_disc(70_001, seed=5)                  # last workgroup partly filled, last wave ragged
    a, sa = evaluate(capi, gas, sinks, 0, steps=3)
    b, _ = evaluate(capi, gas, sinks, capi.FLAG_NO_WHOLE_TILE, steps=3)
    assert sa.tile_fit_pct >= 90
    for f in a:
        assert float(np.max(np.abs(a[f] - b[f]))) <= 1e-12 * float(np.max(np.abs(b[f]))), f


def test_thick_domain_keeps_the_gather_kernels(capi):
    # a cube 36 cells high: the three intervals of a workgroup span whole columns, nothing fits (not even the table-free tile)
    rng = np.random.default_rng(8)
    n = 450_000
    gas = {k: np.zeros(n) for k in "x y z vx vy vz u m alpha".split()}
    for k in "xyz":
        gas[k] = rng.uniform(0.0, 180.0, n)
    gas["u"][:] = 0.25; gas["m"][:] = 1e-4; gas["alpha"][:] = 0.1
    gas["vx"] = rng.normal(0.0, 0.05, n)
    sinks = {k: np.zeros(0) for k in "x y z vx vy vz m".split()}
    a, sa = evaluate(capi, gas, sinks, 0)
    b, _ = evaluate(capi, gas, sinks, capi.FLAG_NO_WHOLE_TILE)
    assert 0 <= sa.tile_fit_pct < 90
    for f in FIELDS:
        assert same(a[f], b[f], f), f


def test_workgroups_that_do_not_fit_fall_back_inside_the_kernel(capi):
    # thin disc plus a thick blob in a corner of it: a few per cent of the workgroups exceed the tile and take the
    # gather loop inside the whole-tile kernels, the rest read LDS -- in the same launch
    gas, sinks = stirred_disc(200_000, seed=9)
    rng = np.random.default_rng(10)
    nb = 12_000
    blob = {k: np.zeros(nb) for k in gas if isinstance(gas[k], np.ndarray)}
    r_out = float(np.max(np.hypot(gas["x"], gas["y"])))
    blob["x"] = 0.5 * r_out + rng.uniform(-20.0, 20.0, nb)
    blob["y"] = rng.uniform(-20.0, 20.0, nb)
    blob["z"] = rng.uniform(-20.0, 20.0, nb)
    blob["u"][:] = 0.25; blob["m"][:] = gas["m"][0]; blob["alpha"][:] = 0.3
    both = {k: np.concatenate([gas[k], blob[k]]) for k in blob}
    a, sa = evaluate(capi, both, sinks, 0)
    b, _ = evaluate(capi, both, sinks, capi.FLAG_NO_WHOLE_TILE)
    assert 90 <= sa.tile_fit_pct < 100
    for f in FIELDS:
        assert same(a[f], b[f], f), f


def test_counting_sort_gives_the_order_of_the_stable_radix_sort(tmp_path):
    """grid.hip sorts by cell with a counting sort (histogram, scan, scatter, per-cell rank); SPH_SORT_RADIX=1 (read once per
    process) restores rocprim's stable radix sort.  Same permutation, hence bitwise the same trajectory."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, numpy as np\n"
        f"sys.path.insert(0, {root!r})\n"
        "from summersph_amd import capi, ic\n"
        "gas, sinks = ic.split_rows(ic.keplerian_disc(60_000, seed=17, nngb=85.0))\n"
        "ctx = capi.Context(device=0)\n"
        "ctx.upload(gas); ctx.set_sinks(sinks)\n"
        "ctx.run(4, 1e-2, 0.0)\n"
        "np.savez(sys.argv[1], **{f: ctx.field(f) for f in 'x y z vx vy vz u alpha'.split()})\n"
    )
    out = {}
    for tag, env in (("count", {}), ("radix", {"SPH_SORT_RADIX": "1"})):
        path = tmp_path / f"{tag}.npz"
        subprocess.run([sys.executable, "-c", code, str(path)], check=True, env={**os.environ, **env}, timeout=300)
        out[tag] = dict(np.load(path))
    for f in out["count"]:
        assert np.array_equal(out["count"][f], out["radix"][f]), f


def test_table_free_tile_kernels_are_bitwise_the_table_kernels(tmp_path):
    """dense neighbourhoods get the tile kernels with the kernel table's knots RECOMPUTED (w_knot / dw_knot: the table's 40 KB
    go to the tile).  SPH_TILE_TABLE=regs (read once per process) forces that variant on an ordinary disc: every field must
    equal the table variant's bit for bit -- the knots are the table's values, the operations and their order the same."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, numpy as np\n"
        f"sys.path.insert(0, {root!r})\n"
        "from summersph_amd import capi, ic\n"
        "gas, sinks = ic.split_rows(ic.keplerian_disc(120_000, seed=19, nngb=85.0))\n"
        "rng = np.random.default_rng(4); gas['vx'] = gas['vx'] + rng.normal(0.0, 0.05, gas['x'].size); gas['alpha'] = np.full(gas['x'].size, 0.3)\n"
        "ctx = capi.Context(device=0)\n"
        "ctx.upload(gas); ctx.set_sinks(sinks)\n"
        "ctx.density(); ctx.forces()\n"
        "st = ctx.stats()\n"
        "assert st.tile_fit_pct >= 90 and st.tile_fit_pct_forces >= 90\n"
        "ev = {f: ctx.field(f) for f in 'rho P c ax ay az du dalpha'.split()}\n"
        "ctx.run(3, 1e-2, 0.0)\n"
        "np.savez(sys.argv[1], **ev, **{'t_' + f: ctx.field(f) for f in 'x vx u alpha'.split()})\n"
    )
    out = {}
    # SPH_TILE_MIN_GROUPS_D=0: the tile kernel also for the density pass of this small set (fewer groups than CUs)
    for tag, env in (("table", {"SPH_TILE_MIN_GROUPS_D": "0"}), ("regs", {"SPH_TILE_TABLE": "regs", "SPH_TILE_MIN_GROUPS_D": "0"})):
        path = tmp_path / f"{tag}.npz"
        subprocess.run([sys.executable, "-c", code, str(path)], check=True, env={**os.environ, **env}, timeout=300)
        out[tag] = dict(np.load(path))
    for f in out["table"]:
        assert np.array_equal(out["table"][f], out["regs"][f]), f


def test_dense_disc_runs_from_the_table_free_tiles(capi):
    """~200 neighbours (the survey's anchor regime): the three intervals of a group no longer fit beside the 40-KB table, but
    they fit the table-free tile -- the context must pick it, and agree with the direct-gather kernels as the table variant does"""
    gas, sinks = ic.split_rows(ic.keplerian_disc(100_000, seed=212, nngb=340.0))
    rng = np.random.default_rng(3)
    gas["vx"] = gas["vx"] + rng.normal(0.0, 0.05, gas["x"].size)
    gas["alpha"] = np.full(gas["x"].size, 0.3)
    a, sa = evaluate(capi, gas, sinks, 0)
    b, _ = evaluate(capi, gas, sinks, capi.FLAG_NO_WHOLE_TILE)
    assert sa.nlist_mean > 150
    assert sa.tile_fit_pct >= 90, (sa.tile_fit_pct, sa.tile_fit_pct_forces)      # the density geometry fits the table-free tile
    for f in FIELDS:
        assert same(a[f], b[f], f), f


def test_half_group_forces_kernel_agrees(tmp_path):
    """forces_q on groups of 128 targets (eight lanes per target, a list row per trip; what dense neighbourhoods get when
    the tile of a group of 256 does not fit): forced with SPH_FORCES_HALF_GROUPS=1 on an ordinary stirred disc, it must give the
    default kernel's forces up to the order in which a target's partial sums are added (<= 1e-14 of the field's scale),
    with the table and table-free tiles alike, and the same 3-step trajectory to rounding"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, numpy as np\n"
        f"sys.path.insert(0, {root!r})\n"
        "from summersph_amd import capi, ic\n"
        "gas, sinks = ic.split_rows(ic.keplerian_disc(90_001, seed=23, nngb=85.0))\n"
        "rng = np.random.default_rng(4); gas['vx'] = gas['vx'] + rng.normal(0.0, 0.05, gas['x'].size); gas['alpha'] = np.full(gas['x'].size, 0.3)\n"
        "ctx = capi.Context(device=0)\n"
        "ctx.upload(gas); ctx.set_sinks(sinks)\n"
        "ctx.density(); ctx.forces()\n"
        "assert ctx.stats().tile_fit_pct_forces >= 90\n"
        "ev = {f: ctx.field(f) for f in 'rho ax ay az du dalpha'.split()}\n"
        "ctx.run(3, 1e-2, 0.0)\n"
        "np.savez(sys.argv[1], **ev, **{'t_' + f: ctx.field(f) for f in 'x vx u alpha'.split()})\n"
    )
    out = {}
    for tag, env in (("default", {}), ("half", {"SPH_FORCES_HALF_GROUPS": "1"}), ("half_regs", {"SPH_FORCES_HALF_GROUPS": "1", "SPH_TILE_TABLE": "regs"})):
        path = tmp_path / f"{tag}.npz"
        subprocess.run([sys.executable, "-c", code, str(path)], check=True, env={**os.environ, **env}, timeout=300)
        out[tag] = dict(np.load(path))
    for tag in ("half", "half_regs"):
        assert np.array_equal(out[tag]["rho"], out["default"]["rho"])
        for f in "ax ay az du dalpha".split():
            assert same(out[tag][f], out["default"][f], f), (tag, f)
        for f in "t_x t_vx t_u t_alpha".split():
            assert float(np.max(np.abs(out[tag][f] - out["default"][f]))) <= 1e-12 * float(np.max(np.abs(out["default"][f]))), (tag, f)
