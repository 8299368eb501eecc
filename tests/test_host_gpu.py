"""The thin Fortran host (summersph_amd/host) end to end on the GPU: ingest of the reference's text
format, the simulate() loop through ISO_C_BINDING, snapshot output -- checked against the
trajectory fixtures dumped from the real reference."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, load_golden, rel_err
from summersph_amd import txtio

HOST_DIR = os.path.join(ROOT, "summersph_amd", "host")
HOST_BIN = os.path.join(HOST_DIR, "run_sph_hip")
HOST_BIN_V = os.path.join(HOST_DIR, "run_sph_hip_v")


def _build(which=HOST_BIN):
    if not os.path.exists(which):
        subprocess.run(["make", "-C", HOST_DIR], check=True, stdout=subprocess.DEVNULL)
    return which


def test_host_builds_with_amdflang():
    """CPU: both Fortran hosts compile and link against the C ABI"""
    assert os.path.exists(_build()) and os.path.exists(_build(HOST_BIN_V))


@pytest.mark.gpu
@pytest.mark.parametrize("name,variant,nsteps", [("sod1000_traj", "sph", 5), ("disc3000_traj", "sph", 5),
                                                 ("disc3000_traj", "full", 5), ("acc2000_traj", "full", 3),
                                                 ("acc2000_traj", "full", 1)])      # the LAST step removes particles
def test_fortran_host_trajectory(tmp_path, name, variant, nsteps):
    """variant 'full' = the host's default: simulate() as the reference runs it (gas self-gravity, accretion,
    boundary cull); 'sph' leaves those three out"""
    g = load_golden(name)
    icf = tmp_path / "ic.txt"
    txtio.write_ic(str(icf), g["ic"])
    snap = tmp_path / "final.txt"
    cmd = [_build(), str(icf), str(nsteps), str(snap)] + (["sph"] if variant == "sph" else [])
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=tmp_path, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Successfully read" in r.stdout
    dts = [float(l.split()[2]) for l in r.stdout.splitlines() if l.startswith("dt ")]
    assert dts == list(g[variant + "_dt_seq"])[:nsteps + 1]       # identical dt decisions through the Fortran loop
    gas, sinks = txtio.read_snapshot(str(snap))
    p = f"{variant}_s{nsteps}_"
    assert gas.shape[0] == g[p + "x"].size             # accretion / cull followed by the host
    for col, f in enumerate("x y z vx vy vz u m alpha".split()):
        assert rel_err(gas[:, col], g[p + f]) <= 1e-11, f
    assert sinks.shape[0] == g[p + "sx"].size
    assert np.max(np.abs(sinks[:, 0] - g[p + "sx"])) <= 1e-11
    assert np.max(np.abs(sinks[:, 3] - g[p + "svx"])) <= 1e-11
    assert np.max(np.abs(sinks[:, 7] - g[p + "sm"])) <= 1e-14


@pytest.mark.gpu
def test_fortran_host_reader_conventions(tmp_path):
    """extra columns are ignored, a u == 0 row becomes a sink, rows come back in file order"""
    rng = np.random.default_rng(1)
    rows = np.zeros((50, 8))
    rows[:, :3] = rng.normal(0, 4, (50, 3)); rows[:, 3:6] = rng.normal(0, 1, (50, 3))
    rows[:, 6] = 1.0; rows[:, 7] = 1e-3
    rows[20, 6] = 0.0; rows[20, 7] = 1.0           # a sink in the middle of the file
    icf = tmp_path / "ic.txt"
    with open(icf, "w") as f:
        f.write("x y z vx vy vz u m extra1 extra2\n")
        for r in rows:
            f.write(" ".join(f"{v:.17e}" for v in r) + " 9.0 8.0\n")
    snap = tmp_path / "final.txt"
    r = subprocess.run([_build(), str(icf), "0", str(snap)], capture_output=True, text=True, cwd=tmp_path, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    gas, sinks = txtio.read_snapshot(str(snap))
    assert gas.shape == (49, 9) and sinks.shape == (1, 8)
    assert np.array_equal(gas[:, :8], np.delete(rows, 20, axis=0))
    assert np.array_equal(sinks[0, :6], rows[20, :6]) and sinks[0, 7] == 1.0


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["sph", "full"])
def test_fortran_host_variable_h_trajectory(tmp_path, variant):
    """the variable-h host (10-column ingest, parameters.txt, simulate with calc_smoothing) against the 5-step
    trajectories of the real "SUMMER_SPH - Variable.f90"; 'full' = its simulate() as it is"""
    g = load_golden("discv3000_traj")
    gamma, eta, tol, maxlen, scale = (float(v) for v in g["params"])
    icf = tmp_path / "ic10.txt"
    txtio.write_ic(str(icf), g["ic"], header="x y z vx vy vz energy mass alpha smoothing")
    pf = tmp_path / "parameters.txt"
    pf.write_text("bounding_size max_depth theta gamma eta convergence_criteria max_length timestep_scale end_time\n"
                  f"1500.0 1000 0.5 {gamma!r} {eta!r} {tol!r} {maxlen!r} {scale!r} 1000.0\n")
    snap = tmp_path / "final.txt"
    cmd = [_build(HOST_BIN_V), str(icf), str(pf), "5", str(snap)] + (["sph"] if variant == "sph" else [])
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=tmp_path, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Successfully read parameters" in r.stdout and "Successfully read" in r.stdout
    dts = [float(l.split()[2]) for l in r.stdout.splitlines() if l.startswith("dt ")]
    assert dts == list(g[variant + "_dt_seq"])
    gas, sinks = txtio.read_snapshot_v(str(snap))      # 10 columns for gas, 8 for the sink row
    p = f"{variant}_s5_"
    assert gas.shape[0] == g[p + "x"].size and sinks.shape[0] == g[p + "sx"].size
    for col, f in enumerate("x y z vx vy vz u m alpha h".split()):
        assert rel_err(gas[:, col], g[p + f]) <= 1e-10, f
    assert np.max(np.abs(sinks[:, 0] - g[p + "sx"])) <= 1e-11 and np.max(np.abs(sinks[:, 7] - g[p + "sm"])) <= 1e-14


@pytest.mark.gpu
def test_fortran_host_variable_h_sink_creation(tmp_path):
    """the variable-h host follows check_sink_creation: a second sink appears (and eats its seed), the snapshot holds
    two sink rows -- against the real reference's loop"""
    g = load_golden("sinkcv1500_traj")
    icf = tmp_path / "ic10.txt"
    txtio.write_ic(str(icf), g["ic"], header="x y z vx vy vz energy mass alpha smoothing")
    snap = tmp_path / "final.txt"
    r = subprocess.run([_build(HOST_BIN_V), str(icf), "-", "3", str(snap)], capture_output=True, text=True, cwd=tmp_path, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    dts = [float(l.split()[2]) for l in r.stdout.splitlines() if l.startswith("dt ")]
    assert dts == list(g["full_dt_seq"])
    gas, sinks = txtio.read_snapshot_v(str(snap))
    assert gas.shape[0] == 1499 and sinks.shape[0] == 2
    assert np.max(np.abs(sinks[:, 7] - g["full_s3_sm"]) / g["full_s3_sm"]) <= 1e-14
    assert np.max(np.abs(sinks[:, 0] - g["full_s3_sx"])) <= 1e-9
    for col, f in enumerate("x y z vx vy vz u m alpha h".split()):
        assert rel_err(gas[:, col], g["full_s3_" + f]) <= 1e-9, f


@pytest.mark.gpu
def test_fortran_host_periodic_saves(tmp_path):
    """the save cadence of simulate(): save0 on the first iteration, then one save per iteration once
    t > k * end_time / 1000 (SUMMER_SPH.f90:881-884); the saves hold the state at the START of their iteration"""
    g = load_golden("disc3000_traj")
    icf = tmp_path / "ic.txt"
    txtio.write_ic(str(icf), g["ic"])
    snap = tmp_path / "final.txt"
    # end time 0.04: dt = 0.01, 0.015, 0.0225 -> t = 0, 0.01, 0.025, 0.0475: three iterations, three saves
    r = subprocess.run([_build(), str(icf), "100", str(snap), "sph", "saves", "tend=0.04"], capture_output=True, text=True, cwd=tmp_path, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    dts = [float(l.split()[2]) for l in r.stdout.splitlines() if l.startswith("dt ")]
    assert dts == list(g["sph_dt_seq"])[:4]
    saves = sorted(p.name for p in tmp_path.glob("save*.txt"))
    assert saves == ["save0.txt", "save1.txt", "save2.txt"]
    gas0, sinks0 = txtio.read_snapshot(str(tmp_path / "save0.txt"))
    ic_gas = g["ic"][g["ic"][:, 6] != 0.0]
    assert np.array_equal(gas0[:, :8], ic_gas) and sinks0.shape[0] == 1
    gas1, _ = txtio.read_snapshot(str(tmp_path / "save1.txt"))
    assert rel_err(gas1[:, 0], g["sph_s1_x"]) <= 1e-12 and rel_err(gas1[:, 6], g["sph_s1_u"]) <= 1e-12
    # a save is a valid input again (alpha restarts at 0 as in the reference)
    back = tmp_path / "back.txt"
    r = subprocess.run([_build(), str(tmp_path / "save2.txt"), "-1", str(back)], capture_output=True, text=True, cwd=tmp_path, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    gas2, _ = txtio.read_snapshot(str(tmp_path / "save2.txt"))
    gasb, _ = txtio.read_snapshot(str(back))
    assert np.array_equal(gas2[:, :8], gasb[:, :8])
