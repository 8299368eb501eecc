"""CPU tests of the Fortran hosts' ingest (no device: `max_steps < 0` is the drivers' ingest-only mode): records that
span physical lines, as the reference's list-directed read accepts them (SUMMER_SPH.f90:647, Variable.f90:782) and as
flang's list-directed saves produce them; extra columns; sink rows."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from summersph_amd import txtio

HOST_DIR = os.path.join(ROOT, "summersph_amd", "host")
FC = shutil.which("amdflang") or "/opt/rocm/bin/amdflang"


def _host(name):
    path = os.path.join(HOST_DIR, name)
    if not os.path.exists(path):
        subprocess.run(["make", "-C", HOST_DIR], check=True, stdout=subprocess.DEVNULL)
    return path


def _wrapper_tool(tmp_path):
    exe = tmp_path / "write_listdirected"
    subprocess.run([FC, "-O1", os.path.join(ROOT, "tests", "tools", "write_listdirected.f90"), "-o", str(exe)], check=True,
                   cwd=tmp_path, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return str(exe)


def _rows(ncol, n=40, seed=3):
    rng = np.random.default_rng(seed)
    rows = np.zeros((n, ncol))
    rows[:, :3] = rng.normal(0, 40, (n, 3)); rows[:, 3:6] = rng.normal(0, 1, (n, 3))
    rows[:, 6] = rng.uniform(0.1, 2.0, n); rows[:, 7] = 10.0 ** rng.uniform(-9, -3, n)
    if ncol > 8:
        rows[:, 8] = rng.uniform(0, 0.1, n)
    if ncol > 9:
        rows[:, 9] = rng.uniform(2.0, 6.0, n)
    rows[n // 2, 6] = 0.0; rows[n // 2, 7] = 1.0        # a sink in the middle
    rows[n - 1, 6] = 0.0; rows[n - 1, 7] = 0.5          # and one as the last record
    return rows


@pytest.mark.skipif(not os.path.exists(FC), reason="needs amdflang")
@pytest.mark.parametrize("variant", ["F", "V"])
def test_ingest_of_wrapped_list_directed_saves(tmp_path, variant):
    """a save written with list-directed output under flang (records wrapped at 80 columns over 2-4 lines) comes back
    record for record, sinks included"""
    ncol = 9 if variant == "F" else 10
    rows = _rows(ncol)
    flat = tmp_path / "flat.txt"
    with open(flat, "w") as f:
        f.write("x y z vx vy vz energy mass alpha smoothing\n")
        for r in rows:
            f.write(" ".join(f"{v:.17e}" for v in (r if r[6] != 0.0 else r[:8])) + "\n")
    wrapped = tmp_path / "wrapped.txt"
    subprocess.run([_wrapper_tool(tmp_path), str(flat), str(wrapped), str(ncol)], check=True)
    lines = open(wrapped).read().splitlines()
    assert len(lines) > 2 * len(rows)                   # the records really are wrapped
    out = tmp_path / "back.txt"
    if variant == "F":
        cmd = [_host("run_sph_hip"), str(wrapped), "-1", str(out)]
    else:
        cmd = [_host("run_sph_hip_v"), str(wrapped), "-", "-1", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=tmp_path, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Successfully read" in r.stdout
    gas, sinks = (txtio.read_snapshot if variant == "F" else txtio.read_snapshot_v)(str(out))
    is_sink = rows[:, 6] == 0.0
    assert gas.shape[0] == int((~is_sink).sum()) and sinks.shape[0] == 2
    want = rows[~is_sink]
    assert np.array_equal(gas[:, :8], want[:, :8])
    if variant == "F":
        assert np.all(gas[:, 8] == 0.0)                 # [F] resets alpha on ingest (SUMMER_SPH.f90:681)
    else:
        assert np.array_equal(gas[:, 8:10], want[:, 8:10])
    assert np.array_equal(sinks[:, :6], rows[is_sink][:, :6]) and np.array_equal(sinks[:, 7], rows[is_sink][:, 7])


@pytest.mark.parametrize("variant", ["F", "V"])
def test_ingest_one_record_per_line_with_extra_columns_and_blank_lines(tmp_path, variant):
    ncol = 8 if variant == "F" else 10
    rows = _rows(ncol, n=25, seed=5)
    icf = tmp_path / "ic.txt"
    with open(icf, "w") as f:
        f.write("header line\n")
        for k, r in enumerate(rows):
            if variant == "V" and r[6] == 0.0:
                f.write(" ".join(f"{v:.17e}" for v in r[:8]) + "\n")        # the sink rows [V] writes carry 8 values
            else:
                f.write(" ".join(f"{v:.17e}" for v in r) + " 9.0 8.0\n")
            if k % 7 == 3:
                f.write("\n")
    out = tmp_path / "back.txt"
    cmd = [_host("run_sph_hip"), str(icf), "-1", str(out)] if variant == "F" else [_host("run_sph_hip_v"), str(icf), "-", "-1", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=tmp_path, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    gas, sinks = (txtio.read_snapshot if variant == "F" else txtio.read_snapshot_v)(str(out))
    is_sink = rows[:, 6] == 0.0
    assert np.array_equal(gas[:, :8], rows[~is_sink][:, :8]) and sinks.shape[0] == 2
    if variant == "V":
        assert np.array_equal(gas[:, 8:10], rows[~is_sink][:, 8:10])


def test_ingest_of_saves_wrapped_four_values_per_line_with_zero_energy_rows(tmp_path):
    """framing must not depend on the values read: a save wrapped at four values per line puts the 8th value
    at a line end and alpha alone on the next line -- also for a 9-value row whose energy is 0 (a sink by the format's
    own rule, [F]:650, but written with an alpha).  Every later record must stay in frame."""
    rows = _rows(9, n=30, seed=11)
    nine_valued_zero = 7                                  # a 9-value record with u == 0 in front of ordinary gas rows
    rows[nine_valued_zero, 6] = 0.0
    wrapped = tmp_path / "wrapped4.txt"
    with open(wrapped, "w") as f:
        f.write("x y z vx vy vz energy mass alpha\n")
        for k, r in enumerate(rows):
            vals = r if (r[6] != 0.0 or k == nine_valued_zero) else r[:8]
            for a in range(0, len(vals), 4):
                f.write(" " + " ".join(f"{v:.17e}" for v in vals[a:a + 4]) + "\n")
    out = tmp_path / "back.txt"
    r = subprocess.run([_host("run_sph_hip"), str(wrapped), "-1", str(out)], capture_output=True, text=True, cwd=tmp_path, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    gas, sinks = txtio.read_snapshot(str(out))
    is_sink = rows[:, 6] == 0.0
    assert gas.shape[0] == int((~is_sink).sum()) and sinks.shape[0] == int(is_sink.sum()) == 3
    assert np.array_equal(gas[:, :8], rows[~is_sink][:, :8])
    assert np.array_equal(sinks[:, :6], rows[is_sink][:, :6]) and np.array_equal(sinks[:, 7], rows[is_sink][:, 7])


def test_ingest_rejects_one_value_per_line(tmp_path):
    """a file with one value per line cannot be framed (8 or 9 values per record?): an error, not a guess"""
    rows = _rows(9, n=6, seed=12)
    bad = tmp_path / "one_per_line.txt"
    with open(bad, "w") as f:
        f.write("header\n")
        for r in rows:
            for v in (r if r[6] != 0.0 else r[:8]):
                f.write(f" {v:.17e}\n")
    r = subprocess.run([_host("run_sph_hip"), str(bad), "-1", str(tmp_path / "o.txt")], capture_output=True, text=True, cwd=tmp_path, timeout=120)
    assert r.returncode != 0
    assert "one value per line" in r.stdout + r.stderr
