"""The reference's plain-text particle format, for the Python test/bench harness.

Ingest convention (/root/reference/SUMMER_SPH.f90:617-653): one header line that is skipped,
then one record per line; the first 8 whitespace/comma separated values are
x y z vx vy vz u m, any further columns are ignored.  Snapshot convention
(SUMMER_SPH.f90:719-738): header, 9 values per gas row (.., alpha), 8 per sink row with u
written as 0.  The Fortran host (summersph_amd/host) carries the product reader/writer; this
module only lets Python tests produce and read the same files.
"""
from __future__ import annotations

import numpy as np

HEADER = "x y z vx vy vz energy mass"


def write_ic(path: str, rows: np.ndarray, header: str = HEADER) -> None:
    rows = np.asarray(rows, dtype=np.float64)
    with open(path, "w") as f:
        f.write(header + "\n")
        for r in rows:
            # 17 significant digits: exact round trip through list-directed reads
            f.write(" ".join(f"{v:.17e}" for v in r) + "\n")


def read_ic(path: str) -> np.ndarray:
    """Returns the (n, 8) array of the first eight values of every data line."""
    out = []
    with open(path) as f:
        f.readline()
        for line in f:
            tok = line.replace(",", " ").split()
            if len(tok) < 8:
                if not tok:
                    continue
                raise ValueError(f"{path}: record with {len(tok)} < 8 values")
            out.append([float(t.replace("D", "E").replace("d", "e")) for t in tok[:8]])
    if not out:
        raise ValueError(f"{path}: no data records")
    return np.asarray(out, dtype=np.float64)


def read_snapshot(path: str):
    """Reads a save file written by the host: returns (gas rows (n,9), sink rows (ns,8)).
    Records may be wrapped over several lines (flang's list-directed output wraps at 80
    columns), so tokens are re-grouped: 9 per gas record, then sinks are the
    records whose 7th value is exactly 0."""
    with open(path) as f:
        f.readline()
        toks = f.read().split()
    vals = [float(t.replace("D", "E")) for t in toks]
    gas, sinks = [], []
    i = 0
    while i < len(vals):
        if i + 6 < len(vals) and vals[i + 6] == 0.0:
            sinks.append(vals[i:i + 8]); i += 8
        else:
            gas.append(vals[i:i + 9]); i += 9
    return np.asarray(gas).reshape(-1, 9), np.asarray(sinks).reshape(-1, 8)


def read_snapshot_v(path: str):
    """Save file of the variable-h host (one record per line): gas rows (n,10) x y z vx vy vz u m alpha h,
    sink rows (ns,8) with u written as 0 ("SUMMER_SPH - Variable.f90":921-940)."""
    gas, sinks = [], []
    with open(path) as f:
        f.readline()
        for line in f:
            v = [float(t.replace("D", "E")) for t in line.split()]
            if not v:
                continue
            (sinks if len(v) == 8 and v[6] == 0.0 else gas).append(v)
    return np.asarray(gas).reshape(-1, 10), np.asarray(sinks).reshape(-1, 8)
