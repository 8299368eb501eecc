#!/usr/bin/env python3
"""Generates the golden fixtures in this directory from the REAL reference.

Runs only in the build container: it needs oracle/_ref/ref_driver, which
oracle/build_ref.sh compiles from /root/reference/SUMMER_SPH.f90 (unmodified module, serial
amdflang -O2 build) plus our dump driver oracle/ref_driver.f90.  The fixtures are data only
(seeded inputs + the reference's outputs); the IC generators are summersph_amd/ic.py.

    python tests/golden/make_golden.py

Fixtures written (numpy .npz, float64, no pickles):
  kernel.npz            lookup_kernel / lookup_grav_kernel at probe radii + the three tables
  sod1000_eval.npz      one force evaluation, 3-D Sod column (no sink row -> dummy-sink path)
  sod1000_traj.npz      simulate-loop steps 1 and 5 (sph variant) and 5 (full variant)
  disc3000_eval.npz     one force evaluation, Keplerian disc 3000 gas + 1 sink
  disc3000_traj.npz     steps 1, 5 (sph), step 5 (full), dt sequences
  disc3000_long.npz     step 40 (sph), dt sequence of 40 steps
  disc3000ns_eval.npz   the same disc without its sink row (dummy-sink path)
  acc2000_traj.npz      simulate() as it is, 3 steps: 3 particles accreted, 3 culled... (full variant)
  bin2000_eval/_traj    circumbinary disc, TWO sinks (sink-sink forces, two accretors): one evaluation, and the
                        full loop for 3 steps
  ring3000_traj.npz     thin ring r ~ N(r0, 0.05 r0) (BASELINE configs[3] shape) with a velocity dispersion: step 8
                        (sph variant; state, rho, du); the viscosity switch alpha grows from 0 and the viscous terms act from step 2 on
                        (`python tests/golden/make_golden.py ring` writes this one only)
"eval" files hold both the `sph_*` rates (zero_rates + sink_gravforces + get_SPH, i.e.
find_forces without Barnes-Hut gas self-gravity) and the `full_*` rates (find_forces as is).
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from summersph_amd import ic, txtio  # noqa: E402

DRIVER = os.path.join(ROOT, "oracle", "_ref", "ref_driver")


def parse_records(path):
    out = {}
    with open(path, "rb") as f:
        buf = f.read()
    off = 0
    while off < len(buf):
        name = buf[off:off + 16].decode().strip(); off += 16
        n = int(np.frombuffer(buf, dtype="<i8", count=1, offset=off)[0]); off += 8
        out[name] = np.frombuffer(buf, dtype="<f8", count=n, offset=off).copy(); off += 8 * n
    return out


def run(mode, infile, *extra):
    with tempfile.TemporaryDirectory() as td:
        outp = os.path.join(td, "out.bin")
        subprocess.run([DRIVER, mode, infile, outp, *[str(e) for e in extra]], check=True,
                       stdout=subprocess.DEVNULL, cwd=td)
        return parse_records(outp)


def keep_steps(rec, steps, extra=("dt_seq", "n_seq", "t_end")):
    keep = {}
    for k, v in rec.items():
        if k in extra:
            keep[k] = v
        elif k[0] == "s" and "_" in k and k[1:k.index("_")].isdigit():
            if int(k[1:k.index("_")]) in steps:
                keep[k] = v
    return keep


def binary_ic():
    """circumbinary disc: 2000 gas particles (inner edge inside the sinks' reach) + two sinks of 0.6 and 0.4 on a circular
    orbit of separation 6 around the origin"""
    rows = ic.keplerian_disc(2000, seed=707, r_in=2.0, nngb=30.0)[:-1]
    G = 39.478416442871094
    m1, m2, a = 0.6, 0.4, 6.0
    w = np.sqrt(G * (m1 + m2) / a ** 3)
    x1, x2 = a * m2 / (m1 + m2), -a * m1 / (m1 + m2)
    s = np.zeros((2, 8))
    s[0, :8] = [x1, 0.0, 0.0, 0.0, w * x1, 0.0, 0.0, m1]
    s[1, :8] = [x2, 0.0, 0.0, 0.0, w * x2, 0.0, 0.0, m2]
    return np.vstack([rows[:700], s[:1], rows[700:], s[1:]])       # sink rows anywhere in the file


def binary_fixture(td):
    b = binary_ic()
    p = os.path.join(td, "bin.txt"); txtio.write_ic(p, b)
    np.savez(os.path.join(HERE, "bin2000_eval.npz"), ic=b, **run("eval", p))
    t_b = keep_steps(run("traj", p, 3, "full"), {1, 3})
    t_s = keep_steps(run("traj", p, 3, "sph"), {3})
    np.savez(os.path.join(HERE, "bin2000_traj.npz"), ic=b, **{"full_" + k: v for k, v in t_b.items()},
             **{"sph_" + k: v for k, v in t_s.items()})


def ring_fixture(td):
    ring = ic.thin_ring(3000, seed=404)
    rng = np.random.default_rng(405)
    ring[:-1, 3:6] += rng.normal(0.0, 0.3, (3000, 3))           # converging and diverging pairs: alpha sources, viscous heating
    p = os.path.join(td, "ring.txt"); txtio.write_ic(p, ring)
    t_r = keep_steps(run("traj", p, 8, "sph"), {8})
    keep = {k: v for k, v in t_r.items() if k in ("dt_seq", "n_seq") or k.split("_", 1)[1] in
            ("x", "y", "z", "vx", "vy", "vz", "u", "alpha", "rho", "du", "sx", "svx", "sm")}
    np.savez_compressed(os.path.join(HERE, "ring3000_traj.npz"), ic=ring, **{"sph_" + k: v for k, v in keep.items()})


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "ring":
        with tempfile.TemporaryDirectory() as td:
            ring_fixture(td)
        return
    if not os.path.exists(DRIVER):
        sys.exit("oracle/_ref/ref_driver missing: run oracle/build_ref.sh first (build container only)")
    with tempfile.TemporaryDirectory() as td:
        # ---- kernel probes: q = 0, knots, mid-knots, 1, 2, 2-eps, beyond support --------
        h, nq = 2.5, 5000
        dq = 2.0 / nq
        qs = [0.0, dq, 0.5 * dq, 1.5 * dq, 0.25, 0.5, 0.75, 1.0 - dq, 1.0 - 0.5 * dq, 1.0, 1.0 + 0.5 * dq,
              1.0 + dq, 1.25, 1.5, 1.75, 2.0 - dq, 2.0 - 0.5 * dq, 2.0 - 1e-12, 2.0, 2.0 + 1e-12, 2.5, 7.0,
              0.123456789, 1.987654321, np.nextafter(2.0, 0.0)]
        rfile = os.path.join(td, "r.txt")
        with open(rfile, "w") as f:
            f.write(f"{len(qs)}\n")
            for q in qs:
                f.write(f"{q * h:.17e}\n")
        np.savez(os.path.join(HERE, "kernel.npz"), **run("kernel", rfile))

        # ---- Sod column ------------------------------------------------------------------
        sod = ic.sod_column(seed=101)
        p = os.path.join(td, "sod.txt"); txtio.write_ic(p, sod)
        np.savez(os.path.join(HERE, "sod1000_eval.npz"), ic=sod, **run("eval", p))
        t_s = keep_steps(run("traj", p, 5, "sph"), {1, 5})
        t_f = keep_steps(run("traj", p, 5, "full"), {5})
        np.savez(os.path.join(HERE, "sod1000_traj.npz"), ic=sod,
                 **{"sph_" + k: v for k, v in t_s.items()}, **{"full_" + k: v for k, v in t_f.items()})

        # ---- Keplerian disc 3000 + sink ----------------------------------------------------
        disc = ic.keplerian_disc(3000, seed=202)
        p = os.path.join(td, "disc.txt"); txtio.write_ic(p, disc)
        np.savez(os.path.join(HERE, "disc3000_eval.npz"), ic=disc, **run("eval", p))
        t_s = keep_steps(run("traj", p, 5, "sph"), {1, 5})
        t_f = keep_steps(run("traj", p, 5, "full"), {5})
        np.savez(os.path.join(HERE, "disc3000_traj.npz"), ic=disc,
                 **{"sph_" + k: v for k, v in t_s.items()}, **{"full_" + k: v for k, v in t_f.items()})
        t_l = keep_steps(run("traj", p, 40, "sph"), {40})
        np.savez(os.path.join(HERE, "disc3000_long.npz"), ic=disc, **{"sph_" + k: v for k, v in t_l.items()})

        # ---- the same disc without a sink row (dummy massless sink) ------------------------
        discns = disc[:-1]
        p = os.path.join(td, "discns.txt"); txtio.write_ic(p, discns)
        np.savez(os.path.join(HERE, "disc3000ns_eval.npz"), ic=discns, **run("eval", p))
        # ---- accretion + boundary cull: inner edge inside the sink's reach, three particles outside the box ----
        acc = ic.keplerian_disc(2000, seed=606, r_in=0.6, nngb=30.0)
        acc[5, 0:3] = [1600.0, 3.0, 0.5]; acc[77, 0:3] = [-20.0, -1700.0, 1.0]; acc[1500, 0:3] = [10.0, 5.0, 1501.0]
        p = os.path.join(td, "acc.txt"); txtio.write_ic(p, acc)
        t_a = keep_steps(run("traj", p, 3, "full"), {1, 2, 3})
        np.savez(os.path.join(HERE, "acc2000_traj.npz"), ic=acc, **{"full_" + k: v for k, v in t_a.items()})
        binary_fixture(td)
        ring_fixture(td)
    for fn in sorted(os.listdir(HERE)):
        if fn.endswith(".npz"):
            print(f"{fn:28s} {os.path.getsize(os.path.join(HERE, fn)) / 1024:8.1f} KiB")


if __name__ == "__main__":
    main()
