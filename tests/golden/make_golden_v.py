#!/usr/bin/env python3
"""Golden fixtures for the VARIABLE-h path, from the REAL reference ("SUMMER_SPH - Variable.f90").

Build container only: needs oracle/_ref/ref_driver_v (oracle/build_ref.sh: unmodified module lines
1-1165, serial amdflang -O2, + our dump driver oracle/ref_driver_v.f90).  The reference ships no
parameters.txt; the values used here are gamma 1.4, eta 1.2, convergence 1e-3, max_length 10,
timestep_scale 0.25.

    python tests/golden/make_golden_v.py

  kernel_v.npz         lookup_kernel(r, h) probes for several h + tables (nq = 2500)
  discv3000_eval.npz   one evaluation (rho, Omega, P, c, rates, dt, h after calc_smoothing), smooth h field
  discv2000r_eval.npz  the same with a ROUGH h field (uniform random 1.5..3.5): stresses the leaf-AABB rule
  discv3000_traj.npz   simulate-loop steps 1 and 5 ("sph" variant), step 5 ("full"), dt sequences
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from make_golden import keep_steps, parse_records  # noqa: E402
from summersph_amd import ic, txtio  # noqa: E402

DRIVER = os.path.join(ROOT, "oracle", "_ref", "ref_driver_v")
PARAMS = dict(gamma=1.4, eta=1.2, tol=1e-3, maxlen=10.0, scale=0.25)
HDR = "x y z vx vy vz energy mass alpha smoothing"


def run(mode, infile, *extra):
    with tempfile.TemporaryDirectory() as td:
        outp = os.path.join(td, "out.bin")
        subprocess.run([DRIVER, mode, infile, outp, *[repr(float(e)) if isinstance(e, float) else str(e) for e in extra]],
                       check=True, stdout=subprocess.DEVNULL, cwd=td)
        return parse_records(outp)


def sink_creation_ic():
    """variable-h disc plus ONE very massive particle well outside it: calc_smoothing drives its h to ~13 (self-dominated),
    it still passes check_sink_creation's test m (eta/h)^3 > 0.5 ([V]:560) and lies farther than radius + 2h from the
    file's sink, so a second sink appears at its place in step 1 -- and accretes its own seed in the same step"""
    rows = ic.keplerian_disc_var(1500, seed=808)
    k = 24
    rows[k, 0:3] = [60.0, 0.0, 0.1]
    rows[k, 3:6] = [0.0, np.sqrt(39.478416442871094 * 1.0 / 60.0), 0.0]
    rows[k, 7] = 2000.0
    return rows, k


def sink_creation_fixture(td, p):
    rows, k = sink_creation_ic()
    f3 = os.path.join(td, "sc.txt"); txtio.write_ic(f3, rows, header=HDR)
    t = keep_steps(run("traj", f3, *p, 3, "full"), {1, 2, 3})
    np.savez(os.path.join(HERE, "sinkcv1500_traj.npz"), ic=rows, params=np.array(p), heavy=np.array([k]),
             **{"full_" + kk: v for kk, v in t.items()})


def sink_cull_fixture(td, p):
    """[V]'s check_bounds also packs the SINKS: a second sink starts outside the box and is gone after step 1"""
    rows = ic.keplerian_disc_var(1000, seed=909)
    far = np.zeros((1, 10)); far[0, 0:3] = [1600.0, 2.0, 0.0]; far[0, 7] = 0.5
    rows = np.vstack([rows[:300], far, rows[300:]])
    f4 = os.path.join(td, "cull.txt"); txtio.write_ic(f4, rows, header=HDR)
    t = keep_steps(run("traj", f4, *p, 3, "full"), {1, 3})
    np.savez(os.path.join(HERE, "sinkcullv1000_traj.npz"), ic=rows, params=np.array(p), **{"full_" + kk: v for kk, v in t.items()})


def main():
    if not os.path.exists(DRIVER):
        sys.exit("oracle/_ref/ref_driver_v missing: run oracle/build_ref.sh first (build container only)")
    p = [PARAMS[k] for k in ("gamma", "eta", "tol", "maxlen", "scale")]
    with tempfile.TemporaryDirectory() as td:
        nq = 2500
        dq = 2.0 / nq
        qs = [0.0, dq, 0.5 * dq, 0.25, 0.5, 1.0 - 0.5 * dq, 1.0, 1.0 + dq, 1.5, 2.0 - dq, 2.0 - 1e-12, 2.0, 2.0 + 1e-12, 3.0,
              0.123456789, 1.987654321]
        rf = os.path.join(td, "rh.txt")
        with open(rf, "w") as f:
            hs = [2.5, 1.7, 3.3333333333333335, 0.9]
            f.write(f"{len(qs) * len(hs)}\n")
            for h in hs:
                for q in qs:
                    f.write(f"{q * h:.17e} {h:.17e}\n")
        np.savez(os.path.join(HERE, "kernel_v.npz"), **run("kernel", rf))

        disc = ic.keplerian_disc_var(3000, seed=303)
        f1 = os.path.join(td, "d.txt"); txtio.write_ic(f1, disc, header=HDR)
        np.savez(os.path.join(HERE, "discv3000_eval.npz"), ic=disc, params=np.array(p), **run("eval", f1, *p))
        t_s = keep_steps(run("traj", f1, *p, 5, "sph"), {1, 5})
        t_f = keep_steps(run("traj", f1, *p, 5, "full"), {5})
        np.savez(os.path.join(HERE, "discv3000_traj.npz"), ic=disc, params=np.array(p),
                 **{"sph_" + k: v for k, v in t_s.items()}, **{"full_" + k: v for k, v in t_f.items()})

        rough = ic.keplerian_disc_var(2000, seed=304)
        rough[:-1, 9] = np.random.default_rng(9).uniform(1.5, 3.5, 2000)
        f2 = os.path.join(td, "r.txt"); txtio.write_ic(f2, rough, header=HDR)
        np.savez(os.path.join(HERE, "discv2000r_eval.npz"), ic=rough, params=np.array(p), **run("eval", f2, *p))
        sink_creation_fixture(td, p)
        sink_cull_fixture(td, p)
    for fn in sorted(os.listdir(HERE)):
        if fn.endswith(".npz") and "v" in fn.split("_")[0]:
            print(f"{fn:28s} {os.path.getsize(os.path.join(HERE, fn)) / 1024:8.1f} KiB")


if __name__ == "__main__":
    main()
