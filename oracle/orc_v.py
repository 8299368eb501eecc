"""TEST INFRASTRUCTURE: ctypes view of the variable-h CPU restatement (oracle/sph_oracle_v.c).
Only tests/, smoke() and bench.py's cpu_baseline leg import this module."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import orc

_D = C.POINTER(C.c_double)


def _p(a):
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_D)


def lib():
    l = orc.lib()
    l.orcv_dt_candidate.restype = C.c_double
    l.orcv_dt_update.restype = C.c_double
    l.orcv_dt_update.argtypes = [C.c_double, C.c_double]
    l.orcv_G.restype = C.c_double
    l.orcv_pi.restype = C.c_double
    return l


def lookup_kernel(r, h, nq=2500):
    w, dw, _ = orc.tables(nq)
    r = np.ascontiguousarray(r, dtype=np.float64); h = np.ascontiguousarray(h, dtype=np.float64)
    W = np.zeros_like(r); dW = np.zeros_like(r)
    lib().orcv_lookup_kernel(_p(w), _p(dw), C.c_int(nq), C.c_int(r.size), _p(r), _p(h), _p(W), _p(dW))
    return W, dW


class OracleV:
    """gas + sink state as numpy arrays; the variable-h passes of 'SUMMER_SPH - Variable.f90'"""

    GAS = "x y z vx vy vz u m alpha h".split()
    DERIVED = "rho omega P c ax ay az du dalpha".split()

    def __init__(self, gas: dict, sinks: dict, gamma=1.4, eta=1.2, tol=1e-3, max_length=10.0, scale=0.25,
                 nq: int = 2500, max_depth: int = 1000, nthreads: int = 1):
        self.gamma, self.eta, self.tol, self.max_length, self.scale = gamma, eta, tol, max_length, scale
        self.nq, self.max_depth, self.nthreads = nq, max_depth, int(nthreads)
        self.n = int(gas["x"].size)
        self.ns = int(sinks["x"].size)
        for k in self.GAS:
            setattr(self, k, np.ascontiguousarray(gas[k], dtype=np.float64).copy())
        for k in self.DERIVED:
            setattr(self, k, np.zeros(self.n))
        for k in "x y z vx vy vz m".split():
            setattr(self, "s" + k, np.ascontiguousarray(sinks[k], dtype=np.float64).copy())
        for k in "sax say saz".split():
            setattr(self, k, np.zeros(self.ns))
        self.w, self.dw, _ = orc.tables(nq)
        self.lc = np.zeros(3 * self.n); self.ls = np.zeros(self.n); self.root = np.zeros(4)

    def leaves(self):
        lib().orcv_leaves(C.c_int(self.n), _p(self.x), _p(self.y), _p(self.z), C.c_int(self.max_depth), _p(self.lc), _p(self.ls),
                          _p(self.root))

    def density(self):
        self.leaves()
        lib().orcv_density(C.c_int(self.n), _p(self.x), _p(self.y), _p(self.z), _p(self.m), _p(self.h), _p(self.lc), _p(self.ls),
                           C.c_int(self.nq), _p(self.w), _p(self.dw), _p(self.rho), _p(self.omega), C.c_int(self.nthreads))
        lib().orcv_eos(C.c_int(self.n), _p(self.u), _p(self.rho), C.c_double(self.gamma), _p(self.P), _p(self.c))

    def forces(self):
        l = lib()
        l.orc_sink_gravity(C.c_int(self.n), _p(self.x), _p(self.y), _p(self.z), _p(self.m), C.c_int(self.ns), _p(self.sx),
                           _p(self.sy), _p(self.sz), _p(self.sm), _p(self.ax), _p(self.ay), _p(self.az), _p(self.sax),
                           _p(self.say), _p(self.saz))
        l.orcv_sph_forces(C.c_int(self.n), _p(self.x), _p(self.y), _p(self.z), _p(self.vx), _p(self.vy), _p(self.vz), _p(self.m),
                          _p(self.h), _p(self.rho), _p(self.omega), _p(self.P), _p(self.c), _p(self.alpha), _p(self.lc),
                          _p(self.ls), C.c_int(self.nq), _p(self.w), _p(self.dw), _p(self.ax), _p(self.ay), _p(self.az),
                          _p(self.du), _p(self.dalpha), C.c_int(self.nthreads))

    def evaluate(self):
        self.density()
        self.forces()

    def kick(self, dt):
        lib().orc_kick(C.c_int(self.n), _p(self.vx), _p(self.vy), _p(self.vz), _p(self.u), _p(self.alpha), _p(self.ax), _p(self.ay),
                       _p(self.az), _p(self.du), _p(self.dalpha), C.c_int(self.ns), _p(self.svx), _p(self.svy), _p(self.svz),
                       _p(self.sax), _p(self.say), _p(self.saz), C.c_double(dt))

    def drift(self, dt):
        lib().orc_drift(C.c_int(self.n), _p(self.x), _p(self.y), _p(self.z), _p(self.vx), _p(self.vy), _p(self.vz), C.c_int(self.ns),
                        _p(self.sx), _p(self.sy), _p(self.sz), _p(self.svx), _p(self.svy), _p(self.svz), C.c_double(dt))

    def dt_candidate(self):
        return lib().orcv_dt_candidate(C.c_int(self.n), _p(self.vx), _p(self.vy), _p(self.vz), _p(self.ax), _p(self.ay), _p(self.az),
                                       _p(self.u), _p(self.du), _p(self.c), _p(self.h), C.c_double(self.scale))

    def next_dt(self, dt):
        return lib().orcv_dt_update(C.c_double(self.dt_candidate()), C.c_double(dt))

    def update_h(self):
        """calc_smoothing on the tree of the last evaluation"""
        lib().orcv_update_h(C.c_int(self.n), _p(self.x), _p(self.y), _p(self.z), _p(self.m), _p(self.h), _p(self.rho),
                            _p(self.omega), _p(self.lc), _p(self.ls), C.c_int(self.nq), _p(self.w), _p(self.dw),
                            C.c_double(self.eta), C.c_double(self.tol), C.c_double(self.max_length), C.c_int(self.nthreads))

    def step(self, dt):
        """one iteration of the variable-h simulate loop body (Variable.f90:1120-1152, 'sph' variant)"""
        self.evaluate(); self.kick(dt); self.drift(dt)
        self.evaluate(); self.kick(dt)
        ndt = self.next_dt(dt)
        self.update_h()
        return ndt
