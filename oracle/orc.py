"""TEST INFRASTRUCTURE: ctypes view of oracle/liborc.so (the CPU restatement, sph_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
The product path (summersph_amd.capi -> libsummersph_hip.so) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_D = C.POINTER(C.c_double)


def build() -> str:
    path = os.path.join(_HERE, "liborc.so")
    srcs = [os.path.join(_HERE, f) for f in ("sph_oracle.c", "sph_oracle_v.c", "sph_oracle_grav.c")]
    if not os.path.exists(path) or os.path.getmtime(path) < max(os.path.getmtime(f) for f in srcs):
        subprocess.run(["make", "-C", _HERE, "liborc.so"], check=True, stdout=subprocess.DEVNULL)
    return path


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.orc_G.restype = C.c_double
        _LIB.orc_dt_candidate.restype = C.c_double
        _LIB.orc_dt_update.restype = C.c_double
        _LIB.orc_dt_update.argtypes = [C.c_double, C.c_double]
        _LIB.orc_step.restype = C.c_double
        _LIB.orc_max_threads.restype = C.c_int
    return _LIB


def _p(a):
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_D)


def tables(nq: int = 5000):
    w = np.zeros(nq + 1); dw = np.zeros(nq + 1); g = np.zeros(nq + 1)
    lib().orc_init_tables(C.c_int(nq), _p(w), _p(dw), _p(g))
    return w, dw, g


def lookup_kernel(r, h=2.5, nq=5000):
    w, dw, _ = tables(nq)
    r = np.ascontiguousarray(r, dtype=np.float64)
    W = np.zeros_like(r); dW = np.zeros_like(r)
    lib().orc_lookup_kernel(_p(w), _p(dw), C.c_int(nq), C.c_int(r.size), _p(r), C.c_double(h), _p(W), _p(dW))
    return W, dW


def lookup_grav(r, h=2.5, nq=5000):
    _, _, g = tables(nq)
    r = np.ascontiguousarray(r, dtype=np.float64)
    out = np.zeros_like(r)
    lib().orc_lookup_grav(_p(g), C.c_int(nq), C.c_int(r.size), _p(r), C.c_double(h), _p(out))
    return out


class _State(C.Structure):
    _fields_ = ([("n", C.c_int), ("ns", C.c_int), ("nq", C.c_int), ("h", C.c_double)]
                + [(k, _D) for k in ("x y z vx vy vz u m alpha rho P c ax ay az du dalpha "
                                     "sx sy sz svx svy svz sm sax say saz w dw").split()])


class Oracle:
    """Holds gas + sink state as numpy arrays and runs the restated passes on them."""

    GAS = "x y z vx vy vz u m alpha".split()
    DERIVED = "rho P c ax ay az du dalpha".split()

    def __init__(self, gas: dict, sinks: dict, h: float = 2.5, nq: int = 5000, nthreads: int = 1):
        self.h, self.nq, self.nthreads = float(h), int(nq), int(nthreads)
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
        self.w, self.dw, self.grav = tables(nq)
        self._st = _State()
        self._st.n, self._st.ns, self._st.nq, self._st.h = self.n, self.ns, self.nq, self.h
        for name, _ in _State._fields_[4:]:
            setattr(self._st, name, _p(getattr(self, name)))

    def density(self):
        lib().orc_density(C.c_int(self.n), _p(self.x), _p(self.y), _p(self.z), _p(self.m), C.c_double(self.h),
                          C.c_int(self.nq), _p(self.w), _p(self.dw), _p(self.rho), C.c_int(self.nthreads))
        lib().orc_eos(C.c_int(self.n), _p(self.u), _p(self.rho), _p(self.P), _p(self.c))

    def evaluate(self):
        lib().orc_evaluate(C.byref(self._st), C.c_int(self.nthreads))

    def dt_candidate(self) -> float:
        return lib().orc_dt_candidate(C.c_int(self.n), _p(self.vx), _p(self.vy), _p(self.vz), _p(self.ax), _p(self.ay),
                                      _p(self.az), _p(self.u), _p(self.du), _p(self.c), C.c_double(self.h))

    def next_dt(self, dt: float) -> float:
        return lib().orc_dt_update(C.c_double(self.dt_candidate()), C.c_double(dt))

    def step(self, dt: float) -> float:
        """one iteration of simulate's loop body; returns the next dt"""
        return lib().orc_step(C.byref(self._st), C.c_double(dt), C.c_int(self.nthreads))


def max_threads() -> int:
    return int(lib().orc_max_threads())
