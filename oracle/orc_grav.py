"""TEST INFRASTRUCTURE: ctypes view of oracle/sph_oracle_grav.c (Barnes-Hut gas self-gravity, sink
accretion, boundary cull) and a 'full' step = the reference's simulate() loop body as is."""
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
    l.orcg_tree_build.restype = C.c_void_p
    l.orcg_tree_free.argtypes = [C.c_void_p]
    l.orcg_tree_nodes.argtypes = [C.c_void_p]
    return l


class Tree:
    def __init__(self, x, y, z, m, max_depth=1000):
        self._keep = (x, y, z, m)
        self.h = C.c_void_p(lib().orcg_tree_build(C.c_int(x.size), _p(x), _p(y), _p(z), _p(m), C.c_int(max_depth)))

    def free(self):
        if self.h:
            lib().orcg_tree_free(self.h); self.h = None

    def __del__(self):
        self.free()


def gravity(tree: Tree, x, y, z, ax, ay, az, h=2.5, h_var=None, theta=0.5, nq=5000, nthreads=1):
    """adds the Barnes-Hut gas self-gravity to ax, ay, az (in place)"""
    _, _, g = orc.tables(nq)
    lib().orcg_gravity(tree.h, C.c_int(x.size), _p(x), _p(y), _p(z), C.c_double(h), _p(h_var) if h_var is not None else None,
                       C.c_double(theta), C.c_int(nq), _p(g), _p(ax), _p(ay), _p(az), C.c_int(nthreads))


class OracleFull(orc.Oracle):
    """orc.Oracle + gas self-gravity + accretion + cull: the reference's simulate() loop body as is ([F]:886-928)"""

    def __init__(self, gas, sinks, bounding_size=1500.0, **kw):
        super().__init__(gas, sinks, **kw)
        self.srad = np.ascontiguousarray(sinks["radius"], dtype=np.float64).copy()
        self.bound = bounding_size
        self.tree = None

    def _rebind(self):
        self._st.n = self.n
        for name, _ in orc._State._fields_[4:]:
            setattr(self._st, name, _p(getattr(self, name)))

    def evaluate(self):
        l = orc.lib()
        self.tree = Tree(self.x, self.y, self.z, self.m)
        l.orc_density(C.c_int(self.n), _p(self.x), _p(self.y), _p(self.z), _p(self.m), C.c_double(self.h), C.c_int(self.nq),
                      _p(self.w), _p(self.dw), _p(self.rho), C.c_int(self.nthreads))
        l.orc_eos(C.c_int(self.n), _p(self.u), _p(self.rho), _p(self.P), _p(self.c))
        # find_forces order ([F]:824-827): zero, BH gravity, sink gravity, SPH.  orc_sink_gravity zeroes first, so the
        # gravity term is added right after it (a = (0 - sink) - bh instead of (0 - bh) - sink: rounding only)
        l.orc_sink_gravity(C.c_int(self.n), _p(self.x), _p(self.y), _p(self.z), _p(self.m), C.c_int(self.ns), _p(self.sx), _p(self.sy),
                           _p(self.sz), _p(self.sm), _p(self.ax), _p(self.ay), _p(self.az), _p(self.sax), _p(self.say), _p(self.saz))
        gravity(self.tree, self.x, self.y, self.z, self.ax, self.ay, self.az, h=self.h, nq=self.nq, nthreads=self.nthreads)
        l.orc_sph_forces(C.c_int(self.n), _p(self.x), _p(self.y), _p(self.z), _p(self.vx), _p(self.vy), _p(self.vz), _p(self.m),
                         _p(self.rho), _p(self.P), _p(self.c), _p(self.alpha), C.c_double(self.h), C.c_int(self.nq), _p(self.w),
                         _p(self.dw), _p(self.ax), _p(self.ay), _p(self.az), _p(self.du), _p(self.dalpha), C.c_int(self.nthreads))

    def accrete_and_cull(self, variant=0):
        """[F]:919-920 on the tree of the last evaluation; returns the number of removed particles"""
        keep = np.ones(self.n, dtype=np.uint8)
        if np.any(self.sm > 0.0):
            lib().orcg_accrete(self.tree.h, C.c_int(self.n), _p(self.vx), _p(self.vy), _p(self.vz), C.c_int(self.ns), _p(self.sx),
                               _p(self.sy), _p(self.sz), _p(self.svx), _p(self.svy), _p(self.svz), _p(self.sm), _p(self.srad),
                               C.c_int(variant), keep.ctypes.data_as(C.POINTER(C.c_ubyte)))
        # the reference packs after accretion and then culls the packed array: same set as one combined mask
        lib().orcg_cull(C.c_int(self.n), _p(self.x), _p(self.y), _p(self.z), C.c_double(self.bound),
                        keep.ctypes.data_as(C.POINTER(C.c_ubyte)))
        removed = int(self.n - keep.sum())
        if removed:
            k = keep.astype(bool)
            for f in self.GAS + self.DERIVED:
                setattr(self, f, np.ascontiguousarray(getattr(self, f)[k]))
            self.n = int(k.sum())
            self._rebind()
        return removed

    def step(self, dt):
        l = orc.lib()
        self.evaluate()
        l.orc_kick(C.c_int(self.n), _p(self.vx), _p(self.vy), _p(self.vz), _p(self.u), _p(self.alpha), _p(self.ax), _p(self.ay), _p(self.az),
                   _p(self.du), _p(self.dalpha), C.c_int(self.ns), _p(self.svx), _p(self.svy), _p(self.svz), _p(self.sax), _p(self.say),
                   _p(self.saz), C.c_double(dt))
        l.orc_drift(C.c_int(self.n), _p(self.x), _p(self.y), _p(self.z), _p(self.vx), _p(self.vy), _p(self.vz), C.c_int(self.ns),
                    _p(self.sx), _p(self.sy), _p(self.sz), _p(self.svx), _p(self.svy), _p(self.svz), C.c_double(dt))
        self.evaluate()
        l.orc_kick(C.c_int(self.n), _p(self.vx), _p(self.vy), _p(self.vz), _p(self.u), _p(self.alpha), _p(self.ax), _p(self.ay), _p(self.az),
                   _p(self.du), _p(self.dalpha), C.c_int(self.ns), _p(self.svx), _p(self.svy), _p(self.svz), _p(self.sax), _p(self.say),
                   _p(self.saz), C.c_double(dt))
        ndt = self.next_dt(dt)
        self.accrete_and_cull()
        return ndt
