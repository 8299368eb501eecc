"""TEST INFRASTRUCTURE: an oracle-backed stand-in for summersph_amd.dist.HipBackend, so that the
multi-rank orchestration (migration, ghost exchange, reductions) can run on CPUs under gloo.
It implements the same interface on numpy arrays with oracle/sph_oracle.c doing the arithmetic."""
import ctypes as C
from types import SimpleNamespace

import numpy as np
import torch

from oracle import orc

_D = C.POINTER(C.c_double)


def _p(a):
    return a.ctypes.data_as(_D)


class OracleBackend:
    STATE = ["x", "y", "z", "vx", "vy", "vz", "u", "m", "alpha"]

    def __init__(self, h=2.5, nq=5000):
        self.device = torch.device("cpu")
        self.h, self.nq = h, nq
        self.w, self.dw, _ = orc.tables(nq)
        self.n = self.n_owned = 0
        self.rank, self.nranks = 0, 1
        self.f = {}
        self.params = SimpleNamespace(h=h, dt_max=float(np.float32(0.1)), dt_min=float(np.float32(0.0001)))

    def set_rank(self, rank, nranks):
        self.rank, self.nranks = rank, nranks

    def upload(self, state, n_owned):
        self.n, self.n_owned = int(state.shape[1]), int(n_owned)
        for i, k in enumerate(self.STATE):
            self.f[k] = np.ascontiguousarray(state[i].cpu().numpy(), dtype=np.float64).copy()
        for k in "rho P c ax ay az du dalpha".split():
            self.f[k] = np.zeros(self.n)
        self._ghost_rho = None

    def gather(self, names, ids=None, count=None):
        idx = np.arange(int(count)) if ids is None else ids.cpu().numpy()
        return torch.from_numpy(np.stack([self.f[k][idx] for k in names]))

    def scatter(self, names, first, vals):
        v = vals.cpu().numpy()
        for i, name in enumerate(names):
            self.f[name][first:first + v.shape[1]] = v[i]
            if name == "rho":  # ghosts keep the rho their owner sent (the HIP kernels never overwrite it)
                self._ghost_rho = self.f["rho"][self.n_owned:].copy()

    def set_sinks(self, s):
        self.s = {k: np.array(s[k], dtype=np.float64, copy=True) for k in "x y z vx vy vz m".split()}
        ns = self.s["x"].size
        for k in ("ax", "ay", "az"):
            self.s[k] = np.zeros(ns)

    def get_sinks(self):
        return {k: v.copy() for k, v in self.s.items()}

    def set_sink_accel(self, ax, ay, az):
        self.s["ax"][:], self.s["ay"][:], self.s["az"][:] = ax, ay, az

    def density(self):
        f = self.f
        orc.lib().orc_density(C.c_int(self.n), _p(f["x"]), _p(f["y"]), _p(f["z"]), _p(f["m"]), C.c_double(self.h),
                              C.c_int(self.nq), _p(self.w), _p(self.dw), _p(f["rho"]), C.c_int(2))
        if self._ghost_rho is not None:
            f["rho"][self.n_owned:] = self._ghost_rho
        self.refresh_eos()

    def refresh_eos(self):
        f = self.f
        orc.lib().orc_eos(C.c_int(self.n), _p(f["u"]), _p(f["rho"]), _p(f["P"]), _p(f["c"]))

    def forces(self):
        f, s = self.f, self.s
        ns = s["x"].size
        sax, say, saz = np.zeros(ns), np.zeros(ns), np.zeros(ns)
        # gas side for everything (ghost results are discarded by the orchestrator)
        orc.lib().orc_sink_gravity(C.c_int(self.n), _p(f["x"]), _p(f["y"]), _p(f["z"]), _p(f["m"]), C.c_int(ns),
                                   _p(s["x"]), _p(s["y"]), _p(s["z"]), _p(s["m"]), _p(f["ax"]), _p(f["ay"]), _p(f["az"]),
                                   _p(sax), _p(say), _p(saz))
        # sink side: OWNED particles only (+ sink-sink pairs on rank 0), as the HIP kernels do
        G = orc.lib().orc_G()
        o = slice(0, self.n_owned)
        for i in range(ns):
            v = np.stack([f["x"][o] - s["x"][i], f["y"][o] - s["y"][i], f["z"][o] - s["z"][i]])
            d3 = np.sqrt((v ** 2).sum(0)) ** 3
            acc = (f["m"][o] * (G * v / d3)).sum(1)
            s["ax"][i], s["ay"][i], s["az"][i] = acc
        if ns >= 2 and self.rank == 0:
            for i in range(ns):
                for j in range(i):
                    v = np.array([s["x"][j] - s["x"][i], s["y"][j] - s["y"][i], s["z"][j] - s["z"][i]])
                    w = G * v / np.sqrt((v ** 2).sum()) ** 3
                    for k, a in enumerate(("ax", "ay", "az")):
                        s[a][i] += s["m"][j] * w[k]
                        s[a][j] -= s["m"][i] * w[k]
        orc.lib().orc_sph_forces(C.c_int(self.n), _p(f["x"]), _p(f["y"]), _p(f["z"]), _p(f["vx"]), _p(f["vy"]), _p(f["vz"]),
                                 _p(f["m"]), _p(f["rho"]), _p(f["P"]), _p(f["c"]), _p(f["alpha"]), C.c_double(self.h),
                                 C.c_int(self.nq), _p(self.w), _p(self.dw), _p(f["ax"]), _p(f["ay"]), _p(f["az"]),
                                 _p(f["du"]), _p(f["dalpha"]), C.c_int(2))

    def kick(self, dt):
        f, s = self.f, self.s
        orc.lib().orc_kick(C.c_int(self.n), _p(f["vx"]), _p(f["vy"]), _p(f["vz"]), _p(f["u"]), _p(f["alpha"]),
                           _p(f["ax"]), _p(f["ay"]), _p(f["az"]), _p(f["du"]), _p(f["dalpha"]), C.c_int(s["x"].size),
                           _p(s["vx"]), _p(s["vy"]), _p(s["vz"]), _p(s["ax"]), _p(s["ay"]), _p(s["az"]), C.c_double(dt))

    def drift(self, dt):
        f, s = self.f, self.s
        orc.lib().orc_drift(C.c_int(self.n), _p(f["x"]), _p(f["y"]), _p(f["z"]), _p(f["vx"]), _p(f["vy"]), _p(f["vz"]),
                            C.c_int(s["x"].size), _p(s["x"]), _p(s["y"]), _p(s["z"]), _p(s["vx"]), _p(s["vy"]), _p(s["vz"]),
                            C.c_double(dt))

    def dt_candidate(self):
        f = self.f
        o = self.n_owned
        a = {k: np.ascontiguousarray(f[k][:o]) for k in "vx vy vz ax ay az u du c".split()}
        return orc.lib().orc_dt_candidate(C.c_int(o), _p(a["vx"]), _p(a["vy"]), _p(a["vz"]), _p(a["ax"]), _p(a["ay"]),
                                          _p(a["az"]), _p(a["u"]), _p(a["du"]), _p(a["c"]), C.c_double(self.h))

    def synchronize(self):
        pass
