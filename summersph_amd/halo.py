"""ctypes face of libsummersph_halo.so (include/summersph_halo.h): the native multi-GPU step loop.

`Halo` wraps one sph_halo* on top of a `capi.Context`.  Three ways to get one, as the header says:
`Halo.rccl(ctx, id, rank, nranks)` (one process per GPU; the 128-byte id comes from `unique_id()` on
rank 0 and reaches the others by any means), `Halo.inproc(ctx, hub, rank, nranks)` (ranks = threads
of this process sharing one GPU: tests) -- and `sph_halo_attach` for hosts that already own an
ncclComm_t (no Python use).  Nothing here computes anything; the orchestration is C++ (csrc/halo.hip).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import capi

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libsummersph_halo.so")
ID_BYTES = 128
SYMBOLS = [
    "sph_halo_unique_id", "sph_halo_create", "sph_halo_attach", "sph_halo_hub_create", "sph_halo_hub_destroy",
    "sph_halo_create_inproc", "sph_halo_destroy", "sph_halo_last_error", "sph_halo_set_slabs", "sph_halo_upload",
    "sph_halo_run", "sph_halo_count", "sph_halo_download", "sph_halo_gather_root", "sph_halo_get_stats", "sph_halo_selftest",
    "sph_halo_upload_v", "sph_halo_download_v", "sph_halo_gather_root_v",
]
STATE = "x y z vx vy vz u m alpha".split()
_D = C.POINTER(C.c_double)
_lib = None


class HaloStats(C.Structure):
    _fields_ = [(k, C.c_int64) for k in "ghosts migrated exchanges collectives migrations host_waits removed sinks_created let_sent let_received let_updates".split()]


def load():
    """loads libsummersph_halo.so after the core library (and after torch's RCCL, when torch is importable: one RCCL and
    one HIP runtime per process, see capi.load)"""
    global _lib
    if _lib is not None:
        return _lib
    capi.load()
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
    try:
        import torch
        cand = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    lib.sph_halo_last_error.restype = C.c_char_p
    lib.sph_halo_last_error.argtypes = [C.c_void_p]
    lib.sph_halo_hub_create.restype = C.c_void_p
    lib.sph_halo_hub_create.argtypes = [C.c_int32]
    lib.sph_halo_hub_destroy.restype = None
    lib.sph_halo_hub_destroy.argtypes = [C.c_void_p]
    lib.sph_halo_count.restype = C.c_int64
    lib.sph_halo_count.argtypes = [C.c_void_p]
    lib.sph_halo_unique_id.argtypes = [C.c_void_p]
    lib.sph_halo_create.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
    lib.sph_halo_attach.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
    lib.sph_halo_create_inproc.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]
    lib.sph_halo_destroy.argtypes = [C.c_void_p]
    lib.sph_halo_set_slabs.argtypes = [C.c_void_p, _D, C.c_int32]
    lib.sph_halo_upload.argtypes = [C.c_void_p, C.c_int64] + [_D] * 9 + [C.POINTER(C.c_int64)]
    lib.sph_halo_run.argtypes = [C.c_void_p, C.c_int32, _D, _D]
    lib.sph_halo_download.argtypes = [C.c_void_p, C.c_int64] + [_D] * 9 + [C.POINTER(C.c_int64)]
    lib.sph_halo_gather_root.argtypes = [C.c_void_p, C.c_int32, C.c_int64, C.POINTER(C.c_int64)] + [_D] * 9 + [C.POINTER(C.c_int64)]
    lib.sph_halo_upload_v.argtypes = [C.c_void_p, C.c_int64] + [_D] * 10 + [C.POINTER(C.c_int64)]
    lib.sph_halo_download_v.argtypes = [C.c_void_p, C.c_int64] + [_D] * 10 + [C.POINTER(C.c_int64)]
    lib.sph_halo_gather_root_v.argtypes = [C.c_void_p, C.c_int32, C.c_int64, C.POINTER(C.c_int64)] + [_D] * 10 + [C.POINTER(C.c_int64)]
    lib.sph_halo_get_stats.argtypes = [C.c_void_p, C.POINTER(HaloStats)]
    lib.sph_halo_selftest.argtypes = [C.c_void_p, C.c_int64]
    _lib = lib
    return lib


def unique_id() -> bytes:
    buf = C.create_string_buffer(ID_BYTES)
    st = load().sph_halo_unique_id(buf)
    if st != 0:
        raise capi.SphError(st, "sph_halo_unique_id failed")
    return buf.raw


class Hub:
    """meeting point of the in-process transport: one per group of `nranks` threads"""

    def __init__(self, nranks: int):
        self.lib = load()
        self.ptr = self.lib.sph_halo_hub_create(int(nranks))
        if not self.ptr:
            raise ValueError("1 <= nranks <= 64")

    def close(self):
        if self.ptr:
            self.lib.sph_halo_hub_destroy(self.ptr)
            self.ptr = None


def _dp(a):
    return a.ctypes.data_as(_D) if a is not None else None


class Halo:
    def __init__(self, ctx: capi.Context, handle, rank: int, nranks: int):
        self.lib = load()
        self.ctx, self._h, self.rank, self.nranks = ctx, handle, rank, nranks

    @staticmethod
    def _made(ctx, st, handle, rank, nranks):
        if st != 0:
            raise capi.SphError(st, load().sph_halo_last_error(None).decode() or "sph_halo_create failed")
        return Halo(ctx, handle, rank, nranks)

    @classmethod
    def rccl(cls, ctx, uid: bytes, rank: int, nranks: int):
        h = C.c_void_p()
        st = load().sph_halo_create(ctx._h, C.c_char_p(uid), rank, nranks, C.byref(h))
        return cls._made(ctx, st, h, rank, nranks)

    @classmethod
    def attach(cls, ctx, nccl_comm: int, comm_stream: int | None, rank: int, nranks: int):
        """on a communicator (ncclComm_t as an integer) and, optionally, a HIP stream the caller owns"""
        h = C.c_void_p()
        st = load().sph_halo_attach(ctx._h, C.c_void_p(nccl_comm), C.c_void_p(comm_stream) if comm_stream else None, rank, nranks, C.byref(h))
        return cls._made(ctx, st, h, rank, nranks)

    @classmethod
    def inproc(cls, ctx, hub: Hub, rank: int, nranks: int):
        h = C.c_void_p()
        st = load().sph_halo_create_inproc(ctx._h, hub.ptr, rank, nranks, C.byref(h))
        return cls._made(ctx, st, h, rank, nranks)

    def _ck(self, st):
        if st != 0:
            raise capi.SphError(st, self.lib.sph_halo_last_error(self._h).decode())

    def close(self):
        if getattr(self, "_h", None):
            self.lib.sph_halo_destroy(self._h)
            self._h = None

    def set_slabs(self, edges, migrate_every: int = 32):
        e = np.ascontiguousarray(edges, dtype=np.float64)
        assert e.size == self.nranks - 1
        self._ck(self.lib.sph_halo_set_slabs(self._h, _dp(e) if e.size else None, int(migrate_every)))

    @property
    def variable(self) -> bool:
        return bool(self.ctx.params.flags & capi.FLAG_VARIABLE_H)

    def upload(self, gas: dict):
        """this rank's particles; a variable-h context also takes gas["h"] (the 10th column of [V]'s reader)"""
        arrs = [np.ascontiguousarray(gas[k], dtype=np.float64) for k in STATE[:8]]
        al = gas.get("alpha")
        al = None if al is None else np.ascontiguousarray(al, dtype=np.float64)
        gid = gas.get("gid")
        gid = None if gid is None else np.ascontiguousarray(gid, dtype=np.int64)
        gp = gid.ctypes.data_as(C.POINTER(C.c_int64)) if gid is not None else None
        if self.variable:
            hs = np.ascontiguousarray(gas["h"], dtype=np.float64)
            self._ck(self.lib.sph_halo_upload_v(self._h, arrs[0].size, *[_dp(a) for a in arrs], _dp(al), _dp(hs), gp))
        else:
            self._ck(self.lib.sph_halo_upload(self._h, arrs[0].size, *[_dp(a) for a in arrs], _dp(al), gp))

    def run(self, nsteps: int, dt: float, t: float = 0.0):
        d, tt = C.c_double(dt), C.c_double(t)
        self._ck(self.lib.sph_halo_run(self._h, int(nsteps), C.byref(d), C.byref(tt)))
        return d.value, tt.value

    @property
    def n_owned(self) -> int:
        return int(self.lib.sph_halo_count(self._h))

    def download(self) -> dict:
        n = self.n_owned
        names = STATE + (["h"] if self.variable else [])
        out = {k: np.empty(n) for k in names}
        gid = np.empty(n, dtype=np.int64)
        fn = self.lib.sph_halo_download_v if self.variable else self.lib.sph_halo_download
        self._ck(fn(self._h, n, *[_dp(out[k]) for k in names], gid.ctypes.data_as(C.POINTER(C.c_int64))))
        out["gid"] = gid
        return out

    def gather_root(self, root: int, capacity: int):
        """collective; on `root` a dict of the whole particle set in global-number order, elsewhere None"""
        cap = int(capacity) if self.rank == root else 0
        names = STATE + (["h"] if self.variable else [])
        out = {k: np.empty(cap) for k in names}
        gid = np.empty(cap, dtype=np.int64)
        nt = C.c_int64()
        fn = self.lib.sph_halo_gather_root_v if self.variable else self.lib.sph_halo_gather_root
        self._ck(fn(self._h, int(root), cap, C.byref(nt), *[_dp(out[k]) for k in names], gid.ctypes.data_as(C.POINTER(C.c_int64))))
        if self.rank != root:
            return None
        res = {k: v[:nt.value] for k, v in out.items()}
        res["gid"] = gid[:nt.value]
        return res

    def stats(self) -> HaloStats:
        s = HaloStats()
        self._ck(self.lib.sph_halo_get_stats(self._h, C.byref(s)))
        return s

    def selftest(self, count: int = 4096):
        self._ck(self.lib.sph_halo_selftest(self._h, int(count)))
