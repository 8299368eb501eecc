"""ctypes binding of the C ABI (include/summersph.h) for the Python harness (tests, bench).

This is plumbing only: every call goes straight to libsummersph_hip.so.  There is no fallback;
if the library is missing or no GPU is present the constructors raise.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libsummersph_hip.so")
LIB_PATH = os.environ.get("SUMMERSPH_LIB", LIB_PATH)      # A/B builds of the same ABI (profiles/)
_D = C.POINTER(C.c_double)

FIELDS = ["x", "y", "z", "vx", "vy", "vz", "u", "m", "alpha", "rho", "P", "c", "ax", "ay", "az", "du", "dalpha", "h", "omega"]
KERNELS = ["grid", "nlist", "density", "forces", "sinkacc", "kick", "drift", "dt", "leaf", "update_h", "gravity", "grav_walk", "reflag"]
FLAG_REUSE_DENSITY = 1
FLAG_VARIABLE_H = 2
FLAG_NO_LDS_TILES = 4
FLAG_SELF_GRAVITY = 16
FLAG_ACCRETE_CULL = 32
FLAG_SINK_CREATION = 64
FLAG_NO_WHOLE_TILE = 128
FLAG_REUSE_GRAVITY = 256
FLAG_NO_REFLAG = 512

# every symbol include/summersph.h declares (tests check that the library exports them all)
SYMBOLS = [
    "sph_set_sink_radii", "sph_accrete_and_cull", "sph_accrete_and_cull_keep",
    "sph_params_default", "sph_params_default_variable", "sph_upload_field", "sph_upload_field_dev", "sph_update_h",
    "sph_ctx_create", "sph_ctx_destroy", "sph_strerror", "sph_last_error", "sph_abi_version", "sph_get_params",
    "sph_upload", "sph_upload_dev", "sph_set_sinks", "sph_get_sinks", "sph_count", "sph_sink_count", "sph_check_sink_creation", "sph_get_sink_radii", "sph_sink_candidate_dev", "sph_add_sink_checked_dev",
    "sph_density", "sph_forces", "sph_kick", "sph_drift", "sph_next_dt", "sph_step", "sph_run",
    "sph_download_field", "sph_download_field_dev", "sph_download_state",
    "sph_gather_fields_dev", "sph_scatter_fields_dev",
    "sph_set_owned", "sph_set_rank", "sph_scatter_field_dev", "sph_refresh_eos", "sph_refresh_eos_ghosts", "sph_dt_candidate", "sph_set_sink_accel",
    "sph_set_stream", "sph_reserve", "sph_owned_bbox", "sph_select_boxes", "sph_selected_ids_dev", "sph_select_boxes_async", "sph_selected_counts", "sph_gather_selected_dev", "sph_replace_ghosts_dev",
    "sph_set_dt", "sph_get_dt", "sph_kick_devdt", "sph_drift_devdt", "sph_dt_candidate_dev", "sph_kick_drift_devdt", "sph_kick_dt_candidate_dev", "sph_kick_dt_candidate_gas_dev", "sph_kick_sinks_devdt", "sph_pack_partials_dev", "sph_pack_partials_ex_dev",
    "sph_apply_partials_dev", "sph_set_boundary_boxes", "sph_forces_part", "sph_set_gravity_sources_dev", "sph_accrete_mark_dev", "sph_accrete_apply_dev", "sph_set_numbers_dev",
    "sph_get_stats", "sph_get_bbox", "sph_timing_enable", "sph_timing_stride", "sph_timing_reset", "sph_timing_get", "sph_synchronize", "sph_stream",
]


class Params(C.Structure):
    _fields_ = [("h", C.c_double), ("gamma", C.c_double), ("gamma_m1", C.c_double), ("nq", C.c_int32),
                ("flags", C.c_int32), ("kernel_pi", C.c_double), ("visc_eps", C.c_double),
                ("alpha_floor", C.c_double), ("alpha_decay", C.c_double), ("G", C.c_double),
                ("dt_scale", C.c_double), ("dt_max", C.c_double), ("dt_min", C.c_double),
                ("bounding_size", C.c_double), ("eta", C.c_double), ("h_tol", C.c_double),
                ("h_max_length", C.c_double), ("h_min_length", C.c_double), ("h_iter_cap", C.c_double),
                ("theta", C.c_double)]


class Stats(C.Structure):
    _fields_ = [("n", C.c_int64), ("n_cells", C.c_int64), ("grid_dim", C.c_int32 * 3),
                ("nlist_capacity", C.c_int32), ("nlist_max", C.c_int32), ("tile_fit_pct", C.c_int32), ("nlist_mean", C.c_double),
                ("grid_builds", C.c_int64), ("nlist_builds", C.c_int64), ("density_passes", C.c_int64),
                ("force_passes", C.c_int64), ("device_bytes", C.c_int64), ("nlist_wave_mean", C.c_double),
                ("tile_fit_pct_forces", C.c_int32), ("host_syncs", C.c_int32), ("lane_efficiency_forces", C.c_double), ("nlist_reflags", C.c_int64)]


class SphError(RuntimeError):
    def __init__(self, status, text):
        super().__init__(f"summersph status {status}: {text}")
        self.status = status


_lib = None


def load():
    """Loads libsummersph_hip.so (built by __graft_entry__.build() / make -C summersph_amd/csrc)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
    # One HIP runtime per process: PyTorch bundles its own libamdhip64 (same SONAME,
    # libamdhip64.so.7, as /opt/rocm's).  Importing torch first makes the dynamic loader bind
    # our library to that already-loaded runtime, so torch tensors (device memory, RCCL via
    # torch.distributed) and this library share one set of devices and streams.
    try:
        import torch  # noqa: F401
    except Exception:  # torch absent: the system runtime in /opt/rocm/lib is used
        pass
    lib = C.CDLL(LIB_PATH)
    lib.sph_strerror.restype = C.c_char_p
    lib.sph_last_error.restype = C.c_char_p
    lib.sph_last_error.argtypes = [C.c_void_p]
    lib.sph_count.restype = C.c_int64
    lib.sph_count.argtypes = [C.c_void_p]
    lib.sph_sink_count.restype = C.c_int32
    lib.sph_sink_count.argtypes = [C.c_void_p]
    lib.sph_check_sink_creation.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
    lib.sph_get_sink_radii.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
    lib.sph_sink_candidate_dev.argtypes = [C.c_void_p, C.c_void_p]
    lib.sph_add_sink_checked_dev.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]
    lib.sph_stream.restype = C.c_void_p
    lib.sph_stream.argtypes = [C.c_void_p]
    lib.sph_ctx_create.argtypes = [C.POINTER(Params), C.c_int, C.POINTER(C.c_void_p)]
    lib.sph_ctx_destroy.argtypes = [C.c_void_p]
    lib.sph_upload.argtypes = [C.c_void_p, C.c_int64] + [C.c_void_p] * 9
    lib.sph_upload_dev.argtypes = [C.c_void_p, C.c_int64] + [C.c_void_p] * 9
    lib.sph_set_sinks.argtypes = [C.c_void_p, C.c_int32] + [C.c_void_p] * 7
    lib.sph_get_sinks.argtypes = [C.c_void_p, C.c_int32] + [C.c_void_p] * 10
    for f in ("sph_density", "sph_forces", "sph_synchronize", "sph_timing_reset"):
        getattr(lib, f).argtypes = [C.c_void_p]
    lib.sph_kick.argtypes = [C.c_void_p, C.c_double]
    lib.sph_drift.argtypes = [C.c_void_p, C.c_double]
    lib.sph_next_dt.argtypes = [C.c_void_p, _D]
    lib.sph_step.argtypes = [C.c_void_p, _D, _D]
    lib.sph_run.argtypes = [C.c_void_p, C.c_int32, _D, _D]
    lib.sph_download_field.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int64]
    lib.sph_download_field_dev.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int64]
    lib.sph_get_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
    lib.sph_get_bbox.argtypes = [C.c_void_p, _D, _D]
    lib.sph_upload_field.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int64]
    lib.sph_upload_field_dev.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int64]
    lib.sph_update_h.argtypes = [C.c_void_p]
    lib.sph_set_sink_radii.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
    lib.sph_accrete_and_cull.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
    lib.sph_set_owned.argtypes = [C.c_void_p, C.c_int64]
    lib.sph_set_rank.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
    lib.sph_scatter_field_dev.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_void_p]
    lib.sph_gather_fields_dev.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.c_int64, C.c_void_p, C.c_void_p]
    lib.sph_scatter_fields_dev.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.c_int64, C.c_int64, C.c_void_p]
    lib.sph_refresh_eos.argtypes = [C.c_void_p]
    lib.sph_refresh_eos_ghosts.argtypes = [C.c_void_p]
    lib.sph_dt_candidate.argtypes = [C.c_void_p, _D]
    lib.sph_set_sink_accel.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.sph_set_stream.argtypes = [C.c_void_p, C.c_void_p]
    lib.sph_reserve.argtypes = [C.c_void_p, C.c_int64]
    lib.sph_owned_bbox.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.sph_select_boxes.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
    lib.sph_selected_ids_dev.argtypes = [C.c_void_p, C.c_int32, C.c_int64, C.c_void_p]
    lib.sph_replace_ghosts_dev.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
    lib.sph_set_dt.argtypes = [C.c_void_p, C.c_double, C.c_double]
    lib.sph_get_dt.argtypes = [C.c_void_p, _D, _D]
    for f in ("sph_kick_devdt", "sph_drift_devdt", "sph_dt_candidate_dev", "sph_kick_drift_devdt", "sph_kick_dt_candidate_dev",
              "sph_kick_dt_candidate_gas_dev", "sph_kick_sinks_devdt"):
        getattr(lib, f).argtypes = [C.c_void_p]
    lib.sph_pack_partials_dev.argtypes = [C.c_void_p, C.c_void_p]
    lib.sph_pack_partials_ex_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
    lib.sph_select_boxes_async.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
    lib.sph_selected_counts.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
    lib.sph_gather_selected_dev.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]
    lib.sph_apply_partials_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32]
    lib.sph_set_boundary_boxes.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
    lib.sph_forces_part.argtypes = [C.c_void_p, C.c_int32]
    lib.sph_set_gravity_sources_dev.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    lib.sph_accrete_mark_dev.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
    lib.sph_set_numbers_dev.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]
    lib.sph_accrete_apply_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.POINTER(C.c_int64)]
    lib.sph_timing_enable.argtypes = [C.c_void_p, C.c_int]
    lib.sph_timing_stride.argtypes = [C.c_void_p, C.c_int]
    lib.sph_timing_get.argtypes = [C.c_void_p, C.c_int, _D, C.POINTER(C.c_int64)]
    _lib = lib
    return lib


def default_params(variable: bool = False) -> Params:
    p = Params()
    if variable:
        load().sph_params_default_variable(C.byref(p))
    else:
        load().sph_params_default(C.byref(p))
    return p


def _hp(a):
    if a is None:
        return None
    assert isinstance(a, np.ndarray) and a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data


class Context:
    """Thin object wrapper over an sph_ctx*.  Arrays are float64 numpy (host) unless a method says _dev."""

    def __init__(self, params: Params | None = None, device: int = 0, variable: bool = False, **overrides):
        self.lib = load()
        p = params if params is not None else default_params(variable)
        for k, v in overrides.items():
            setattr(p, k, v)
        self.params = p
        h = C.c_void_p()
        st = self.lib.sph_ctx_create(C.byref(p), int(device), C.byref(h))
        if st != 0:
            raise SphError(st, self.lib.sph_strerror(st).decode())
        self._h = h

    def _ck(self, st):
        if st != 0:
            raise SphError(st, self.lib.sph_strerror(st).decode() + " -- " + self.lib.sph_last_error(self._h).decode())

    def close(self):
        if getattr(self, "_h", None):
            self.lib.sph_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- state -------------------------------------------------------------------------
    @property
    def n(self) -> int:
        return int(self.lib.sph_count(self._h))

    def upload(self, gas: dict):
        arrs = [np.ascontiguousarray(gas[k], dtype=np.float64) for k in "x y z vx vy vz u m".split()]
        al = gas.get("alpha")
        al = None if al is None else np.ascontiguousarray(al, dtype=np.float64)
        self._ck(self.lib.sph_upload(self._h, arrs[0].size, *[_hp(a) for a in arrs], _hp(al)))
        if gas.get("h") is not None and (self.params.flags & FLAG_VARIABLE_H):
            self.upload_field("h", gas["h"])

    def upload_field(self, name: str, values):
        a = np.ascontiguousarray(values, dtype=np.float64)
        self._ck(self.lib.sph_upload_field(self._h, FIELDS.index(name), _hp(a), a.size))

    def upload_field_dev(self, name: str, dev_ptr: int, n: int):
        self._ck(self.lib.sph_upload_field_dev(self._h, FIELDS.index(name), C.c_void_p(int(dev_ptr)), int(n)))

    def update_h(self):
        self._ck(self.lib.sph_update_h(self._h))

    def upload_dev(self, n: int, ptrs):
        """ptrs: 9 device addresses (ints; alpha may be 0/None), e.g. torch tensors' data_ptr()"""
        self._ck(self.lib.sph_upload_dev(self._h, int(n), *[C.c_void_p(int(p) if p else 0) for p in ptrs]))

    def set_sinks(self, sinks: dict):
        arrs = [np.ascontiguousarray(sinks[k], dtype=np.float64) for k in "x y z vx vy vz m".split()]
        self._ck(self.lib.sph_set_sinks(self._h, arrs[0].size, *[_hp(a) for a in arrs]))
        self.ns = int(arrs[0].size)
        if sinks.get("radius") is not None:
            r = np.ascontiguousarray(sinks["radius"], dtype=np.float64)
            self._ck(self.lib.sph_set_sink_radii(self._h, r.size, _hp(r)))

    def accrete_and_cull(self) -> int:
        r = C.c_int64(0)
        self._ck(self.lib.sph_accrete_and_cull(self._h, C.byref(r)))
        return int(r.value)

    def get_sinks(self) -> dict:
        ns = self.ns = int(self.lib.sph_sink_count(self._h))      # check_sink_creation may have added one
        out = {k: np.zeros(ns) for k in "x y z vx vy vz m ax ay az".split()}
        self._ck(self.lib.sph_get_sinks(self._h, ns, *[_hp(out[k]) for k in "x y z vx vy vz m ax ay az".split()]))
        out["radius"] = np.zeros(ns)
        self._ck(self.lib.sph_get_sink_radii(self._h, ns, _hp(out["radius"])))
        return out

    def sink_candidate_dev(self, dev_ptr: int):
        self._ck(self.lib.sph_sink_candidate_dev(self._h, C.c_void_p(int(dev_ptr))))

    def add_sink_checked_dev(self, dev_ptr: int) -> bool:
        cr = C.c_int32(0)
        self._ck(self.lib.sph_add_sink_checked_dev(self._h, C.c_void_p(int(dev_ptr)), C.byref(cr)))
        return bool(cr.value)

    def check_sink_creation(self) -> bool:
        cr = C.c_int32(0)
        self._ck(self.lib.sph_check_sink_creation(self._h, C.byref(cr)))
        return bool(cr.value)

    # ---- hot path ------------------------------------------------------------------------
    def density(self):
        self._ck(self.lib.sph_density(self._h))

    def forces(self):
        self._ck(self.lib.sph_forces(self._h))

    def kick(self, dt: float):
        self._ck(self.lib.sph_kick(self._h, float(dt)))

    def drift(self, dt: float):
        self._ck(self.lib.sph_drift(self._h, float(dt)))

    def next_dt(self, dt: float) -> float:
        d = C.c_double(dt)
        self._ck(self.lib.sph_next_dt(self._h, C.byref(d)))
        return d.value

    def step(self, dt: float, t: float = 0.0):
        d, tt = C.c_double(dt), C.c_double(t)
        self._ck(self.lib.sph_step(self._h, C.byref(d), C.byref(tt)))
        return d.value, tt.value

    def run(self, nsteps: int, dt: float, t: float = 0.0):
        d, tt = C.c_double(dt), C.c_double(t)
        self._ck(self.lib.sph_run(self._h, int(nsteps), C.byref(d), C.byref(tt)))
        return d.value, tt.value

    # ---- multi-GPU building blocks ---------------------------------------------------------
    def set_owned(self, n_owned: int):
        self._ck(self.lib.sph_set_owned(self._h, int(n_owned)))

    def set_rank(self, rank: int, nranks: int):
        self._ck(self.lib.sph_set_rank(self._h, int(rank), int(nranks)))

    def scatter_field_dev(self, name: str, first: int, count: int, dev_ptr: int):
        self._ck(self.lib.sph_scatter_field_dev(self._h, FIELDS.index(name), int(first), int(count), C.c_void_p(int(dev_ptr))))

    def gather_fields_dev(self, names, count: int, ids_ptr: int, out_ptr: int):
        """out[f, k] = field names[f] of original id ids[k] (ids_ptr 0: ids 0..count-1); device pointers"""
        f = (C.c_int32 * len(names))(*[FIELDS.index(n) for n in names])
        self._ck(self.lib.sph_gather_fields_dev(self._h, len(names), f, int(count), C.c_void_p(int(ids_ptr) or None),
                                                C.c_void_p(int(out_ptr))))

    def scatter_fields_dev(self, names, first: int, count: int, vals_ptr: int):
        f = (C.c_int32 * len(names))(*[FIELDS.index(n) for n in names])
        self._ck(self.lib.sph_scatter_fields_dev(self._h, len(names), f, int(first), int(count), C.c_void_p(int(vals_ptr))))

    # ---- the device-resident multi-GPU exchange (include/summersph.h, "kept on the device") ----
    def set_stream(self, stream_handle: int):
        self._ck(self.lib.sph_set_stream(self._h, C.c_void_p(stream_handle)))

    def reserve(self, n_slots: int):
        self._ck(self.lib.sph_reserve(self._h, n_slots))

    def owned_bbox(self, dev_ptr: int = 0) -> np.ndarray | None:
        """dev_ptr == 0: returns min xyz, max xyz on the host; else writes them to device memory (no sync)"""
        if dev_ptr:
            self._ck(self.lib.sph_owned_bbox(self._h, None, C.c_void_p(dev_ptr)))
            return None
        out = np.empty(6)
        self._ck(self.lib.sph_owned_bbox(self._h, out.ctypes.data, None))
        return out

    def select_boxes(self, boxes: np.ndarray) -> np.ndarray:
        boxes = np.ascontiguousarray(boxes, dtype=np.float64).reshape(-1, 6)
        counts = np.zeros(boxes.shape[0], dtype=np.int64)
        self._ck(self.lib.sph_select_boxes(self._h, boxes.shape[0], boxes.ctypes.data, counts.ctypes.data))
        return counts

    def select_boxes_async(self, boxes: np.ndarray):
        """the same selection without waiting for its counts (read them with selected_counts after a synchronisation)"""
        boxes = np.ascontiguousarray(boxes, dtype=np.float64).reshape(-1, 6)
        self._ck(self.lib.sph_select_boxes_async(self._h, boxes.shape[0], boxes.ctypes.data))
        return boxes.shape[0]

    def selected_counts(self, nbox: int) -> np.ndarray:
        counts = np.zeros(nbox, dtype=np.int64)
        self._ck(self.lib.sph_selected_counts(self._h, int(nbox), counts.ctypes.data))
        return counts

    def gather_selected_dev(self, box: int, names, capacity: int, out_ptr: int):
        """out[0] = count, out[1] = 0, out[2 + f*count + k] (if count <= capacity) for selection `box`; device pointer"""
        f = (C.c_int32 * len(names))(*[FIELDS.index(n) for n in names])
        self._ck(self.lib.sph_gather_selected_dev(self._h, int(box), len(names), f, int(capacity), C.c_void_p(int(out_ptr))))

    def selected_ids_dev(self, box: int, count: int, dev_ptr: int):
        self._ck(self.lib.sph_selected_ids_dev(self._h, box, count, C.c_void_p(dev_ptr)))

    def replace_ghosts_dev(self, count: int, dev_ptr: int):
        self._ck(self.lib.sph_replace_ghosts_dev(self._h, count, C.c_void_p(dev_ptr)))

    def set_dt(self, dt: float, t: float = 0.0):
        self._ck(self.lib.sph_set_dt(self._h, dt, t))

    def get_dt(self):
        dt, t = C.c_double(0.0), C.c_double(0.0)
        self._ck(self.lib.sph_get_dt(self._h, C.byref(dt), C.byref(t)))
        return dt.value, t.value

    def kick_devdt(self):
        self._ck(self.lib.sph_kick_devdt(self._h))

    def drift_devdt(self):
        self._ck(self.lib.sph_drift_devdt(self._h))

    def dt_candidate_dev(self):
        self._ck(self.lib.sph_dt_candidate_dev(self._h))

    def kick_drift_devdt(self):
        self._ck(self.lib.sph_kick_drift_devdt(self._h))

    def kick_dt_candidate_dev(self):
        self._ck(self.lib.sph_kick_dt_candidate_dev(self._h))

    def kick_dt_candidate_gas_dev(self):
        self._ck(self.lib.sph_kick_dt_candidate_gas_dev(self._h))

    def kick_sinks_devdt(self):
        self._ck(self.lib.sph_kick_sinks_devdt(self._h))

    def pack_partials_dev(self, dev_ptr: int, predict_box: bool = True):
        self._ck(self.lib.sph_pack_partials_ex_dev(self._h, C.c_void_p(dev_ptr), 1 if predict_box else 0))

    def apply_partials_dev(self, dev_ptr: int, nranks: int, stride: int, apply_dt: bool):
        self._ck(self.lib.sph_apply_partials_dev(self._h, C.c_void_p(dev_ptr), nranks, stride, 1 if apply_dt else 0))

    def set_boundary_boxes(self, boxes: np.ndarray):
        boxes = np.ascontiguousarray(boxes, dtype=np.float64).reshape(-1, 6)
        self._ck(self.lib.sph_set_boundary_boxes(self._h, boxes.shape[0], boxes.ctypes.data if boxes.size else None))

    def forces_part(self, part: int):
        """1: sink gravity + the wavefronts that cannot see a ghost; 2: the others (after refresh_eos)"""
        self._ck(self.lib.sph_forces_part(self._h, part))

    def set_gravity_sources_dev(self, n_src: int, dev_ptr: int, lo_hi):
        """n_src records {x,y,z,m} at dev_ptr (kept alive by the caller) + their bounding box; n_src = 0 resets"""
        box = np.ascontiguousarray(lo_hi, dtype=np.float64) if n_src else np.zeros(6)
        self._ck(self.lib.sph_set_gravity_sources_dev(self._h, int(n_src), C.c_void_p(int(dev_ptr)) if n_src else None,
                                                      box.ctypes.data))

    def set_numbers_dev(self, first: int, count: int, dev_ptr: int):
        """global particle numbers (int64 on the device) of original ids [first, first + count)"""
        self._ck(self.lib.sph_set_numbers_dev(self._h, int(first), int(count), C.c_void_p(int(dev_ptr)) if count else None))

    def accrete_mark_dev(self, src_offset: int, partials_ptr: int):
        self._ck(self.lib.sph_accrete_mark_dev(self._h, int(src_offset), C.c_void_p(int(partials_ptr))))

    def accrete_apply_dev(self, all_ptr: int, nranks: int, stride: int, keep_ptr: int = 0) -> int:
        r = C.c_int64(0)
        self._ck(self.lib.sph_accrete_apply_dev(self._h, C.c_void_p(int(all_ptr)), nranks, stride,
                                                C.c_void_p(int(keep_ptr)) if keep_ptr else None, C.byref(r)))
        return int(r.value)

    def refresh_eos(self, ghosts_only: bool = False):
        self._ck((self.lib.sph_refresh_eos_ghosts if ghosts_only else self.lib.sph_refresh_eos)(self._h))

    def dt_candidate(self) -> float:
        d = C.c_double(0)
        self._ck(self.lib.sph_dt_candidate(self._h, C.byref(d)))
        return d.value

    def set_sink_accel(self, ax, ay, az):
        a = [np.ascontiguousarray(v, dtype=np.float64) for v in (ax, ay, az)]
        self._ck(self.lib.sph_set_sink_accel(self._h, a[0].size, *[_hp(v) for v in a]))

    # ---- read-back -----------------------------------------------------------------------
    def field(self, name: str) -> np.ndarray:
        out = np.zeros(self.n)
        self._ck(self.lib.sph_download_field(self._h, FIELDS.index(name), _hp(out), out.size))
        return out

    def field_dev(self, name: str, dev_ptr: int, n: int):
        self._ck(self.lib.sph_download_field_dev(self._h, FIELDS.index(name), C.c_void_p(int(dev_ptr)), int(n)))

    def state(self) -> dict:
        return {k: self.field(k) for k in FIELDS[:9]}

    # ---- diagnostics ---------------------------------------------------------------------
    def stats(self) -> Stats:
        s = Stats()
        self._ck(self.lib.sph_get_stats(self._h, C.byref(s)))
        return s

    def bbox(self):
        lo = (C.c_double * 3)(); hi = (C.c_double * 3)()
        self._ck(self.lib.sph_get_bbox(self._h, lo, hi))
        return np.array(lo[:]), np.array(hi[:])

    def timing(self, on, only=None, stride=1):
        """HIP events around every kernel group (on=True), none (False), or only the groups named in `only`; stride: bracket
        only every stride-th launch of a timed group (a sample: timing_get returns the bracketed launches)"""
        mask = 0 if not on else (1 if only is None else sum(2 << KERNELS.index(k) for k in only))
        self._ck(self.lib.sph_timing_stride(self._h, max(int(stride), 1)))
        self._ck(self.lib.sph_timing_enable(self._h, mask))

    def timing_reset(self):
        self._ck(self.lib.sph_timing_reset(self._h))

    def timing_get(self, kernel: str):
        ms, cnt = C.c_double(0), C.c_int64(0)
        self._ck(self.lib.sph_timing_get(self._h, KERNELS.index(kernel), C.byref(ms), C.byref(cnt)))
        return ms.value, cnt.value

    def synchronize(self):
        self._ck(self.lib.sph_synchronize(self._h))

    def stream(self) -> int:
        return int(self.lib.sph_stream(self._h) or 0)
