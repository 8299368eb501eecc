#!/usr/bin/env python3
"""bench.py -- particle-steps/s of the SPH hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 10 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one iteration of the reference's simulate() loop body (SUMMER_SPH.f90:889-916):
2 density passes + 2 force passes + 2 half-kicks + 1 drift + the dt reduction (variable-h: + the
h update, Variable.f90:1152), on a seeded synthetic Keplerian disc (summersph_amd/ic.py).  Inputs
are resident in HBM before the timed region starts.  Rank 0 prints ONE JSON line.

Workloads
  fixed (headline, every N)  uniform disc, N x 1e6 gas particles + 1 sink, fixed h = 2.5: the [F] path
                             (BASELINE configs[1] shape at the metric's N = 1e6); WEAK scaling: shards over GPUs as
                             x-slabs with ghost exchange + migration over RCCL -- the native step loop of
                             libsummersph_halo.so where it applies (--halo auto), else summersph_amd/dist.py.
  ring4m_strong (every N)    BASELINE configs[3]: the 4e6-particle thin viscous ring split N ways (STRONG scaling; the
                             >= 6x at 8 GPUs of BASELINE.md is quoted on this).  Side object, own timed region.
  full_1e7 (every N)         BASELINE configs[4]: 1e7-particle disc + sink, Barnes-Hut self-gravity, accretion + cull --
                             simulate() as the reference runs it, split N ways (STRONG scaling).  Side object.
  variable_h, full_simulate, side_records, fixed_reuse_density (N = 1)
                             BASELINE configs[2] (1e6, per-particle h, grad-h, leaf-box rule, h update), the reference's
                             whole loop body at 1e6, and the kernel selection away from the friendly geometry.

Every figure can be recomputed from the line itself or from a file under profiles/:
  roofline     dominant kernel of a workload: ALGORITHMIC HBM bytes per launch / its mean launch
               duration, measured live with HIP events on the library's own stream inside the timed steps; peak 8 TB/s.
               `traffic` (PMC bytes per launch) and `limiter` (TA / vector-issue / LDS occupancy) are quoted from the newest
               profiles/rNN_limiters.json; that file records the sha256 of the kernel sources it was profiled on, and the
               object says `"stale": true` when summersph_amd/csrc has changed since.
  config       mean_neighbours etc. are read from the context directly after the timed region (the state the timed steps
               ran on), before any repeat or breakdown pass.
  repeat_ms_per_step   restarts of the SAME state (same upload, same warm-up, same K steps): run-to-run spread.
  cpu_baseline the UNMODIFIED reference (oracle/_ref/ref_driver: /root/reference's module compiled by oracle/build_ref.sh,
               serial as the reference is) timed on this box's host cores on a bounded sample of the same workload
               (kind "reference"); the CPU restatement (oracle/sph_oracle.c, OpenMP) is reported beside it; if the
               binary is absent the restatement is the baseline (kind "port").  Rank 0, N = 1 only.
"""
import argparse
import glob
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic HBM bytes per particle per pass
BYTES = {"density": 40 + 8, "forces": 80 + 40, "kick": 80 + 40, "drift": 48 + 24}
BYTES_PER_STEP = 2 * (BYTES["density"] + BYTES["forces"]) + 2 * BYTES["kick"] + BYTES["drift"]   # 648
BYTES_VAR = {"density": 48 + 16, "forces": 96 + 40}                                               # + h in, Omega out
BYTES_PER_STEP_VAR = 2 * (BYTES_VAR["density"] + BYTES_VAR["forces"]) + 2 * BYTES["kick"] + BYTES["drift"] + 32   # ~744
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_PEAK_TFLOPS = 78.6    # vector fp64
# fp64 operations per pair visit as written in the kernels (div/sqrt counted as 1 each)
FLOPS_DENSITY_PAIR, FLOPS_FORCE_PAIR = 22, 75


def csrc_sha256():
    """fingerprint of the kernel sources (what a counter pass was taken on): sha256 over csrc/*.hip, *.hpp and include/*.h"""
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "summersph_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "summersph_amd", "csrc", "*.hpp"))
                   + glob.glob(os.path.join(ROOT, "include", "*.h")))
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def load_limiters():
    """newest profiles/rNN_limiters.json -> (dict, file name, stale?)"""
    cand = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_limiters.json")))
    if not cand:
        return {}, None, None
    try:
        d = json.load(open(cand[-1]))
    except (OSError, ValueError):
        return {}, None, None
    sha = d.get("_meta", {}).get("csrc_sha256")
    return d, os.path.relpath(cand[-1], ROOT), (sha != csrc_sha256())


# Inside a timed region the dominant kernel group is bracketed by HIP events -- every EVENT_STRIDE-th launch of it: an event pair
# costs the stream ~14 us (measured: 1.19 ms per step with every force launch bracketed, 1.13-1.14 without any), and the
# roofline's duration is a mean over launches anyway.  Odd, so that with N > 1 (interior and boundary launches alternate) both kinds
# are sampled.
EVENT_STRIDE = int(os.environ.get("SPH_BENCH_EVENT_STRIDE", "5"))

def roofline_record(lim, workload, prefixes, alg_bytes, avg_launch_s, launches, lane_eff, n, default_n=1_000_000):
    """roofline object of one kernel: algorithmic bytes / measured duration against the HBM peak, plus what the committed
    counter passes say limits it (only when they were taken on this workload size)"""
    limiters, lim_file, stale = lim
    achieved = alg_bytes / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0
    rec, name = None, None
    if n == default_n:
        for k, v in limiters.get(workload, {}).get("kernels", {}).items():
            if any(k.startswith(p) for p in prefixes) and (rec is None or v.get("total_ms", 0) > rec.get("total_ms", 0)):
                rec, name = v, k
    limiter = {"lane_efficiency": lane_eff}
    bound = "unmeasured (no counter pass for this workload size)"
    if rec:
        for key in ("ta_busy", "valu_issue", "valu_active", "lds_busy", "clock_GHz_est"):
            if key in rec:
                limiter[key] = rec[key]
        limiter["hbm_frac_algorithmic"] = achieved / HBM_PEAK_GBS
        cand = {"texture addresser (divergent gathers)": rec.get("ta_busy", 0.0), "fp64 vector issue": rec.get("valu_issue", 0.0),
                "LDS": rec.get("lds_busy", 0.0), "hbm": (rec.get("traffic_bytes_per_launch", 0.0) / avg_launch_s / 1e9 / HBM_PEAK_GBS) if avg_launch_s > 0 else 0.0}
        bound = max(cand, key=cand.get)
        limiter["source"] = f"{lim_file} [{workload}][{name}] (profiles/profile.sh, rocprofv3 counter passes)"
        limiter["profiled_git"] = limiters.get("_meta", {}).get("git")
        limiter["stale"] = bool(stale)          # true: summersph_amd/csrc changed since the counter passes were taken
        limiter["avg_launch_us_rocprof"] = rec.get("avg_us")
    return {"bound": bound, "kernel": name or prefixes[0], "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": rec.get("traffic_bytes_per_launch") if rec else None,
            "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": avg_launch_s * 1e3, "launches": launches,
            "launches_bracketed": f"every {EVENT_STRIDE}th launch of the timed region carries the HIP-event pair", "limiter": limiter}


def usable_cores():
    """cores this process may really use: the affinity mask capped by the cgroup CPU quota (a GPU box hands a
    one-GPU job a share of its host cores; more threads than that only oversubscribe)"""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return n


def reference_baseline(nngb, seconds_target=12.0):
    """the unmodified reference (oracle/_ref/ref_driver, built from /root/reference by oracle/build_ref.sh; the binary
    travels to the GPU box, the sources do not) timed on one host core -- it is a serial program -- on a bounded sample of
    the headline workload: `sph` = the loop body without Barnes-Hut gravity (what the headline measures), and the same
    sample through find_forces as the reference has it (`full`).  None when the binary is absent or fails."""
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_driver")
    if not os.access(exe, os.X_OK):
        return None
    from summersph_amd import ic, txtio

    def run(n, steps, variant, d):
        icf = os.path.join(d, f"ic_{n}.txt")
        if not os.path.exists(icf):
            txtio.write_ic(icf, ic.keplerian_disc(n, seed=2, nngb=nngb))
        r = subprocess.run([exe, "time", icf, os.path.join(d, "o.bin"), str(steps), variant], capture_output=True, text=True, timeout=600)
        m = re.search(r"loop_seconds\s+([0-9.eE+-]+)\s+particles\s+(\d+)\s+steps\s+(\d+)", r.stdout)
        if r.returncode != 0 or not m:
            raise RuntimeError(f"ref_driver failed: rc {r.returncode} {r.stdout[-200:]} {r.stderr[-200:]}")
        return float(m.group(1)), int(m.group(2)), int(m.group(3))

    try:
        with tempfile.TemporaryDirectory() as d:
            s, n, k = run(4000, 1, "sph", d)                       # calibration
            rate = n * k / s
            n_s = int(min(100_000, max(8000, rate * seconds_target / 3)))
            steps = 3
            s, n, k = run(n_s, steps, "sph", d)
            sf, nf, kf = run(n_s, 1, "full", d)
        return {"value": n * k / s, "unit": "particle-steps/s", "cores": 1, "kind": "reference",
                "sample": f"{k} steps of the reference's loop body without gas self-gravity (zero_rates, sink_gravforces, get_SPH; "
                          f"2 tree builds, 2 density + 2 force passes, kick/drift/dt) on a {n}-particle disc of the headline's "
                          f"surface density, 1 core (the reference is serial), {s:.1f} s; binary oracle/_ref/ref_driver = "
                          f"/root/reference/SUMMER_SPH.f90:1-931 compiled unmodified (amdflang -O2) + oracle/ref_driver.f90",
                "full_simulate_value": nf * kf / sf,
                "full_simulate_sample": f"{kf} step with find_forces as the reference has it (Barnes-Hut gas self-gravity, accretion, cull), same disc, {sf:.1f} s"}
    except Exception as e:       # noqa: BLE001 -- a baseline that cannot run must not take the bench line with it
        print(f"[bench] reference baseline unavailable: {type(e).__name__}: {e}", file=sys.stderr, flush=True)
        return None


def port_baseline(n_total, nngb, seconds_target=10.0):
    """CPU restatement (oracle/sph_oracle.c) on the host cores of this box, bounded sample"""
    from oracle import orc
    from summersph_amd import ic
    threads = max(1, min(orc.max_threads(), usable_cores()))
    # calibrate on a small disc of the same surface density, then size the sample
    rows = ic.keplerian_disc(20000, seed=1, nngb=nngb)
    gas, sinks = ic.split_rows(rows)
    o = orc.Oracle(gas, sinks, nthreads=threads)
    t0 = time.perf_counter(); o.step(1e-2); t1 = time.perf_counter()
    rate = 20000 / (t1 - t0)
    n_s = int(min(n_total, max(20000, rate * seconds_target)))
    rows = ic.keplerian_disc(n_s, seed=2, nngb=nngb)
    gas, sinks = ic.split_rows(rows)
    o = orc.Oracle(gas, sinks, nthreads=threads)
    k = int(max(1, min(8, rate * seconds_target / n_s)))
    dt = 1e-2
    t0 = time.perf_counter()
    for _ in range(k):
        dt = o.step(dt)
    t1 = time.perf_counter()
    # one thread, on the calibration-size disc
    rows = ic.keplerian_disc(20000, seed=1, nngb=nngb)
    gas, sinks = ic.split_rows(rows)
    o1 = orc.Oracle(gas, sinks, nthreads=1)
    s0 = time.perf_counter(); o1.step(1e-2); s1 = time.perf_counter()
    return {"value": k * n_s / (t1 - t0), "unit": "particle-steps/s", "cores": threads, "kind": "port",
            "sample": f"{k} full step(s) (2 density + 2 force passes, kick/drift/dt each) of a {n_s}-particle disc of "
                      f"the same surface density, OpenMP x{threads}, {t1 - t0:.1f} s",
            "single_thread_value": 20000 / (s1 - s0),
            "single_thread_sample": f"1 full step of a 20000-particle disc of the same surface density, {s1 - s0:.1f} s"}


def cpu_baseline(n_total, nngb):
    ref = reference_baseline(nngb)
    port = port_baseline(n_total, nngb, seconds_target=10.0 if ref else 15.0)
    if ref is None:
        return port
    ref["port"] = port           # the CPU restatement beside it (the checker of the parity tests; OpenMP over the host cores)
    return ref


def stream_copy_gbs(torch, device, nbytes=1 << 30, reps=5):
    """device-to-device copy rate (bytes read + bytes written per second): the practical HBM ceiling of this box,
    measured in the same run"""
    src = torch.empty(nbytes // 8, dtype=torch.float64, device=f"cuda:{device}").fill_(1.0)
    dst = torch.empty_like(src)
    dst.copy_(src)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    ev0.record()
    for _ in range(reps):
        dst.copy_(src)
    ev1.record()
    torch.cuda.synchronize()
    return 2.0 * nbytes * reps / (ev0.elapsed_time(ev1) * 1e-3) / 1e9


class Workload:
    """one context on `device` and the means to put it back into its initial state (inputs stay resident in HBM)"""

    def __init__(self, capi, torch, device, gas, sinks, flags=0, variable=False):
        self.capi, self.torch, self.sinks, self.variable = capi, torch, sinks, variable
        self.n = int(gas["x"].size)
        self.ctx = capi.Context(device=device, variable=True, flags=flags | capi.FLAG_VARIABLE_H) if variable else capi.Context(device=device, flags=flags)
        self.dev = [torch.from_numpy(np.ascontiguousarray(gas[k])).to(f"cuda:{device}") for k in "x y z vx vy vz u m alpha".split()]
        self.hdev = torch.from_numpy(np.ascontiguousarray(gas["h"])).to(f"cuda:{device}") if variable else None
        torch.cuda.synchronize()
        self.reset()

    def reset(self):
        self.ctx.upload_dev(self.n, [t.data_ptr() for t in self.dev])               # inputs resident in HBM
        if self.variable:
            self.ctx.upload_field_dev("h", self.hdev.data_ptr(), self.n)
        self.ctx.set_sinks(self.sinks)

    def close(self):
        self.ctx.close()


def stats_dict(st):
    return {"mean_neighbours": st.nlist_mean, "mean_wave_trips": st.nlist_wave_mean, "max_neighbours": st.nlist_max,
            "tile_fit_pct": st.tile_fit_pct, "tile_fit_pct_forces": st.tile_fit_pct_forces, "grid": list(st.grid_dim),
            "lane_efficiency_forces": st.lane_efficiency_forces, "list_builds": st.nlist_builds, "list_reflags": st.nlist_reflags,
            "device_bytes": st.device_bytes, "slots": int(st.n)}


def timed_run(wl, steps, warmup, dominant=("forces",), breakdown=True, repeats=0, late_window=None, preheat=True):
    """-> dict(elapsed, dt, table, stats, repeat_ms, late).  Inside the timed region only the `dominant` kernel groups are
    bracketed by HIP events (the roofline's duration is measured live there; bracketing every group costs ~4 % of a
    fixed-h step).  `stats` is read directly after the timed region: it describes the state the timed steps ran on.
    `repeats`: the context is put back into its initial state (same upload) and the same warm-up + K steps run again --
    run-to-run spread of the SAME trajectory.  The per-group table comes from up to 10 further steps with every group
    bracketed, outside the timed region, scaled to `steps`; the dominant groups keep their totals from the timed region.
    late_window = (a, b): the trajectory continues and steps a..b (counted from the upload) are timed once more."""
    capi, torch, ctx = wl.capi, wl.torch, wl.ctx

    def one(keep_timing):
        dt, t = ctx.run(warmup, 1e-2, 0.0)
        if keep_timing:
            ctx.timing(True, only=list(dominant), stride=EVENT_STRIDE); ctx.timing_reset()
        ctx.synchronize(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        dt, t = ctx.run(steps, dt, t)
        ctx.synchronize(); torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if keep_timing:
            ctx.timing(False)
        return el, dt, t

    if preheat:
        # One untimed pass of the same W + K steps first, then the same upload again: the timed region of the default run is
        # ~25 ms long and, started on an idle GPU, fell into the clock ramp (measured: 1.166 ms per step against 1.13 for
        # every later pass of the same trajectory; with 30 warm-up steps the first pass gives 1.119 like the later ones).
        one(False)
        wl.reset()
    el, dt, t = one(True)
    table = {k: ctx.timing_get(k) for k in capi.KERNELS}
    for k in dominant:                                  # every EVENT_STRIDE-th launch was bracketed: the group's total, extrapolated
        table[k] = (table[k][0] * EVENT_STRIDE, table[k][1] * EVENT_STRIDE)
    st = ctx.stats()                                   # the state of the timed steps, nothing ran since
    out = {"elapsed": el, "dt": dt, "stats": st, "n_left": ctx.n, "repeat_ms": None, "late": None}
    if breakdown:
        bs = max(1, min(steps, 10))
        ctx.timing_reset(); ctx.timing(True)
        dt_b, t_b = ctx.run(bs, dt, t)
        ctx.synchronize()
        ctx.timing(False)
        for k in capi.KERNELS:
            if k not in dominant:
                ms, cnt = ctx.timing_get(k)
                table[k] = (ms * steps / bs, int(round(cnt * steps / bs)))
        done = warmup + steps + bs
        if late_window:
            a, b = late_window
            if done < a:
                dt_b, t_b = ctx.run(a - done, dt_b, t_b)
            ctx.timing_reset(); ctx.timing(True, only=["update_h", "nlist", "reflag"])
            ctx.synchronize()
            l0 = time.perf_counter()
            dt_b, t_b = ctx.run(b - a, dt_b, t_b)
            ctx.synchronize()
            lel = time.perf_counter() - l0
            ctx.timing(False)
            out["late"] = {"steps": [max(a, done), max(a, done) + (b - a)], "ms_per_step": lel / (b - a) * 1e3,
                           "kernel_ms_per_step": {k: ctx.timing_get(k)[0] / (b - a) for k in ("update_h", "nlist", "reflag")}}
    out["table"] = table
    if repeats:
        spread = []
        for _ in range(repeats):
            wl.reset()
            r_el, r_dt, _ = one(False)
            spread.append(r_el / steps * 1e3)
            if r_dt != dt:
                spread.append(f"final dt differs: {r_dt} vs {dt}")
        out["repeat_ms"] = spread
    return out


# ---- several GPUs ------------------------------------------------------------------------------------------------------

class NativeSim:
    """bench face of the native multi-GPU step loop (summersph_amd/halo.py): what DistSim offers the timed region"""

    @classmethod
    def create(cls, env, flags, mine, sinks, bounds, variable=False):
        from summersph_amd import halo
        capi, dist, torch, rank, world = env["capi"], env["dist"], env["torch"], env["rank"], env["world"]
        uid = [halo.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        self = cls()

        def agreed(ok):          # every rank must have succeeded: the next call is collective
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            return int(flag[0]) == 1

        ok = True
        try:
            self.ctx = (capi.Context(device=env["local_rank"], variable=True, flags=flags | capi.FLAG_VARIABLE_H) if variable
                        else capi.Context(device=env["local_rank"], flags=flags))
            self.h = halo.Halo.rccl(self.ctx, uid[0], rank, world)
        except Exception as e:       # noqa: BLE001 -- e.g. two ranks on one GPU: RCCL refuses
            ok = False
            print(f"[bench rank {rank}] native halo: {type(e).__name__}: {e}", file=sys.stderr, flush=True)
        if not agreed(ok):
            return None
        try:
            self.h.selftest(4096)        # every rank sends to every rank and all-gathers, payloads checked
        except Exception as e:       # noqa: BLE001
            ok = False
            print(f"[bench rank {rank}] native halo self-test: {type(e).__name__}: {e}", file=sys.stderr, flush=True)
        if not agreed(ok):
            return None
        self.ctx.set_sinks(sinks)
        self.h.set_slabs(bounds, 32)
        self.h.upload(mine)
        self.t, self.profile, self.phase_s = 0.0, False, {}
        return self

    def run(self, nsteps, dt):
        dt, self.t = self.h.run(nsteps, dt, self.t)
        return dt

    @property
    def n_owned(self):
        return self.h.n_owned

    @property
    def stats(self):
        s = self.h.stats()
        return {"ghosts": s.ghosts, "migrated": s.migrated, "exchanges": s.exchanges, "collectives": s.collectives, "migrations": s.migrations,
                "host_waits": s.host_waits, "removed": s.removed, "sinks_created": s.sinks_created}

    def close(self):
        self.h.close(); self.ctx.close()


PLANES = {"nccl": "RCCL (summersph_amd/dist.py over torch.distributed nccl)",
          "rccl(native)": "RCCL, native step loop (libsummersph_halo.so: grouped send/recv on a second HIP stream)",
          "gloo": "host-staged gloo messages (summersph_amd/dist.py) -- REHEARSAL plane, not RCCL"}


def native_applies(variable, flags, capi):
    """what libsummersph_halo.so's step loop covers (include/summersph_halo.h): every flavour of the loop body; accretion on
    several ranks needs the shared octree, i.e. self-gravity or variable h (as in the reference, which always has gravity)"""
    return variable or bool(flags & capi.FLAG_SELF_GRAVITY) or not (flags & capi.FLAG_ACCRETE_CULL)


def dist_run(env, rows, variable, flags, steps, warmup, halo_mode, dominant=("forces",), profile=False):
    """the rows' gas particles cut into one equal-count x-slab per rank, W + K steps of the reference's loop body, timed
    between barriers, max over ranks.  Returns a dict on every rank (kernel table and stats of rank 0's context)."""
    capi, dist, torch, ic = env["capi"], env["dist"], env["torch"], env["ic"]
    rank, world = env["rank"], env["world"]
    from summersph_amd.dist import DistSim, HipBackend, slab_bounds
    gas, sinks = ic.split_rows(rows)
    n_total = int(gas["x"].size)
    bounds = slab_bounds(gas["x"], world)
    sel = np.searchsorted(bounds, gas["x"], side="right") == rank
    mine = {k: v[sel] for k, v in gas.items()}
    mine["gid"] = np.nonzero(sel)[0]
    del gas
    plane, sim = env["data_backend"], None
    if halo_mode in ("auto", "native") and plane == "nccl" and native_applies(variable, flags, capi):
        sim = NativeSim.create(env, flags, mine, sinks, bounds, variable)
        if sim is None and rank == 0:
            print("[bench] native halo unavailable on some rank; using dist.py", file=sys.stderr, flush=True)
        if sim is not None:
            plane = "rccl(native)"
    if sim is None:
        be = HipBackend(env["local_rank"], variable=True, flags=flags | capi.FLAG_VARIABLE_H) if variable else HipBackend(env["local_rank"], flags=flags)
        # device tensors over RCCL (nccl); host-staged for the gloo rehearsal
        sim = DistSim(be, mine, sinks, bounds, group=env["data_group"], comm_device=None if plane == "nccl" else "cpu")
        sim.ctx = be.ctx
    ctx = sim.ctx

    def barrier():
        ctx.synchronize(); torch.cuda.synchronize(); dist.barrier()

    # clocks at their working level before the (short) timed region: ~100 ms of unrelated work on the same device
    heat = torch.ones(2048, 2048, device=f"cuda:{env['local_rank']}" if "local_rank" in env else "cuda")
    t_h = time.perf_counter()
    while time.perf_counter() - t_h < 0.1:
        heat = (heat @ heat) * (1.0 / 2048.0)
        torch.cuda.synchronize()
    del heat
    dt = sim.run(warmup, 1e-2)
    sim.profile = profile
    ctx.timing(True, only=list(dominant), stride=EVENT_STRIDE); ctx.timing_reset()        # the other groups: untimed inside the region
    barrier()
    t0 = time.perf_counter()
    dt = sim.run(steps, dt)
    barrier()
    elapsed = time.perf_counter() - t0
    ctx.timing(False)
    red = torch.tensor([elapsed], dtype=torch.float64)
    dist.all_reduce(red, op=dist.ReduceOp.MAX)
    per = torch.zeros(world, 3, dtype=torch.int64)
    per[rank, 0], per[rank, 1], per[rank, 2] = int(sim.n_owned), int(sim.stats["ghosts"]), 1
    dist.all_reduce(per)
    out = {"elapsed": float(red[0]), "dt": dt, "plane": plane, "n_total": n_total, "ranks_seen": int(per[:, 2].sum()),
           "owned_per_rank": per[:, 0].tolist(), "ghosts_per_rank": per[:, 1].tolist(), "steps": steps, "warmup": warmup,
           "halo_stats_rank0": dict(sim.stats)}
    if rank == 0:
        out["stats"] = ctx.stats()
        out["table"] = {k: ctx.timing_get(k) for k in capi.KERNELS}
        for k in dominant:                              # every EVENT_STRIDE-th launch was bracketed (see EVENT_STRIDE)
            out["table"][k] = (out["table"][k][0] * EVENT_STRIDE, out["table"][k][1] * EVENT_STRIDE)
        if profile:
            out["phase_ms_per_step"] = {k: 1e3 * v / steps for k, v in sim.phase_s.items()}
    if hasattr(sim, "close"):
        sim.close()
    else:
        ctx.close()
    return out


def strong_record(r, what):
    return {"workload": what, "n_total": r["n_total"], "value": r["n_total"] * r["steps"] / r["elapsed"], "unit": "particle-steps/s",
            "ms_per_step": r["elapsed"] / r["steps"] * 1e3, "steps": r["steps"], "warmup": r["warmup"], "scaling": "strong",
            "ranks": r["ranks_seen"], "plane": PLANES.get(r["plane"], r["plane"]), "owned_per_rank": r["owned_per_rank"],
            "ghosts_per_rank": r["ghosts_per_rank"], "final_dt": r["dt"], "halo_stats_rank0": r["halo_stats_rank0"]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--particles", dest="n", type=int, default=1_000_000, help="gas particles per GPU")
    ap.add_argument("--nngb", type=float, default=85.0, help="midplane neighbour target of the fixed-h IC (mean is ~0.6x)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-variable", action="store_true", help="skip the extra records at N = 1 (variable_h, full_simulate, side records)")
    ap.add_argument("--no-extras", action="store_true", help="skip ring4m_strong and full_1e7 (BASELINE configs[3] and [4])")
    ap.add_argument("--extras-scale", type=float, default=1.0, help="shrink the two strong-scaling records (rehearsals): "
                    "particle counts are multiplied by this")
    ap.add_argument("--reuse-density", action="store_true", help="SPH_FLAG_REUSE_DENSITY (NOT the headline mode)")
    ap.add_argument("--no-tiles", action="store_true", help="SPH_FLAG_NO_LDS_TILES: per-lane-gather list build (A/B)")
    ap.add_argument("--mode", default="fixed", choices=["fixed", "variable"],
                    help="headline workload: [F] fixed h (default) or [V] per-particle h")
    ap.add_argument("--ic", default="disc", choices=["disc", "ring"], help="fixed-h workload: the uniform disc (headline) or "
                    "BASELINE configs[3]'s thin ring r ~ N(r0, 0.05 r0) (artificial viscosity at work)")
    ap.add_argument("--self-gravity", action="store_true", help="SPH_FLAG_SELF_GRAVITY in the headline run (NOT the "
                    "default workload): Barnes-Hut gas self-gravity; with --gpus > 1 every rank builds the replicated tree")
    ap.add_argument("--full-simulate", action="store_true", help="the reference's whole loop body in the headline run (NOT the "
                    "default workload): --self-gravity plus sink accretion and the boundary cull, on any number of GPUs")
    ap.add_argument("--dist-profile", action="store_true", help="N>1: synchronise at phase boundaries and report wall "
                    "time per phase of the distributed step (perturbs the headline value)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo = rehearsal of the N>1 path on fewer GPUs than ranks (host-staged messages)")
    ap.add_argument("--halo", default="auto", choices=["auto", "python", "native"],
                    help="N>1 orchestrator: the native step loop of libsummersph_halo.so (own RCCL communicator, second HIP "
                    "stream) or summersph_amd/dist.py over torch.distributed.  auto: native where it applies and RCCL is "
                    "usable on every rank, else dist.py")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    import torch
    from summersph_amd import capi, ic

    dist = None
    data_group, data_backend = None, args.backend
    if world > 1:
        import torch.distributed as dist
        # control plane (barriers, the final max-reduction): gloo.  Data plane (halos, reductions of the run): RCCL over
        # xGMI, one rank per GPU; if the RCCL group cannot be set up or fails its probe on ANY rank (an exception, e.g. two
        # ranks given the same GPU), every rank falls back to host-staged gloo messages and the JSON line says so.
        local_rank = local_rank % max(torch.cuda.device_count(), 1)      # fewer GPUs than ranks: share (RCCL refuses, gloo runs)
        torch.cuda.set_device(local_rank)
        # gloo announces its connections on stdout ("[Gloo] Rank 0 is connected to ..."): keep stdout for the ONE JSON line
        sys.stdout.flush()
        saved_out = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group(backend="gloo")
            dist.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(saved_out, 1)
            os.close(saved_out)
        if args.backend == "nccl":
            ok = 1
            try:
                dev = torch.device("cuda", local_rank)
                data_group = dist.new_group(backend="nccl", device_id=dev)
                probe = torch.full((8,), float(rank), dtype=torch.float64, device=dev)
                dist.all_reduce(probe, group=data_group)
                got = torch.empty((world, 2), dtype=torch.int64, device=dev)
                dist.all_gather_into_tensor(got.view(-1), torch.tensor([rank, world], dtype=torch.int64, device=dev), group=data_group)
                nb = [q for q in (rank - 1, rank + 1) if 0 <= q < world]
                rbuf = [torch.empty(4, dtype=torch.float64, device=dev) for _ in nb]
                ops = [dist.P2POp(dist.isend, probe[:4].contiguous(), q, data_group) for q in nb] + \
                      [dist.P2POp(dist.irecv, b, q, data_group) for q, b in zip(nb, rbuf)]
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
                torch.cuda.synchronize()
                assert float(probe[0]) == world * (world - 1) / 2 and all(float(b[0]) == world * (world - 1) / 2 for b in rbuf)
            except Exception as e:       # noqa: BLE001 -- whatever RCCL raises, the run continues on gloo
                ok = 0
                print(f"[bench rank {rank}] RCCL data plane unavailable ({type(e).__name__}: {e}); falling back to gloo", file=sys.stderr, flush=True)
            flag = torch.tensor([ok], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag[0]) == 0:
                data_group, data_backend = None, "gloo"
    env = {"capi": capi, "ic": ic, "torch": torch, "dist": dist, "rank": rank, "world": world, "local_rank": local_rank,
           "data_group": data_group, "data_backend": data_backend}

    if args.full_simulate:
        args.self_gravity = True
    variable = args.mode == "variable"
    flags = (capi.FLAG_REUSE_DENSITY if args.reuse_density else 0) | (capi.FLAG_NO_LDS_TILES if args.no_tiles else 0) \
        | (capi.FLAG_SELF_GRAVITY if args.self_gravity else 0) \
        | (capi.FLAG_ACCRETE_CULL if args.full_simulate else 0) \
        | (capi.FLAG_SINK_CREATION if args.full_simulate and args.mode == "variable" else 0)

    def headline_rows(n):
        return ic.keplerian_disc_var(n, seed=303) if variable else (
            ic.thin_ring(n, seed=404) if args.ic == "ring" else ic.keplerian_disc(n, seed=202, nngb=args.nngb))

    # ---- headline workload -----------------------------------------------------------------------
    lim = load_limiters()
    head_dist = None
    if world == 1:
        gas, sinks = ic.split_rows(headline_rows(args.n))
        wl = Workload(capi, torch, local_rank, gas, sinks, flags, variable)
        del gas
        res = timed_run(wl, args.steps, args.warmup, repeats=3)
        wl.close()
        elapsed, dt, kt, st = res["elapsed"], res["dt"], res["table"], res["stats"]
        owned, ghosts, plane = [args.n], [0], None
    else:
        # weak scaling: the disc holds n x world particles (same surface density, larger radius); every rank
        # owns one equal-count x-slab of it
        head_dist = dist_run(env, headline_rows(args.n * world), variable, flags, args.steps, args.warmup, args.halo, profile=args.dist_profile)
        elapsed, dt, plane = head_dist["elapsed"], head_dist["dt"], head_dist["plane"]
        owned, ghosts = head_dist["owned_per_rank"], head_dist["ghosts_per_rank"]
        kt, st = head_dist.get("table"), head_dist.get("stats")

    out = None
    if rank == 0:
        f_ms, f_cnt = kt["forces"]
        if world > 1 and not args.self_gravity:
            f_cnt = max(f_cnt // 2, 1)        # N > 1: one force pass = two launches (interior + boundary wavefronts)
        f_avg_s = f_ms / max(f_cnt, 1) * 1e-3
        bytes_forces = (BYTES_VAR if variable else BYTES)["forces"]
        alg_bytes = bytes_forces * args.n
        achieved = alg_bytes / f_avg_s / 1e9
        value = args.n * world * args.steps / elapsed
        bps = BYTES_PER_STEP_VAR if variable else BYTES_PER_STEP
        pair_visits = 2 * 2 * st.nlist_mean * args.n * args.steps / elapsed
        flops = (FLOPS_DENSITY_PAIR + FLOPS_FORCE_PAIR) * 2 * st.nlist_mean * args.n * args.steps / elapsed / 1e12
        fk = ["forces_v_kernel"] if variable else (["forces_q", "forces_wt"] if st.tile_fit_pct_forces >= 90 else ["forces_kernel"])
        wkey = "variable" if variable else ("full" if args.full_simulate else "fixed")
        comparable = world == 1 and args.ic == "disc" and (args.full_simulate or not args.self_gravity)
        roof = roofline_record(lim if comparable else ({}, None, None), wkey, fk, alg_bytes, f_avg_s, f_cnt,
                               st.lane_efficiency_forces if not variable else (st.nlist_mean / st.nlist_wave_mean if st.nlist_wave_mean else None), args.n)
        wl_name = ((f"thin Keplerian ring (r ~ N(r0, 0.05 r0)), " if args.ic == "ring" and not variable else "uniform Keplerian disc, ")
                   + f"{args.n} gas particles + 1 sink per GPU, "
                   + ("variable h (BASELINE configs[2]: grad-h, leaf-box neighbour rule, h update every step), "
                      if variable else "fixed h=2.5 ([F] path, BASELINE configs[1] shape at the metric's N=1e6), ")
                   + f"mean {st.nlist_mean:.1f} list entries per particle in the timed steps, 2 density + 2 force passes per step"
                   + (", Barnes-Hut gas self-gravity (theta 0.5)" if args.self_gravity else "")
                   + (", sink accretion + boundary cull" if args.full_simulate else ""))
        out = {
            "metric": "particle-steps/sec", "value": value, "unit": "particle-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            # restarts of the same state (same upload, same warm-up, the same K steps): run-to-run spread
            "repeat_ms_per_step": res["repeat_ms"] if world == 1 else None,
            "preheat": ("one untimed pass of the same W + K steps, then the same upload again, before the W warm-up + K timed steps"
                        if world == 1 else "~100 ms of unrelated device work before the W warm-up steps")
                       + " (the timed region is ~25 ms: started on an idle GPU it falls into the clock ramp)",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": wl_name, "mode": args.mode, "n_particles_per_gpu": args.n, "mean_neighbours": st.nlist_mean, "mean_wave_trips": st.nlist_wave_mean,
                       "stats_read": "directly after the timed region",
                       "tile_fit_pct": st.tile_fit_pct, "tile_fit_pct_forces": st.tile_fit_pct_forces,
                       "max_neighbours": st.nlist_max, "grid": list(st.grid_dim), "reuse_density": bool(args.reuse_density),
                       "parallelism": "1 GPU" if world == 1 else f"{world} x-slabs, ghost exchange + migration over " + PLANES.get(plane, plane),
                       "plane": plane, "owned_per_rank": owned, "ghosts_per_rank": ghosts,
                       "max_owned_per_gpu": max(owned), "max_ghosts_per_gpu": max(ghosts), "rank0_slots": int(st.n)},
            "roofline": roof,
            "valu_fp64": {"achieved_tflops_est": flops, "peak_tflops": FP64_PEAK_TFLOPS, "frac": flops / FP64_PEAK_TFLOPS,
                          "pair_visits_per_s": pair_visits},
            "hbm_step": {"algorithmic_bytes_per_particle_step": bps, "achieved_GBs": bps * value / world / 1e9},
            "kernel_ms_per_step": {k: v[0] / args.steps for k, v in kt.items()},
            "kernel_ms_note": "the dominant group (forces) timed inside the timed region; the others from further steps of the same "
                              "trajectory with every group bracketed by HIP events (that costs ~4 % and stays out of `value`)",
            "final_dt": dt, "device_bytes": st.device_bytes, "list_builds": st.nlist_builds, "list_reflags": st.nlist_reflags,
        }
        copy_gbs = stream_copy_gbs(torch, local_rank)
        out["roofline"]["stream_copy_GBs"] = copy_gbs
        out["roofline"]["frac_of_stream_copy"] = achieved / copy_gbs
        out["roofline"]["note"] = ("compulsory HBM traffic is tiny for this path (hbm_step); what the pair kernels wait for is "
                                   "in `limiter`: fp64 vector issue, the LDS tile reads and idle lanes (lane_efficiency)")
        if head_dist is not None:
            out["halo_stats_rank0"] = head_dist["halo_stats_rank0"]
            if "phase_ms_per_step" in head_dist:
                out["dist_phase_ms_per_step_rank0"] = head_dist["phase_ms_per_step"]

    default_headline = not variable and args.ic == "disc" and not args.self_gravity and not args.reuse_density and not args.no_tiles

    # ---- BASELINE configs[3] and [4]: strong scaling, every N (the N = 1 points of the two curves come from here too) -----
    if default_headline and not args.no_extras:
        n_ring = max(20_000, int(4_000_000 * args.extras_scale))
        n_full = max(20_000, int(10_000_000 * args.extras_scale))
        full_flags = capi.FLAG_SELF_GRAVITY | capi.FLAG_ACCRETE_CULL
        fsteps = max(2, args.steps // 2)
        if world == 1:
            gas, sinks = ic.split_rows(ic.thin_ring(n_ring, seed=404))
            w = Workload(capi, torch, local_rank, gas, sinks, 0, False); del gas
            r = timed_run(w, args.steps, args.warmup, breakdown=False); w.close()
            ring = {"elapsed": r["elapsed"], "dt": r["dt"], "plane": "1 GPU (sph_run)", "n_total": n_ring, "ranks_seen": 1, "owned_per_rank": [n_ring],
                    "ghosts_per_rank": [0], "steps": args.steps, "warmup": args.warmup, "halo_stats_rank0": None}
            gas, sinks = ic.split_rows(ic.keplerian_disc(n_full, seed=505, nngb=args.nngb))
            w = Workload(capi, torch, local_rank, gas, sinks, full_flags, False); del gas
            r = timed_run(w, fsteps, 1, dominant=("grav_walk",), breakdown=False); w.close()
            full = {"elapsed": r["elapsed"], "dt": r["dt"], "plane": "1 GPU (sph_run)", "n_total": n_full, "ranks_seen": 1, "owned_per_rank": [r["n_left"]],
                    "ghosts_per_rank": [0], "steps": fsteps, "warmup": 1, "halo_stats_rank0": None}
        else:
            ring = dist_run(env, ic.thin_ring(n_ring, seed=404), False, 0, args.steps, args.warmup, args.halo)
            full = dist_run(env, ic.keplerian_disc(n_full, seed=505, nngb=args.nngb), False, full_flags, fsteps, 1, args.halo, dominant=("grav_walk",))
        if rank == 0:
            out["ring4m_strong"] = strong_record(ring, f"BASELINE configs[3]: thin Keplerian ring r ~ N(r0, 0.05 r0), {n_ring} gas particles + 1 sink in "
                                                       f"total, artificial viscosity on, fixed h = 2.5, split into {world} x-slab(s)")
            out["full_1e7"] = strong_record(full, f"BASELINE configs[4]: uniform Keplerian disc, {n_full} gas particles + central sink in total, Barnes-Hut gas "
                                                  f"self-gravity (theta 0.5), sink accretion + boundary cull: simulate() as the reference runs it, split into {world} x-slab(s)")

    if rank == 0:
        if world == 1 and default_headline and not args.no_variable:
            # BASELINE configs[2] on the same GPU, same step count.  Both list groups are bracketed: `nlist` = builds
            # (nlist_v_tiled) only, `reflag` = the re-flag pass (nlist_v_reflag) -- the roofline is the build's
            gas, sinks = ic.split_rows(ic.keplerian_disc_var(args.n, seed=303))
            vw = Workload(capi, torch, local_rank, gas, sinks, 0, True); del gas
            vr = timed_run(vw, args.steps, args.warmup, dominant=("nlist", "reflag"), late_window=(25, 45))
            vw.close()
            vel, vkt, vst = vr["elapsed"], vr["table"], vr["stats"]
            out["variable_h"] = {
                "workload": f"BASELINE configs[2]: uniform Keplerian disc, {args.n} particles, variable h "
                            f"(h 2.5..8, eta 1.2), grad-h, leaf-box neighbour rule, h update every step",
                "value": args.n * args.steps / vel, "unit": "particle-steps/s", "ms_per_step": vel / args.steps * 1e3,
                "mean_list_entries": vst.nlist_mean, "mean_wave_trips": vst.nlist_wave_mean, "max_list_entries": vst.nlist_max, "grid": list(vst.grid_dim),
                "list_builds": vst.nlist_builds, "list_reflags": vst.nlist_reflags,
                "kernel_ms_per_step": {k: v[0] / args.steps for k, v in vkt.items()}, "final_dt": vr["dt"],
                "late_window": vr["late"],       # the same trajectory later on, where the h update iterates for most particles
                "target_BASELINE_md": 1.0e7,
                # dominant kernel: the list build (nlist_v_tiled): reads {x,y,z,h} + leaf box + id (68 B), writes the entries
                # (4 B each) and two counts (8 B) per particle.  Duration = the mean over list BUILDS alone.
                "roofline": roofline_record(lim, "variable", ["nlist_v_tiled"], (68 + 8 + 4 * vst.nlist_mean) * args.n,
                                            vkt["nlist"][0] / max(vkt["nlist"][1], 1) * 1e-3, vkt["nlist"][1],
                                            vst.nlist_mean / vst.nlist_wave_mean if vst.nlist_wave_mean else None, args.n),
                "reflag_avg_launch_ms": vkt["reflag"][0] / max(vkt["reflag"][1], 1), "reflag_launches": vkt["reflag"][1],
                "hbm_step": {"algorithmic_bytes_per_particle_step": BYTES_PER_STEP_VAR,
                             "achieved_GBs": BYTES_PER_STEP_VAR * args.n * args.steps / vel / 1e9}}
            # simulate() as the reference runs it: + Barnes-Hut gas self-gravity, accretion, boundary cull
            gas, sinks = ic.split_rows(headline_rows(args.n))
            fw = Workload(capi, torch, local_rank, gas, sinks, capi.FLAG_SELF_GRAVITY | capi.FLAG_ACCRETE_CULL, False)
            fsteps = max(2, args.steps // 2)
            fr = timed_run(fw, fsteps, 1, dominant=("grav_walk",))
            fw.close()
            fel, fkt = fr["elapsed"], fr["table"]
            out["full_simulate"] = {
                "workload": f"the headline disc with find_forces as the reference has it (Barnes-Hut gas self-gravity, "
                            f"theta 0.5) and the end-of-step sink accretion + boundary cull",
                "value": args.n * fsteps / fel, "unit": "particle-steps/s", "ms_per_step": fel / fsteps * 1e3, "steps": fsteps,
                "kernel_ms_per_step": {k: v[0] / fsteps for k, v in fkt.items()}, "particles_left": fr["n_left"], "final_dt": fr["dt"],
                # dominant kernel: the tree walk (grav_walk_wave): reads {x,y,z,m}, leaf and id (40 B), writes a (24 B) per particle
                "roofline": roofline_record(lim, "full", ["grav_walk"], 64 * args.n,
                                            fkt["grav_walk"][0] / max(fkt["grav_walk"][1], 1) * 1e-3, fkt["grav_walk"][1], None, args.n)}
            # the same loop with SPH_FLAG_REUSE_GRAVITY: the start-of-step evaluation copies the Barnes-Hut term of the
            # previous step's last walk (bitwise the same accelerations) -- reported beside the as-the-reference-runs number
            gw = Workload(capi, torch, local_rank, gas, sinks, capi.FLAG_SELF_GRAVITY | capi.FLAG_ACCRETE_CULL | capi.FLAG_REUSE_GRAVITY, False)
            gr = timed_run(gw, fsteps, 1, dominant=("gravity",), breakdown=False)
            gw.close()
            out["full_simulate"]["reuse_gravity"] = {"value": args.n * fsteps / gr["elapsed"], "unit": "particle-steps/s", "ms_per_step": gr["elapsed"] / fsteps * 1e3,
                                                      "gravity_ms_per_step": gr["table"]["gravity"][0] / fsteps, "final_dt": gr["dt"],
                                                      "note": "SPH_FLAG_REUSE_GRAVITY: one tree walk per step instead of two, same results bit for bit"}
            # two side records that show the kernel selection away from the friendly geometry: a dense disc (the survey's
            # anchor: ~200 neighbours) and a thick 3-D box (nothing fits a tile: the direct-gather kernels of pairs.hip)
            out["side_records"] = {}
            for tag, n_s, rows_fn in (("dense_disc_200k", 200_000, lambda: ic.keplerian_disc(200_000, seed=212, nngb=340.0)),
                                      ("thick_box_300k", 300_000, lambda: ic.uniform_box(300_000, seed=213)),
                                      ("reference_size_12k", 12_000, lambda: ic.keplerian_disc(12_000, seed=214, nngb=args.nngb))):
                g_s, s_s = ic.split_rows(rows_fn())
                sw = Workload(capi, torch, local_rank, g_s, s_s, 0, False)
                ss = max(2, args.steps // 2) if n_s > 50_000 else max(20, 4 * args.steps)
                sr = timed_run(sw, ss, 1)
                sw.close()
                sst = sr["stats"]
                out["side_records"][tag] = {"value": n_s * ss / sr["elapsed"], "unit": "particle-steps/s", "ms_per_step": sr["elapsed"] / ss * 1e3, "n": n_s,
                                            "mean_neighbours": sst.nlist_mean, "mean_wave_trips": sst.nlist_wave_mean,
                                            "tile_fit_pct": sst.tile_fit_pct, "tile_fit_pct_forces": sst.tile_fit_pct_forces,
                                            "lane_efficiency_forces": sst.lane_efficiency_forces,
                                            "kernel_ms_per_step": {k: v[0] / ss for k, v in sr["table"].items() if v[0] > 0}}
            # the start-of-step density pass recomputes a bitwise identical rho (positions, masses, h unchanged
            # since the end of the last step): SPH_FLAG_REUSE_DENSITY keeps it.  Reported beside the headline,
            # which runs every pass the reference runs.
            rw = Workload(capi, torch, local_rank, gas, sinks, flags | capi.FLAG_REUSE_DENSITY, False)
            rr = timed_run(rw, args.steps, args.warmup, breakdown=False)
            rw.close()
            out["fixed_reuse_density"] = {"value": args.n * args.steps / rr["elapsed"], "unit": "particle-steps/s",
                                          "ms_per_step": rr["elapsed"] / args.steps * 1e3, "final_dt": rr["dt"],
                                          "note": "same results as the headline run, 1 density + 2 force passes per step"}
        if not args.no_cpu and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.n, args.nngb)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
