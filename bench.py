#!/usr/bin/env python3
"""bench.py -- particle-steps/s of the SPH hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 10 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one iteration of the reference's simulate() loop body (SUMMER_SPH.f90:889-916):
2 density passes + 2 force passes + 2 half-kicks + 1 drift + the dt reduction (variable-h: + the
h update, Variable.f90:1152), on a seeded synthetic Keplerian disc (summersph_amd/ic.py).  Inputs
are resident in HBM before the timed region starts.  Rank 0 prints ONE JSON line.

Workloads
  fixed (default, every N)   uniform disc, N x 1e6 gas particles + 1 sink, fixed h = 2.5: the [F] path
                             (BASELINE configs[1] shape at the metric's N = 1e6); shards over GPUs as
                             x-slabs with ghost exchange + migration over RCCL: the native step loop of
                             libsummersph_halo.so where it applies (--halo auto), else summersph_amd/dist.py.
  variable (N = 1)           BASELINE configs[2]: 1e6 particles, per-particle h, grad-h terms, the
                             reference's leaf-box neighbour rule, h update every step.  At N = 1 the
                             default run measures it too and reports it as "variable_h" next to the
                             headline (the >= 1e7 target of BASELINE.md is quoted on this config).

Extra objects in the JSON line
  roofline     dominant kernel of the headline workload (the forces kernel): ALGORITHMIC HBM bytes per launch (SURVEY.md
               8(d): 80 B read + 40 B written per particle) / its mean launch duration, measured with HIP events on the
               library's own stream during the timed steps; peak 8 TB/s.  `traffic` (PMC bytes per launch) and `limiter`
               (TA / vector-issue / LDS occupancy of that kernel) are quoted from profiles/r02_limiters.json -- the
               rocprofv3 passes of profiles/r02_profile.sh on this code -- when they were taken on this workload size;
               `limiter.lane_efficiency` is computed live.  `bound` names the resource that is busiest: the path's
               compulsory HBM traffic is tiny, so it is never "hbm".  The `variable_h` and `full_simulate` records carry
               their own roofline object (dominant kernels: the variable-h list build, the gravity tree walk).
  cpu_baseline the CPU oracle (oracle/sph_oracle.c, OpenMP over the host cores) timed on a bounded
               sample of the same workload on this box (rank 0, N = 1 only).
Timing: inside the timed region only the dominant kernel group is bracketed by HIP events (the roofline's duration);
`kernel_ms_per_step` of the other groups comes from further steps of the same trajectory with every group bracketed, and
`repeat_ms_per_step` repeats the K steps three more times (run-to-run spread) -- neither enters `value`.
"""
import argparse
import json
import os
import sys
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


def load_limiters():
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "r02_limiters.json")))
    except (OSError, ValueError):
        return {}


def roofline_record(limiters, workload, prefixes, alg_bytes, avg_launch_s, launches, lane_eff, n, default_n=1_000_000):
    """roofline object of one kernel: algorithmic bytes / measured duration against the HBM peak, plus what the committed
    counter passes say limits it (only when they were taken on this workload size)"""
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
        limiter["source"] = f"profiles/r02_limiters.json [{workload}][{name}] (profiles/r02_profile.sh, rocprofv3 counter passes)"
        limiter["avg_launch_us_rocprof"] = rec.get("avg_us")
    return {"bound": bound, "kernel": name or prefixes[0], "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": rec.get("traffic_bytes_per_launch") if rec else None,
            "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": avg_launch_s * 1e3, "launches": launches, "limiter": limiter}


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


def cpu_baseline(n_total, nngb, seconds_target=15.0):
    """CPU oracle on the host cores of this box, bounded sample (about seconds_target of CPU work)."""
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
    # one thread, on the calibration-size disc (the unmodified reference is serial: SURVEY.md 5)
    rows = ic.keplerian_disc(20000, seed=1, nngb=nngb)
    gas, sinks = ic.split_rows(rows)
    o1 = orc.Oracle(gas, sinks, nthreads=1)
    s0 = time.perf_counter(); o1.step(1e-2); s1 = time.perf_counter()
    return {"value": k * n_s / (t1 - t0), "unit": "particle-steps/s", "cores": threads, "kind": "port",
            "sample": f"{k} full step(s) (2 density + 2 force passes, kick/drift/dt each) of a {n_s}-particle disc of "
                      f"the same surface density, OpenMP x{threads}, {t1 - t0:.1f} s",
            "single_thread_value": 20000 / (s1 - s0),
            "single_thread_sample": f"1 full step of a 20000-particle disc of the same surface density, {s1 - s0:.1f} s"}


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


def make_single_ctx(capi, ic, torch, variable, n, nngb, device, flags, ring=False):
    """one context on `device` with the workload uploaded from device memory"""
    rows = ic.keplerian_disc_var(n, seed=303) if variable else (
        ic.thin_ring(n, seed=404) if ring else ic.keplerian_disc(n, seed=202, nngb=nngb))
    gas, sinks = ic.split_rows(rows)
    ctx = capi.Context(device=device, variable=True, flags=flags | capi.FLAG_VARIABLE_H) if variable else capi.Context(device=device, flags=flags)
    dev = [torch.from_numpy(gas[k]).to(f"cuda:{device}") for k in "x y z vx vy vz u m alpha".split()]
    torch.cuda.synchronize()
    ctx.upload_dev(n, [t.data_ptr() for t in dev])               # inputs resident in HBM
    if variable:
        hdev = torch.from_numpy(gas["h"]).to(f"cuda:{device}")
        torch.cuda.synchronize()
        ctx.upload_field_dev("h", hdev.data_ptr(), n)
    ctx.set_sinks(sinks)
    return ctx


class NativeSim:
    """bench face of the native multi-GPU step loop (summersph_amd/halo.py): what DistSim offers the timed region"""

    @classmethod
    def create(cls, capi, dist, torch, local_rank, rank, world, flags, mine, sinks, bounds):
        from summersph_amd import halo
        uid = [halo.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        self = cls()

        def agreed(ok):          # every rank must have succeeded: the next call is collective
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            return int(flag[0]) == 1

        ok = True
        try:
            self.ctx = capi.Context(device=local_rank, flags=flags)
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
        return {"ghosts": s.ghosts, "migrated": s.migrated, "exchanges": s.exchanges, "migrations": s.migrations}


def timed_run(ctx, torch, steps, warmup, dominant=("forces",), breakdown=True, repeats=0):
    """-> (seconds of the timed region, final dt, kernel table).  Inside the timed region only the `dominant` kernel
    groups are bracketed by HIP events (the roofline's duration is measured live there; bracketing every group costs
    ~4 % of a fixed-h step).  The per-group table comes from up to 10 further steps of the same trajectory with every
    group bracketed -- outside the timed region -- in the format of Context.timing_get scaled to `steps` steps; the
    dominant groups keep their totals from the timed region."""
    dt, t = ctx.run(warmup, 1e-2, 0.0)
    ctx.timing(True, only=list(dominant)); ctx.timing_reset()
    ctx.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    dt, t = ctx.run(steps, dt, t)
    ctx.synchronize(); torch.cuda.synchronize()
    el = time.perf_counter() - t0
    ctx.timing(False)
    from summersph_amd import capi
    table = {k: ctx.timing_get(k) for k in capi.KERNELS}
    if repeats:
        # run-to-run spread: the same number of steps again, `repeats` times, on the continuing trajectory (not part of `value`)
        spread = []
        for _ in range(repeats):
            ctx.synchronize()
            r0 = time.perf_counter()
            dt_r, t = ctx.run(steps, dt if not spread else dt_r, t)
            ctx.synchronize()
            spread.append((time.perf_counter() - r0) / steps * 1e3)
        table["_repeat_ms_per_step"] = spread
        dt_b = dt_r
    else:
        dt_b = dt
    if breakdown:
        bs = max(1, min(steps, 10))
        ctx.timing_reset(); ctx.timing(True)
        ctx.run(bs, dt_b, t)
        ctx.synchronize()
        ctx.timing(False)
        for k in capi.KERNELS:
            if k not in dominant:
                ms, cnt = ctx.timing_get(k)
                table[k] = (ms * steps / bs, int(round(cnt * steps / bs)))
    return el, dt, table


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--particles", dest="n", type=int, default=1_000_000, help="gas particles per GPU")
    ap.add_argument("--nngb", type=float, default=85.0, help="midplane neighbour target of the fixed-h IC (mean is ~0.6x)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-variable", action="store_true", help="skip the extra variable-h measurement at N = 1")
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
                    "stream; fixed h without self-gravity) or summersph_amd/dist.py over torch.distributed.  auto: native "
                    "where it applies and RCCL is usable on every rank, else dist.py")
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
        dist.init_process_group(backend="gloo")
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
    red_dev = "cpu" if world > 1 else f"cuda:{local_rank}"

    if args.full_simulate:
        args.self_gravity = True
    variable = args.mode == "variable"
    flags = (capi.FLAG_REUSE_DENSITY if args.reuse_density else 0) | (capi.FLAG_NO_LDS_TILES if args.no_tiles else 0) \
        | (capi.FLAG_SELF_GRAVITY if args.self_gravity else 0) \
        | (capi.FLAG_ACCRETE_CULL if args.full_simulate else 0) \
        | (capi.FLAG_SINK_CREATION if args.full_simulate and args.mode == "variable" else 0)

    # ---- headline workload -----------------------------------------------------------------------
    sim = None
    if world == 1:
        ctx = make_single_ctx(capi, ic, torch, variable, args.n, args.nngb, local_rank, flags, ring=args.ic == "ring")
        elapsed, dt, kt = timed_run(ctx, torch, args.steps, args.warmup, repeats=3)
        repeat_ms = kt.pop("_repeat_ms_per_step", None)
        n_max = [args.n, 0]
    else:
        # weak scaling: the disc holds n x world particles (same surface density, larger radius); every rank
        # owns one equal-count x-slab of it
        from summersph_amd.dist import DistSim, HipBackend, slab_bounds
        rows = ic.keplerian_disc_var(args.n * world, seed=303) if variable else (
            ic.thin_ring(args.n * world, seed=404) if args.ic == "ring"
            else ic.keplerian_disc(args.n * world, seed=202, nngb=args.nngb))
        gas, sinks = ic.split_rows(rows)
        bounds = slab_bounds(gas["x"], world)
        sel = np.searchsorted(bounds, gas["x"], side="right") == rank
        mine = {k: v[sel] for k, v in gas.items()}
        mine["gid"] = np.nonzero(sel)[0]
        del rows, gas
        native = None
        if args.halo in ("auto", "native") and data_backend == "nccl" and not variable and not args.self_gravity:
            native = NativeSim.create(capi, dist, torch, local_rank, rank, world, flags, mine, sinks, bounds)
            if native is None and rank == 0:
                print("[bench] native halo unavailable on some rank; using dist.py", file=sys.stderr, flush=True)
        if native is not None:
            sim, ctx, data_backend = native, native.ctx, "rccl(native)"
        else:
            be = HipBackend(local_rank, variable=True, flags=flags | capi.FLAG_VARIABLE_H) if variable else HipBackend(local_rank, flags=flags)
            # device tensors over RCCL (nccl); host-staged for the gloo rehearsal
            sim = DistSim(be, mine, sinks, bounds, group=data_group, comm_device=None if data_backend == "nccl" else "cpu")
            ctx = be.ctx

        def barrier():
            ctx.synchronize(); torch.cuda.synchronize(); dist.barrier()

        dt = sim.run(args.warmup, 1e-2)
        sim.profile = args.dist_profile
        ctx.timing(True, only=["forces"]); ctx.timing_reset()        # the other groups: untimed inside the region (see timed_run)
        barrier()
        t0 = time.perf_counter()
        dt = sim.run(args.steps, dt)
        barrier()
        elapsed = time.perf_counter() - t0
        ctx.timing(False)
        red = torch.tensor([elapsed, float(sim.n_owned), float(sim.stats["ghosts"])], dtype=torch.float64, device=red_dev)
        dist.all_reduce(red, op=dist.ReduceOp.MAX)
        elapsed, n_max = float(red[0]), [int(red[1]), int(red[2])]

    if rank == 0:
        st = ctx.stats()
        if sim is not None:
            kt = {k: ctx.timing_get(k) for k in capi.KERNELS}
        f_ms, f_cnt = kt["forces"]
        if sim is not None and not args.self_gravity:
            f_cnt = max(f_cnt // 2, 1)        # N > 1: one force pass = two launches (interior + boundary wavefronts)
        f_avg_s = f_ms / max(f_cnt, 1) * 1e-3
        bytes_forces = (BYTES_VAR if variable else BYTES)["forces"]
        alg_bytes = bytes_forces * args.n
        achieved = alg_bytes / f_avg_s / 1e9
        value = args.n * world * args.steps / elapsed
        bps = BYTES_PER_STEP_VAR if variable else BYTES_PER_STEP
        pair_visits = 2 * 2 * st.nlist_mean * args.n * args.steps / elapsed
        flops = (FLOPS_DENSITY_PAIR + FLOPS_FORCE_PAIR) * 2 * st.nlist_mean * args.n * args.steps / elapsed / 1e12
        limiters = load_limiters()
        fk = ["forces_v_kernel"] if variable else (["forces_q", "forces_wt"] if st.tile_fit_pct_forces >= 90 else ["forces_kernel"])
        wkey = "variable" if variable else ("full" if args.full_simulate else "fixed")
        comparable = world == 1 and args.ic == "disc" and (args.full_simulate or not args.self_gravity)
        roof = roofline_record(limiters if comparable else {}, wkey, fk, alg_bytes, f_avg_s, f_cnt,
                               st.lane_efficiency_forces if not variable else (st.nlist_mean / st.nlist_wave_mean if st.nlist_wave_mean else None), args.n)
        wl = ((f"thin Keplerian ring (r ~ N(r0, 0.05 r0)), " if args.ic == "ring" and not variable else "uniform Keplerian disc, ")
              + f"{args.n} gas particles + 1 sink per GPU, "
              + ("variable h (BASELINE configs[2]: grad-h, leaf-box neighbour rule, h update every step), "
                 if variable else "fixed h=2.5 ([F] path, BASELINE configs[1] shape at the metric's N=1e6), ")
              + f"mean {st.nlist_mean:.1f} list entries per particle, 2 density + 2 force passes per step"
              + (", Barnes-Hut gas self-gravity (theta 0.5)" if args.self_gravity else "")
              + (", sink accretion + boundary cull" if args.full_simulate else ""))
        out = {
            "metric": "particle-steps/sec", "value": value, "unit": "particle-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "repeat_ms_per_step": repeat_ms if world == 1 else None,      # the same K steps three more times: run-to-run spread
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": wl, "mode": args.mode, "n_particles_per_gpu": args.n, "mean_neighbours": st.nlist_mean, "mean_wave_trips": st.nlist_wave_mean,
                       "tile_fit_pct": st.tile_fit_pct, "tile_fit_pct_forces": st.tile_fit_pct_forces,
                       "max_neighbours": st.nlist_max, "grid": list(st.grid_dim), "reuse_density": bool(args.reuse_density),
                       "parallelism": "1 GPU" if world == 1 else
                                      f"{world} x-slabs, ghost exchange + migration over "
                                      + ({"nccl": "RCCL (summersph_amd/dist.py over torch.distributed nccl)",
                                          "rccl(native)": "RCCL, native step loop (libsummersph_halo.so: grouped send/recv on a second HIP stream)"}
                                         .get(data_backend, "host-staged gloo messages (summersph_amd/dist.py)")),
                       "max_owned_per_gpu": n_max[0], "max_ghosts_per_gpu": n_max[1], "rank0_slots": int(st.n)},
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
        if sim is not None and sim.profile:
            out["dist_phase_ms_per_step_rank0"] = {k: 1e3 * v / args.steps for k, v in sim.phase_s.items()}
            out["dist_stats_rank0"] = dict(sim.stats)
        ctx.close()
        if world == 1 and not variable and not args.no_variable and args.ic == "disc" and not args.self_gravity:
            # BASELINE configs[2] on the same GPU, same step count
            vctx = make_single_ctx(capi, ic, torch, True, args.n, args.nngb, local_rank, 0)
            vel, vdt, vkt = timed_run(vctx, torch, args.steps, args.warmup, dominant=("nlist",))
            vst = vctx.stats()
            out["variable_h"] = {
                "workload": f"BASELINE configs[2]: uniform Keplerian disc, {args.n} particles, variable h "
                            f"(h 2.5..8, eta 1.2), grad-h, leaf-box neighbour rule, h update every step",
                "value": args.n * args.steps / vel, "unit": "particle-steps/s", "ms_per_step": vel / args.steps * 1e3,
                "mean_list_entries": vst.nlist_mean, "mean_wave_trips": vst.nlist_wave_mean, "max_list_entries": vst.nlist_max, "grid": list(vst.grid_dim),
                "list_builds": vst.nlist_builds, "list_reflags": vst.nlist_reflags,
                "kernel_ms_per_step": {k: v[0] / args.steps for k, v in vkt.items()}, "final_dt": vdt,
                "target_BASELINE_md": 1.0e7,
                # dominant kernel: the list build (nlist_v_tiled): reads {x,y,z,h} + leaf box + id (68 B), writes the entries
                # (4 B each) and two counts (8 B) per particle
                "roofline": roofline_record(limiters, "variable", ["nlist_v_tiled"], (68 + 8 + 4 * vst.nlist_mean) * args.n,
                                            vkt["nlist"][0] / max(vkt["nlist"][1], 1) * 1e-3, vkt["nlist"][1],
                                            vst.nlist_mean / vst.nlist_wave_mean if vst.nlist_wave_mean else None, args.n),
                "hbm_step": {"algorithmic_bytes_per_particle_step": BYTES_PER_STEP_VAR,
                             "achieved_GBs": BYTES_PER_STEP_VAR * args.n * args.steps / vel / 1e9}}
            vctx.close()
            # simulate() as the reference runs it: + Barnes-Hut gas self-gravity, accretion, boundary cull
            fctx = make_single_ctx(capi, ic, torch, False, args.n, args.nngb, local_rank,
                                   capi.FLAG_SELF_GRAVITY | capi.FLAG_ACCRETE_CULL)
            fsteps = max(2, args.steps // 2)
            fel, fdt, fkt = timed_run(fctx, torch, fsteps, 1, dominant=("grav_walk",))
            out["full_simulate"] = {
                "workload": f"the headline disc with find_forces as the reference has it (Barnes-Hut gas self-gravity, "
                            f"theta 0.5) and the end-of-step sink accretion + boundary cull",
                "value": args.n * fsteps / fel, "unit": "particle-steps/s", "ms_per_step": fel / fsteps * 1e3, "steps": fsteps,
                "kernel_ms_per_step": {k: v[0] / fsteps for k, v in fkt.items()}, "particles_left": fctx.n, "final_dt": fdt,
                # dominant kernel: the tree walk (grav_walk_wave): reads {x,y,z,m}, leaf and id (40 B), writes a (24 B) per particle
                "roofline": roofline_record(limiters, "full", ["grav_walk_wave"], 64 * args.n,
                                            fkt["grav_walk"][0] / max(fkt["grav_walk"][1], 1) * 1e-3, fkt["grav_walk"][1], None, args.n)}
            fctx.close()
            # the same loop with SPH_FLAG_REUSE_GRAVITY: the start-of-step evaluation copies the Barnes-Hut term of the
            # previous step's last walk (bitwise the same accelerations) -- reported beside the as-the-reference-runs number
            gctx = make_single_ctx(capi, ic, torch, False, args.n, args.nngb, local_rank,
                                   capi.FLAG_SELF_GRAVITY | capi.FLAG_ACCRETE_CULL | capi.FLAG_REUSE_GRAVITY)
            gel, gdt, gkt = timed_run(gctx, torch, fsteps, 1, dominant=("gravity",), breakdown=False)
            out["full_simulate"]["reuse_gravity"] = {"value": args.n * fsteps / gel, "unit": "particle-steps/s", "ms_per_step": gel / fsteps * 1e3,
                                                      "gravity_ms_per_step": gkt["gravity"][0] / fsteps, "final_dt": gdt,
                                                      "note": "SPH_FLAG_REUSE_GRAVITY: one tree walk per step instead of two, same results bit for bit"}
            gctx.close()
            # two side records that show the kernel selection away from the friendly geometry: a dense disc (the survey's
            # anchor: ~200 neighbours) and a thick 3-D box (nothing fits a tile: the direct-gather kernels of pairs.hip)
            out["side_records"] = {}
            for tag, n_s, rows_fn in (("dense_disc_200k", 200_000, lambda: ic.keplerian_disc(200_000, seed=212, nngb=340.0)),
                                      ("thick_box_300k", 300_000, lambda: ic.uniform_box(300_000, seed=213))):
                g_s, s_s = ic.split_rows(rows_fn())
                sctx = capi.Context(device=local_rank)
                sctx.upload(g_s); sctx.set_sinks(s_s)
                sel, sdt, skt = timed_run(sctx, torch, max(2, args.steps // 2), 1)
                sst = sctx.stats()
                ss = max(2, args.steps // 2)
                out["side_records"][tag] = {"value": n_s * ss / sel, "unit": "particle-steps/s", "ms_per_step": sel / ss * 1e3, "n": n_s,
                                            "mean_neighbours": sst.nlist_mean, "mean_wave_trips": sst.nlist_wave_mean,
                                            "tile_fit_pct": sst.tile_fit_pct, "tile_fit_pct_forces": sst.tile_fit_pct_forces,
                                            "lane_efficiency_forces": sst.lane_efficiency_forces,
                                            "kernel_ms_per_step": {k: v[0] / ss for k, v in skt.items() if v[0] > 0}}
                sctx.close()
            if not args.reuse_density:
                # the start-of-step density pass recomputes a bitwise identical rho (positions, masses, h unchanged
                # since the end of the last step): SPH_FLAG_REUSE_DENSITY keeps it.  Reported beside the headline,
                # which runs every pass the reference runs.
                rctx = make_single_ctx(capi, ic, torch, False, args.n, args.nngb, local_rank, flags | capi.FLAG_REUSE_DENSITY)
                rel, rdt, _ = timed_run(rctx, torch, args.steps, args.warmup, breakdown=False)
                out["fixed_reuse_density"] = {"value": args.n * args.steps / rel, "unit": "particle-steps/s",
                                              "ms_per_step": rel / args.steps * 1e3, "final_dt": rdt,
                                              "note": "same results as the headline run, 1 density + 2 force passes per step"}
                rctx.close()
        if not args.no_cpu and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.n, args.nngb)
        print(json.dumps(out), flush=True)
    else:
        ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
