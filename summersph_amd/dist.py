"""One-process-per-GPU domain decomposition of the SPH hot path.

Interactions are short range (2h), so the path shards by space: every rank owns the particles of one
slab along x (slab edges = particle-count quantiles) and keeps GHOST copies of the other ranks'
particles that lie within 2h of its own particles' bounding box.  Per force evaluation:

  positions changed (after a drift)                    positions unchanged (start of the next step)
  1. migrate particles that left their slab            1. refresh ghost v, u, alpha (they were kicked
  2. all-gather the ranks' bounding boxes                 by their owners)
  3. send owned particles inside bbox_q (+2h) to q     2. density of owned particles again (as the
  4. upload owned + ghosts, density of owned              reference does), EOS of everything
  5. send rho of the same particles -> ghost rho       3. forces
  6. EOS of ghosts, forces of owned
  then: sink accelerations summed over ranks (all-reduce), and at the end of a step the dt
  candidate is min-reduced and the reference's dt rule ([F]:855-858) is applied on every rank.

torch.distributed carries every exchange: backend "nccl" (= RCCL over xGMI; every pair of GPUs has
a direct link, so the point-to-point halo messages of different pairs do not share links) on a GPU
node, "gloo" for the CPU tests.  The arithmetic is done by a *backend object*: `HipBackend` (the
C ABI, device memory) in production; tests plug in an oracle-based backend to exercise this
orchestration on CPUs.  Nothing in this module computes physics.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

STATE = ["x", "y", "z", "vx", "vy", "vz", "u", "m", "alpha"]


class HipBackend:
    """The C-ABI context as seen by the orchestrator.  Tensors are float64 on `device`.

    The library runs on its own HIP stream, torch on its current stream: every hand-over of a torch
    buffer to the library is preceded by torch.cuda.synchronize (torch's caching allocator may recycle
    memory that queued torch kernels still read), and every library call that fills a buffer
    synchronises its stream before returning."""

    def __init__(self, device_index: int = 0, **param_overrides):
        from . import capi
        self.capi = capi
        self.ctx = capi.Context(device=device_index, **param_overrides)
        self.device = torch.device("cuda", device_index)
        self.n = 0
        self.n_owned = 0

    @property
    def params(self):
        return self.ctx.params

    def set_rank(self, rank, nranks):
        self.ctx.set_rank(rank, nranks)

    def upload(self, state: torch.Tensor, n_owned: int):
        """state: [9, n] (rows in STATE order), owned particles first, then ghosts"""
        state = state.contiguous()
        self.n = int(state.shape[1])
        self.n_owned = int(n_owned)
        torch.cuda.synchronize(self.device)
        self.ctx.upload_dev(self.n, [state[k].data_ptr() for k in range(9)])
        self.ctx.set_owned(self.n_owned)

    def gather(self, names, ids: torch.Tensor | None = None, count: int | None = None) -> torch.Tensor:
        """[len(names), count] values of the particles with original ids `ids` (None: 0..count-1)"""
        count = int(ids.numel()) if ids is not None else int(count)
        out = torch.empty((len(names), count), dtype=torch.float64, device=self.device)
        if count:
            torch.cuda.synchronize(self.device)
            self.ctx.gather_fields_dev(names, count, ids.data_ptr() if ids is not None else 0, out.data_ptr())
        return out

    def scatter(self, names, first: int, vals: torch.Tensor):
        vals = vals.contiguous()
        if vals.numel():
            torch.cuda.synchronize(self.device)
            self.ctx.scatter_fields_dev(names, first, vals.shape[1], vals.data_ptr())

    def set_sinks(self, sinks):
        self.ctx.set_sinks(sinks)

    def get_sinks(self):
        return self.ctx.get_sinks()

    def set_sink_accel(self, ax, ay, az):
        self.ctx.set_sink_accel(ax, ay, az)

    def density(self):
        self.ctx.density()

    def refresh_eos(self):
        self.ctx.refresh_eos()

    def forces(self):
        self.ctx.forces()

    def kick(self, dt):
        self.ctx.kick(dt)

    def drift(self, dt):
        self.ctx.drift(dt)

    def dt_candidate(self):
        return self.ctx.dt_candidate()

    def synchronize(self):
        self.ctx.synchronize()


def slab_bounds(x_all: np.ndarray, nranks: int) -> np.ndarray:
    """interior slab edges along x with equal particle counts (length nranks-1)"""
    if nranks == 1:
        return np.zeros(0)
    q = np.quantile(x_all, np.arange(1, nranks) / nranks)
    return np.asarray(q, dtype=np.float64)


class DistSim:
    """Runs the reference's step sequence on P ranks.  `gas` holds THIS rank's particles (dict of numpy
    arrays, STATE keys, optional 'gid'); `bounds` are the interior slab edges shared by all ranks."""

    def __init__(self, backend, gas: dict, sinks: dict, bounds: np.ndarray, h: float | None = None,
                 group=None, comm_device=None, migrate: bool = True):
        self.be = backend
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.P = dist.get_world_size(group) if dist.is_initialized() else 1
        self.dev = backend.device
        self.comm_dev = torch.device(comm_device) if comm_device is not None else self.dev
        self.h = float(h if h is not None else backend.params.h)
        self.bounds = torch.as_tensor(np.asarray(bounds, dtype=np.float64), device=self.dev)
        self.migrate = migrate
        backend.set_rank(self.rank, self.P)
        n = int(np.asarray(gas["x"]).size)
        rows = [np.ascontiguousarray(gas[k] if k in gas and gas[k] is not None else np.zeros(n), dtype=np.float64) for k in STATE]
        self.owned = torch.as_tensor(np.stack(rows), device=self.dev)            # [9, n_owned]
        gid = gas.get("gid")
        self.gid = torch.as_tensor(np.asarray(gid if gid is not None else np.arange(n), dtype=np.int64), device=self.dev)
        self.n_owned = n
        self.sinks = {k: np.array(v, dtype=np.float64, copy=True) for k, v in sinks.items()}
        backend.set_sinks(self.sinks)
        self.pos_dirty = True     # ghosts (and the backend's arrays) do not match the owned positions
        self.vel_dirty = False    # ghost v, u, alpha are older than their owners'
        self.in_backend = False   # the current owned state lives in the backend (self.owned is stale)
        self.send_idx = [None] * self.P      # per peer: original ids of my particles it holds as ghosts
        self.ghost_first = [0] * self.P      # per peer: first original id of its ghosts in my context
        self.ghost_count = [0] * self.P
        self.t = 0.0
        self.stats = {"ghosts": 0, "migrated": 0, "exchanges": 0}

    # ---- communication helpers ------------------------------------------------------------------
    def _p2p(self, send: list, recv_counts: list, width: int):
        """send[q]: tensor [width, n_q] (or None) for peer q; returns recv[q]: tensor [width, recv_counts[q]]"""
        recv = [None] * self.P
        ops, keep = [], []
        for q in range(self.P):
            if q == self.rank:
                continue
            if send[q] is not None and send[q].numel() > 0:
                buf = send[q].to(self.comm_dev).contiguous()
                keep.append(buf)
                ops.append(dist.P2POp(dist.isend, buf, q, self.group))
            if recv_counts[q] > 0:
                r = torch.empty((width, recv_counts[q]), dtype=torch.float64, device=self.comm_dev)
                recv[q] = r
                ops.append(dist.P2POp(dist.irecv, r, q, self.group))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        self.stats["exchanges"] += 1
        return [r.to(self.dev) if r is not None else None for r in recv]

    def _all_gather_rows(self, row: torch.Tensor) -> torch.Tensor:
        """every rank contributes one 1-D tensor of equal length -> [P, len] on the host"""
        mine = row.to(self.comm_dev).contiguous()
        out = [torch.empty_like(mine) for _ in range(self.P)]
        dist.all_gather(out, mine, group=self.group)
        return torch.stack(out).cpu()

    def _allreduce(self, vals, op):
        t = torch.tensor(vals, dtype=torch.float64, device=self.comm_dev)
        if self.P > 1:
            dist.all_reduce(t, op=op, group=self.group)
        return t.cpu().numpy()

    # ---- domain bookkeeping -----------------------------------------------------------------------
    def _pull_owned(self):
        """owned state out of the backend: one fused gather of the 9 state fields"""
        self.owned = self.be.gather(STATE, None, self.n_owned)
        self.in_backend = False

    def _migrate(self):
        x = self.owned[0]
        dest = torch.bucketize(x, self.bounds, right=True)
        counts = torch.bincount(dest, minlength=self.P)
        cm = self._all_gather_rows(counts)                      # [P, P], row = sender, host
        mine_out = cm[self.rank].clone(); mine_out[self.rank] = 0
        incoming = cm[:, self.rank].clone(); incoming[self.rank] = 0
        moved = int(cm.sum() - cm.diag().sum())
        if moved == 0:
            return
        payload = torch.cat([self.owned, self.gid.to(torch.float64)[None, :]])     # [10, n]; gid < 2^53 is exact
        send = [None] * self.P
        for q in range(self.P):
            if int(mine_out[q]):
                send[q] = payload[:, dest == q]
        recv = self._p2p(send, [int(v) for v in incoming], 10)
        parts = [payload[:, dest == self.rank]] + [r for r in recv if r is not None]
        allp = torch.cat(parts, dim=1)
        self.owned = allp[:9].contiguous()
        self.gid = allp[9].to(torch.int64)
        self.n_owned = int(allp.shape[1])
        self.stats["migrated"] += moved

    def _exchange_ghosts(self):
        """steps 2-4 of the module docstring: who needs which of my particles, ship them, upload"""
        pos = self.owned[:3]
        if self.n_owned:
            bb = torch.cat([pos.min(dim=1).values, pos.max(dim=1).values])
        else:
            bb = torch.tensor([np.inf] * 3 + [-np.inf] * 3, dtype=torch.float64, device=self.dev)
        boxes = self._all_gather_rows(bb).numpy()                # [P, 6] on the host
        r = 2.0 * self.h * (1.0 + 1e-9)
        me_lo, me_hi = boxes[self.rank, :3], boxes[self.rank, 3:]
        send, counts = [None] * self.P, [0] * self.P
        for q in range(self.P):
            self.send_idx[q] = None
            if q == self.rank or not np.all(np.isfinite(boxes[q])) or not np.all(np.isfinite(boxes[self.rank])):
                continue
            lo, hi = boxes[q, :3] - r, boxes[q, 3:] + r
            if np.any(me_hi < lo) or np.any(me_lo > hi):          # boxes do not touch: nothing to send
                continue
            lo_t = torch.as_tensor(lo, device=self.dev)[:, None]
            hi_t = torch.as_tensor(hi, device=self.dev)[:, None]
            idx = torch.nonzero(((pos >= lo_t) & (pos <= hi_t)).all(dim=0)).flatten()
            counts[q] = int(idx.numel())
            if counts[q]:
                self.send_idx[q] = idx
                send[q] = self.owned[:, idx]
        cm = self._all_gather_rows(torch.tensor(counts, dtype=torch.int64, device=self.dev))
        rc = [int(cm[q, self.rank]) for q in range(self.P)]
        recv = self._p2p(send, rc, 9)
        first = self.n_owned
        parts = [self.owned]
        for q in range(self.P):
            self.ghost_first[q], self.ghost_count[q] = first, rc[q]
            if recv[q] is not None:
                parts.append(recv[q])
            first += rc[q]
        allp = torch.cat(parts, dim=1) if len(parts) > 1 else self.owned
        self.stats["ghosts"] = int(allp.shape[1]) - self.n_owned
        self.be.upload(allp, self.n_owned)
        self.in_backend = True

    def _refresh_ghost_fields(self, names):
        """ship the listed fields of the particles my peers hold as ghosts; scatter what I receive"""
        if self.P == 1:
            return
        send = [self.be.gather(names, idx) if idx is not None else None for idx in self.send_idx]
        recv = self._p2p(send, self.ghost_count, len(names))
        for q in range(self.P):
            if recv[q] is not None:
                self.be.scatter(names, self.ghost_first[q], recv[q])

    # ---- the hot path, distributed -----------------------------------------------------------------
    def evaluate(self):
        """one force evaluation: create_tree..find_forces of the reference, [F]:894-898"""
        be = self.be
        if self.pos_dirty:
            if self.P > 1:
                if self.in_backend:
                    self._pull_owned()
                if self.migrate:
                    self._migrate()
                self._exchange_ghosts()
            elif not self.in_backend:          # single rank: upload once, everything stays on the device
                be.upload(self.owned, self.n_owned)
                self.in_backend = True
            be.density()
            self._refresh_ghost_fields(["rho"])
            if self.P > 1:
                be.refresh_eos()
        else:
            if self.vel_dirty:
                self._refresh_ghost_fields(["vx", "vy", "vz", "u", "alpha"])
            be.density()
            if self.P > 1:
                be.refresh_eos()
        self.pos_dirty = self.vel_dirty = False
        be.forces()
        if self.P > 1:
            s = be.get_sinks()
            tot = self._allreduce(np.concatenate([s["ax"], s["ay"], s["az"]]), dist.ReduceOp.SUM)
            ns = s["ax"].size
            be.set_sink_accel(tot[:ns], tot[ns:2 * ns], tot[2 * ns:])

    def next_dt(self, dt: float) -> float:
        """get_next_timestep, [F]:851-859, with the candidate min-reduced over ranks"""
        cand = self.be.dt_candidate()
        if self.P > 1:
            cand = float(self._allreduce([cand], dist.ReduceOp.MIN)[0])
        p = self.be.params
        if cand > 2 * dt and 1.5 * dt < p.dt_max:
            return 1.5 * dt
        if cand < 0.5 * dt and dt * 0.5 > p.dt_min:
            return 0.5 * dt
        return dt

    def step(self, dt: float) -> float:
        """one iteration of simulate()'s loop body, [F]:889-916; returns the next dt"""
        be = self.be
        self.evaluate()
        be.kick(dt)
        be.drift(dt)
        self.pos_dirty = True
        self.evaluate()
        be.kick(dt)
        self.vel_dirty = True
        self.t += dt
        return self.next_dt(dt)

    def run(self, nsteps: int, dt: float) -> float:
        for _ in range(nsteps):
            dt = self.step(dt)
        return dt

    def gather_state(self) -> dict:
        """owned state + gid of this rank as numpy (for checks and saves)"""
        if self.in_backend:
            self._pull_owned()
            self.in_backend = True      # the backend copy stays valid
        out = {k: self.owned[i].cpu().numpy() for i, k in enumerate(STATE)}
        out["gid"] = self.gid.cpu().numpy()
        return out
