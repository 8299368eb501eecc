"""One-process-per-GPU domain decomposition of the SPH hot path.

Interactions are short range (2h), so the path shards by space: every rank owns the particles of one
slab along x (slab edges = particle-count quantiles) and keeps GHOST copies of the other ranks'
particles that lie within 2h of its own particles' bounding box.  The owned particles never leave
the GPU context; per step (= the reference's loop body, [F]:889-916) the ranks exchange only:

  start of step (positions as at the end of the last step)
    1. point-to-point: v, u, alpha of the ghosts (their owners kicked them)
       density + EOS + forces
    2. all-gather [199 doubles/rank]: partial sink accelerations, the dt candidate left by the previous
       step -> summed / min-reduced on the device in rank order, dt rule [F]:855-858; and every rank's
       PREDICTED bounding box after the coming kick + drift
  kick, drift
  end of step (positions changed)
    3. (only on migration steps and for the octree paths) all-gather: bounding boxes of the owned particles;
       otherwise the boxes predicted in 2. are used (supersets: a few ghosts too many, never one too few)
    4. all-gather: how many particles each rank sends to each peer (host: message sizes)
    5. point-to-point: the 9 state fields of the particles inside (peer box + 2h); they replace the
       ghost slots of the context (sph_replace_ghosts_dev)
       density of the owned particles
    6. point-to-point: rho of the same particles -> ghost rho; EOS, forces
    7. all-gather [199 doubles/rank]: partial sink accelerations
  kick; the local dt candidate stays on the device until 2. of the next step.

With SPH_FLAG_SELF_GRAVITY every rank additionally all-gathers {x, y, z, m} of all particles after a drift and builds
the same Barnes-Hut tree as the undecomposed run (replicated build, shared walk; `_gravity_sources`).

Every `migrate_every` steps the particles that left their slab change owner first (the ghost
selection uses bounding boxes, not slab edges, so ownership only matters for load balance).

torch.distributed carries every exchange: backend "nccl" (= RCCL over xGMI; every pair of GPUs has
a direct link, so the halo messages of different pairs do not share links) on a GPU node, "gloo"
for the CPU tests.  The arithmetic is done by a *backend object*: `HipBackend` (the C ABI, device
memory, running on torch's current stream so that no host synchronisation separates the library's
kernels from torch's collectives) in production; tests plug in an oracle-based backend to exercise
this orchestration on CPUs.  Nothing in this module computes physics.
"""
from __future__ import annotations

import contextlib
import inspect
import time

import numpy as np
import torch
import torch.distributed as dist

STATE = ["x", "y", "z", "vx", "vy", "vz", "u", "m", "alpha"]
PARTIALS = 199          # SPH_PARTIALS: ax[64] ay[64] az[64] of the sinks, the dt candidate, the predicted bounding box
DT_SLOT = 192           # position of the dt candidate in that vector
ACC_PARTIALS = 448      # SPH_ACC_PARTIALS: per sink m, m x, m y, m z, m vx, m vy, m vz of the accreted particles


class HipBackend:
    """The C-ABI context as seen by the orchestrator.  Tensors are float64 on `device`.

    The context runs on torch's current stream (sph_set_stream): the library's kernels, torch's own
    kernels and the waits torch inserts around its collectives are all ordered on that one stream,
    so buffers are handed back and forth without host synchronisation."""

    def __init__(self, device_index: int = 0, **param_overrides):
        from . import capi
        self.capi = capi
        self.ctx = capi.Context(device=device_index, **param_overrides)
        self.device = torch.device("cuda", device_index)
        with torch.cuda.device(self.device):
            self.ctx.set_stream(torch.cuda.current_stream(self.device).cuda_stream)
        self.n_owned = 0
        self._reserved = 0

    @property
    def params(self):
        return self.ctx.params

    def set_rank(self, rank, nranks):
        self.ctx.set_rank(rank, nranks)

    @property
    def variable(self) -> bool:
        return bool(self.ctx.params.flags & self.capi.FLAG_VARIABLE_H)

    def upload(self, state: torch.Tensor):
        """state: [9, n] (rows in STATE order; a 10th row h with variable smoothing lengths): this rank's owned
        particles; ghosts come later"""
        state = state.contiguous()
        n = int(state.shape[1])
        self.n_owned = n
        if n + n // 8 + 32768 > self._reserved:         # room for the ghost swaps; grows rarely (a re-allocation)
            self._reserved = n + n // 4 + 65536
            self.ctx.reserve(self._reserved)
        self.ctx.upload_dev(n, [state[k].data_ptr() for k in range(9)])
        if self.variable and n:
            self.ctx.upload_field_dev("h", state[9].data_ptr(), n)

    def set_numbers(self, first: int, numbers: torch.Tensor):
        numbers = numbers.to(torch.int64).contiguous()
        self.ctx.set_numbers_dev(first, int(numbers.numel()), numbers.data_ptr())

    def update_h(self):
        self.ctx.update_h()

    def owned_bbox(self) -> torch.Tensor:
        out = torch.empty(6, dtype=torch.float64, device=self.device)
        self.ctx.owned_bbox(out.data_ptr())
        return out

    def select_boxes(self, boxes: np.ndarray) -> list:
        """per box {lo xyz, hi xyz}: original ids (ascending, int64 tensor) of the owned particles inside"""
        counts = self.ctx.select_boxes(boxes)
        out = []
        for b, cnt in enumerate(counts):
            ids = torch.empty(int(cnt), dtype=torch.int64, device=self.device)
            self.ctx.selected_ids_dev(b, int(cnt), ids.data_ptr())
            out.append(ids)
        return out

    def replace_ghosts(self, state: torch.Tensor):
        state = state.contiguous()
        self.ctx.replace_ghosts_dev(int(state.shape[1]), state.data_ptr())

    def gather(self, names, ids: torch.Tensor | None = None, count: int | None = None) -> torch.Tensor:
        """[len(names), count] values of the particles with original ids `ids` (None: 0..count-1)"""
        count = int(ids.numel()) if ids is not None else int(count)
        out = torch.empty((len(names), count), dtype=torch.float64, device=self.device)
        if count:
            self.ctx.gather_fields_dev(names, count, ids.data_ptr() if ids is not None else 0, out.data_ptr())
        return out

    def scatter(self, names, first: int, vals: torch.Tensor):
        vals = vals.contiguous()
        if vals.numel():
            self.ctx.scatter_fields_dev(names, first, vals.shape[1], vals.data_ptr())

    def set_sinks(self, sinks):
        self.ctx.set_sinks(sinks)

    def get_sinks(self):
        return self.ctx.get_sinks()

    def density(self):
        self.ctx.density()

    def refresh_eos(self):
        self.ctx.refresh_eos(ghosts_only=True)      # owned records: written by the density pass itself

    def forces(self):
        self.ctx.forces()

    def set_boundary_boxes(self, boxes):
        self.ctx.set_boundary_boxes(boxes)

    @property
    def gravity(self) -> bool:
        return bool(self.ctx.params.flags & self.capi.FLAG_SELF_GRAVITY)

    def set_gravity_sources(self, src: torch.Tensor, lo_hi: np.ndarray):
        """src: [N, 4] records {x, y, z, m} of EVERY rank's particles; kept alive here until replaced"""
        self._grav_src = src.contiguous()
        self.ctx.set_gravity_sources_dev(int(self._grav_src.shape[0]), self._grav_src.data_ptr(), lo_hi)

    @property
    def accrete(self) -> bool:
        return bool(self.ctx.params.flags & self.capi.FLAG_ACCRETE_CULL)

    @property
    def sink_creation(self) -> bool:
        return bool(self.ctx.params.flags & self.capi.FLAG_SINK_CREATION)

    def sink_candidate(self) -> torch.Tensor:
        out = torch.empty(9, dtype=torch.float64, device=self.device)
        self.ctx.sink_candidate_dev(out.data_ptr())
        return out

    def add_sink_checked(self, cand: torch.Tensor) -> bool:
        cand = cand.contiguous()
        return self.ctx.add_sink_checked_dev(cand.data_ptr())

    def accrete_mark(self, src_offset: int) -> torch.Tensor:
        out = torch.empty(ACC_PARTIALS, dtype=torch.float64, device=self.device)
        self.ctx.accrete_mark_dev(src_offset, out.data_ptr())
        return out

    def accrete_apply(self, allp: torch.Tensor):
        """-> (owned particles removed, keep flags [n_owned before] as a bool tensor)"""
        allp = allp.contiguous()
        keep = torch.empty(max(self.n_owned, 1), dtype=torch.int32, device=self.device)
        removed = self.ctx.accrete_apply_dev(allp.data_ptr(), int(allp.shape[0]), int(allp.shape[1]), keep.data_ptr())
        keep = keep[:self.n_owned].bool()
        self.n_owned -= removed
        return removed, keep

    def forces_interior(self):
        self.ctx.forces_part(1)

    def forces_boundary(self):
        self.ctx.forces_part(2)

    def set_dt(self, dt, t):
        self.ctx.set_dt(dt, t)

    def get_dt(self):
        return self.ctx.get_dt()

    def kick(self):
        self.ctx.kick_devdt()

    def drift(self):
        self.ctx.drift_devdt()

    def dt_candidate_local(self):
        self.ctx.dt_candidate_dev()

    def kick_drift(self):
        self.ctx.kick_drift_devdt()          # one pass over the state, bitwise kick() + drift()

    def kick_dt_candidate(self):
        self.ctx.kick_dt_candidate_dev()     # bitwise kick() + dt_candidate_local()

    def pack_partials(self, predict_box: bool = True) -> torch.Tensor:
        out = torch.empty(PARTIALS, dtype=torch.float64, device=self.device)
        self.ctx.pack_partials_dev(out.data_ptr(), predict_box)
        return out

    def apply_partials(self, allp: torch.Tensor, apply_dt: bool):
        allp = allp.contiguous()
        self.ctx.apply_partials_dev(allp.data_ptr(), int(allp.shape[0]), int(allp.shape[1]), apply_dt)

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
                 group=None, comm_device=None, migrate: bool = True, migrate_every: int = 32):
        self.be = backend
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.P = dist.get_world_size(group) if dist.is_initialized() else 1
        self.dev = backend.device
        self.comm_dev = torch.device(comm_device) if comm_device is not None else self.dev
        self.h = float(h if h is not None else backend.params.h)
        self.bounds = torch.as_tensor(np.asarray(bounds, dtype=np.float64), device=self.dev)
        self.migrate = migrate
        self.migrate_every = max(1, int(migrate_every))
        backend.set_rank(self.rank, self.P)
        n = int(np.asarray(gas["x"]).size)
        # variable smoothing lengths ("SUMMER_SPH - Variable.f90"): h travels with the state, ghosts also carry their
        # global particle number (the pair rule of [V]:383 depends on it) and the octree of ALL particles supplies the
        # leaf boxes (the all-gathered sources of the self-gravity path)
        self.variable = bool(getattr(backend, "variable", False))
        self.fields = STATE + (["h"] if self.variable else [])
        rows = [np.ascontiguousarray(gas[k] if k in gas and gas[k] is not None else np.zeros(n), dtype=np.float64) for k in self.fields]
        self.owned = torch.as_tensor(np.stack(rows), device=self.dev)            # [9|10, n_owned]; stale once in_backend
        gid = gas.get("gid")
        self.gid = torch.as_tensor(np.asarray(gid if gid is not None else np.arange(n), dtype=np.int64), device=self.dev)
        self.n_owned = n
        backend.set_sinks({k: np.array(v, dtype=np.float64, copy=True) for k, v in sinks.items()})
        self.pos_dirty = True     # ghosts do not match the owned positions
        self.vel_dirty = False    # ghost v, u, alpha are older than their owners'
        self.in_backend = False   # the owned particles live in the backend (self.owned is stale)
        self.dt_pending = False   # a local dt candidate waits for the next reduction
        self.since_migrate = 0
        self.send_idx = [None] * self.P      # per peer: original ids of my particles it holds as ghosts
        self.cap_send = [0] * self.P         # per peer: particles the next ghost message to / from it has room for
        self.cap_recv = [0] * self.P         # (both sides of a pair derive the same number from their last message)
        self.ghost_first = [0] * self.P      # per peer: first original id of its ghosts in my context
        self.ghost_count = [0] * self.P
        self.t = 0.0
        self.stats = {"ghosts": 0, "migrated": 0, "exchanges": 0, "migrations": 0}
        self.gravity = bool(getattr(backend, "gravity", False))     # Barnes-Hut self-gravity: replicated tree (below)
        # sink accretion + boundary cull at the end of a step ([F]:918-920); decided on the octree of all particles,
        # i.e. it needs the all-gathered sources of the self-gravity path
        self.accrete = bool(getattr(backend, "accrete", False))
        if self.accrete and (self.P == 1 or not self.gravity):
            raise ValueError("accretion through DistSim needs several ranks and SPH_FLAG_SELF_GRAVITY (the shared octree); "
                             "a single GPU runs it inside sph_step / sph_run")
        self.stats_removed = 0
        # check_sink_creation ([V]:549-597): the first particle by global number that qualifies, found with one all-gather
        self.sink_creation = bool(getattr(backend, "sink_creation", False)) and self.variable
        self.counts_all = None    # owned particles of every rank (changes with migrations only)
        self.sources_valid = False  # the all-gathered {x, y, z, m} of all particles match the current positions and owners
        self.boxes = None         # every rank's owned bounding box at the last ghost exchange (host)
        self._pred, self._pred_event, self._pred_pinned = None, None, None
        self.pred_for_drift = False   # the boxes predicted at the last reduction describe the positions after the drift just done
        self.profile = False      # True: synchronise at phase boundaries and accumulate wall time per phase
        self.phase_s = {}
        self._gather_into = True  # all_gather_into_tensor until the backend refuses it
        self._pack_has_switch = "predict_box" in inspect.signature(backend.pack_partials).parameters
        if self.P > 1:
            self.prime()

    @contextlib.contextmanager
    def _phase(self, name):
        if not self.profile:
            yield
            return
        self.be.synchronize()
        t0 = time.perf_counter()
        yield
        self.be.synchronize()
        self.phase_s[name] = self.phase_s.get(name, 0.0) + time.perf_counter() - t0

    # ---- communication helpers ------------------------------------------------------------------
    def _p2p_start(self, send: list, recv_counts: list, width: int):
        """send[q]: tensor [width, n_q] (or None) for peer q; posts the sends and receives and returns a handle"""
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
        works = dist.batch_isend_irecv(ops) if ops else []
        self.stats["exchanges"] += 1
        return works, recv, keep

    def _p2p_finish(self, handle):
        """waits for a _p2p_start; returns recv[q]: tensor [width, recv_counts[q]] on the compute device (or None)"""
        works, recv, _keep = handle
        for w in works:
            w.wait()
        return [r.to(self.dev) if r is not None else None for r in recv]

    def _p2p(self, send: list, recv_counts: list, width: int):
        return self._p2p_finish(self._p2p_start(send, recv_counts, width))

    def prime(self):
        """first use of a torch kernel loads its code object (tens of ms): do that for the ops of the rare migration
        path now instead of in the middle of a run"""
        x = torch.linspace(-1.0, 1.0, 16, dtype=torch.float64, device=self.dev)
        b = torch.zeros(1, dtype=torch.float64, device=self.dev)
        dest = torch.bucketize(x, b, right=True)
        torch.bincount(dest, minlength=2)
        pay = torch.cat([x[None, :].repeat(9, 1), torch.arange(16, device=self.dev).to(torch.float64)[None, :]])
        part = torch.cat([pay[:, dest == 0], pay[:, dest == 1]], dim=1)
        part[:9].contiguous(); part[9].to(torch.int64)
        # ... and the first collective / point-to-point message between two ranks sets up their connection: one of each
        # with the slab neighbours now
        self._all_gather(torch.zeros(PARTIALS, dtype=torch.float64, device=self.dev))
        self._all_gather(torch.zeros(self.P, dtype=torch.int64))
        one = torch.zeros((1, 8), dtype=torch.float64, device=self.dev)
        send = [one if abs(q - self.rank) == 1 else None for q in range(self.P)]
        self._p2p(send, [8 if abs(q - self.rank) == 1 else 0 for q in range(self.P)], 1)

    def _all_gather(self, row: torch.Tensor) -> torch.Tensor:
        """every rank contributes one 1-D tensor of equal length -> [P, len] on the communication device"""
        mine = row.to(self.comm_dev).contiguous()
        if self.P == 1:
            return mine[None, :]
        out = torch.empty((self.P, mine.numel()), dtype=mine.dtype, device=self.comm_dev)
        if self._gather_into:
            try:
                dist.all_gather_into_tensor(out.view(-1), mine, group=self.group)
                return out
            except (RuntimeError, NotImplementedError):
                self._gather_into = False
        dist.all_gather(list(out.unbind(0)), mine, group=self.group)
        return out

    # ---- domain bookkeeping -----------------------------------------------------------------------
    def _pull_owned(self):
        """owned state out of the backend: one fused gather of the 9 state fields"""
        self.owned = self.be.gather(self.fields, None, self.n_owned)
        self.in_backend = False

    def _migrate(self):
        """particles that left their slab change owner (host-sized messages; only every migrate_every steps)"""
        x = self.owned[0]
        dest = torch.bucketize(x, self.bounds, right=True)
        counts = torch.bincount(dest, minlength=self.P)
        cm = self._all_gather(counts).cpu()                       # [P, P], row = sender
        mine_out = cm[self.rank].clone(); mine_out[self.rank] = 0
        incoming = cm[:, self.rank].clone(); incoming[self.rank] = 0
        moved = int(cm.sum() - cm.diag().sum())
        self.stats["migrations"] += 1
        if moved == 0:
            return
        payload = torch.cat([self.owned, self.gid.to(torch.float64)[None, :]])     # [nf + 1, n]; gid < 2^53 is exact
        send = [None] * self.P
        for q in range(self.P):
            if int(mine_out[q]):
                send[q] = payload[:, dest == q]
        nf = len(self.fields)
        recv = self._p2p(send, [int(v) for v in incoming], nf + 1)
        parts = [payload[:, dest == self.rank]] + [r for r in recv if r is not None]
        allp = torch.cat(parts, dim=1)
        self.owned = allp[:nf].contiguous()
        self.gid = allp[nf].to(torch.int64)
        self.n_owned = int(allp.shape[1])
        self.stats["migrated"] += moved

    def _exchange_ghosts(self):
        """steps 3-5 of the module docstring: who needs which of my particles, ship them, swap them in"""
        be = self.be
        use_pred = self.pred_for_drift and self._pred is not None and not (self.gravity or self.variable)
        self.pred_for_drift = False
        if use_pred:
            if self._pred_event is not None:
                self._pred_event.synchronize()
            boxes = np.array(self._pred.numpy(), dtype=np.float64, copy=True)
            use_pred = bool(np.all(np.isfinite(boxes) | np.isinf(boxes)))     # NaN rows: some rank had no prediction
        if use_pred:
            # supersets of the true boxes: a few ghosts too many, never one too few; the octree paths (self-gravity,
            # variable h) need the exact global box and take the exchange below instead
            r = 2.0 * self.h * (1.0 + 1e-9)
        elif self.variable:
            # i and j interact within 2 max(h_i, h_j): the ghost layer is as wide as twice the largest h anywhere
            hm = be.gather(["h"], None, self.n_owned).max() if self.n_owned else torch.zeros((), dtype=torch.float64, device=self.dev)
            bh = self._all_gather(torch.cat([be.owned_bbox(), hm.reshape(1)])).cpu().numpy()
            boxes = np.ascontiguousarray(bh[:, :6])
            r = 2.0 * float(bh[:, 6].max()) * (1.0 + 1e-9)
        else:
            boxes = self._all_gather(be.owned_bbox()).cpu().numpy()     # [P, 6] on the host
            r = 2.0 * self.h * (1.0 + 1e-9)
        self.boxes = boxes
        me_lo, me_hi = boxes[self.rank, :3], boxes[self.rank, 3:]
        mine_ok = bool(np.all(np.isfinite(boxes[self.rank])))
        peers, sel = [], []
        for q in range(self.P):
            self.send_idx[q] = None
            if q == self.rank or not mine_ok or not np.all(np.isfinite(boxes[q])):
                continue
            lo, hi = boxes[q, :3] - r, boxes[q, 3:] + r
            # one expression both ranks of a pair evaluate identically (operands ordered by rank): each side posts a receive
            # for the other's message, so the two decisions must never differ by an ulp at gap == r
            a_, b_ = (boxes[q], boxes[self.rank]) if q < self.rank else (boxes[self.rank], boxes[q])
            gap = np.maximum(b_[:3] - a_[3:], a_[:3] - b_[3:])
            if not np.all(gap <= r):                              # boxes do not touch: nothing to send
                continue
            peers.append(q)
            sel.append(np.concatenate([lo, hi]))
        counts = [0] * self.P
        if peers:
            for q, ids in zip(peers, be.select_boxes(np.stack(sel))):
                counts[q] = int(ids.numel())
                if counts[q]:
                    self.send_idx[q] = ids
        # The payload travels without a size exchange (as in csrc/halo.hip): both sides of a pair agree on the room of the
        # next message (the count of their last message x 1.25 + 256; 0 at first), the first two doubles say how many
        # particles it holds, the rows follow.  A count that does not fit is known to both sides -- to the sender from its
        # count, to the receiver from the header -- and only those pairs exchange the rows again at their exact size.
        nf = len(self.fields)
        width = nf + (1 if self.variable else 0)            # variable h: + the global particle number of every ghost
        rows = {}
        msg = [None] * self.P
        room = [0] * self.P
        for q in peers:
            idx = self.send_idx[q]
            if idx is not None:
                rows[q] = be.gather(self.fields, idx)
                if self.variable:
                    rows[q] = torch.cat([rows[q], self.gid[idx].to(torch.float64)[None, :]])
            cap = self.cap_send[q]
            m = torch.zeros((1, 2 + width * cap), dtype=torch.float64, device=self.dev)
            m[0, 0] = counts[q]
            if 0 < counts[q] <= cap:
                m[0, 2:2 + width * counts[q]] = rows[q].reshape(-1)
            msg[q] = m
            room[q] = 2 + width * self.cap_recv[q]
        got = self._p2p(msg, room, 1)
        heads = torch.stack([got[q][0, 0] for q in peers]).cpu() if peers else torch.zeros(0)      # the one read-back
        rc = [0] * self.P
        for k, q in enumerate(peers):
            rc[q] = int(heads[k])
        again_send = [rows[q] if (q in peers and counts[q] > self.cap_send[q]) else None for q in range(self.P)]
        again_recv = [rc[q] if (q in peers and rc[q] > self.cap_recv[q]) else 0 for q in range(self.P)]
        if any(t is not None for t in again_send) or any(again_recv):
            late = self._p2p(again_send, again_recv, width)
        else:
            late = [None] * self.P
        recv = [None] * self.P
        for q in peers:
            if rc[q] == 0:
                continue
            recv[q] = late[q] if again_recv[q] else got[q][0, 2:2 + width * rc[q]].reshape(width, rc[q])
        for q in peers:
            self.cap_send[q] = counts[q] + counts[q] // 4 + 256
            self.cap_recv[q] = rc[q] + rc[q] // 4 + 256
        # every ghost I am about to receive lies inside its owner's box: particles farther than 2h from all of them
        # cannot have a ghost neighbour (their forces do not wait for the ghost fields)
        senders = [q for q in range(self.P) if rc[q] > 0]
        be.set_boundary_boxes(boxes[senders] if senders else np.zeros((0, 6)))
        first = self.n_owned
        parts = []
        for q in range(self.P):
            self.ghost_first[q], self.ghost_count[q] = first, rc[q]
            if recv[q] is not None:
                parts.append(recv[q])
            first += rc[q]
        if len(parts) == 1:
            ghosts = parts[0]
        elif parts:
            ghosts = torch.cat(parts, dim=1)
        else:
            ghosts = torch.empty((nf + (1 if self.variable else 0), 0), dtype=torch.float64, device=self.dev)
        self.stats["ghosts"] = int(ghosts.shape[1])
        be.replace_ghosts(ghosts[:nf])
        if self.variable:
            be.set_numbers(0, self.gid)
            be.set_numbers(self.n_owned, ghosts[nf].to(torch.int64))

    def _gravity_sources(self):
        """Self-gravity is long range: every rank gets {x, y, z, m} of ALL particles (one all-gather, padded to the
        largest rank) and builds the same Barnes-Hut tree as the undecomposed run would -- same particles, same root
        box, hence the same nodes and the same accepted set for every target -- and walks it for its own particles.
        The tree build is replicated work; the walk, which dominates, is shared."""
        be = self.be
        if self.counts_all is None:
            self.counts_all = [int(v) for v in self._all_gather(torch.tensor([self.n_owned], dtype=torch.int64)).cpu()[:, 0]]
        mine = be.gather(["x", "y", "z", "m"], None, self.n_owned)            # [4, n_owned]
        maxn = max(max(self.counts_all), 1)
        buf = torch.zeros((4, maxn), dtype=torch.float64, device=self.dev)
        buf[:, :self.n_owned] = mine
        allb = self._all_gather(buf.view(-1)).to(self.dev).view(self.P, 4, maxn)
        src = torch.cat([allb[r, :, :self.counts_all[r]].T for r in range(self.P)], dim=0).contiguous()   # [N, 4]
        ok = [r for r in range(self.P) if self.counts_all[r] > 0]
        lo_hi = np.concatenate([self.boxes[ok, :3].min(0), self.boxes[ok, 3:].max(0)])
        be.set_gravity_sources(src, lo_hi)

    def _refresh_ghost_start(self, names):
        """ship the listed fields of the particles my peers hold as ghosts (posts the messages)"""
        if self.P == 1:
            return None
        send = [self.be.gather(names, idx) if idx is not None else None for idx in self.send_idx]
        return names, self._p2p_start(send, self.ghost_count, len(names))

    def _refresh_ghost_finish(self, handle):
        """scatter what the peers sent into my ghost slots"""
        if handle is None:
            return
        names, h = handle
        recv = self._p2p_finish(h)
        for q in range(self.P):
            if recv[q] is not None:
                self.be.scatter(names, self.ghost_first[q], recv[q])

    def _refresh_ghost_fields(self, names):
        self._refresh_ghost_finish(self._refresh_ghost_start(names))

    def _reduce(self, before_drift: bool = True):
        """sink accelerations summed over ranks; a pending dt candidate min-reduced and the dt rule applied.  The same
        message carries every rank's PREDICTED bounding box after the coming kick + drift (the rates are known now), so
        that the ghost exchange that follows the drift needs no exchange of bounding boxes of its own -- asked for only
        where a drift follows and somebody reads it (it costs a pass over the particles)."""
        predict = before_drift and self.P > 1 and not (self.gravity or self.variable)
        part = self.be.pack_partials(predict) if self._pack_has_switch else self.be.pack_partials()
        allc = self._all_gather(part)
        allp = allc.to(self.dev)
        self.be.apply_partials(allp, self.dt_pending)
        self.dt_pending = False
        pred = allc[:, DT_SLOT + 1:DT_SLOT + 7]
        if pred.device.type == "cpu":
            self._pred, self._pred_event = pred.clone(), None
        else:                                   # to the host without stalling the stream: read after the drift
            if self._pred_pinned is None:
                self._pred_pinned = torch.empty((self.P, 6), dtype=torch.float64, pin_memory=True)
            self._pred_pinned.copy_(pred, non_blocking=True)
            self._pred_event = torch.cuda.Event()
            self._pred_event.record()
            self._pred = self._pred_pinned

    # ---- the hot path, distributed -----------------------------------------------------------------
    def evaluate(self, before_drift: bool = True):
        """one force evaluation: create_tree..find_forces of the reference, [F]:894-898"""
        be = self.be
        if self.pos_dirty:
            if not self.in_backend or (self.P > 1 and self.migrate and self.since_migrate >= self.migrate_every):
                if self.P > 1 and self.migrate:
                    if self.in_backend:
                        with self._phase("pull_owned"):
                            self._pull_owned()
                    with self._phase("migrate"):
                        self._migrate()
                    self.since_migrate = 0
                    self.pred_for_drift = False      # ownership changed: the predicted boxes are void
                with self._phase("upload"):
                    be.upload(self.owned)
                self.in_backend = True
                self.counts_all = None
                self.sources_valid = False
            if self.P > 1:
                with self._phase("ghost_exchange"):
                    self._exchange_ghosts()
                if (self.gravity or self.variable) and not self.sources_valid:
                    with self._phase("gravity_sources"):
                        self._gravity_sources()
                    self.sources_valid = True
            with self._phase("compute"):
                be.density()
            pending, tag = (self._refresh_ghost_start(["rho", "omega"] if self.variable else ["rho"]) if self.P > 1 else None), "ghost_rho"
        else:
            # the density sum needs positions and masses only: it runs while the ghosts' v, u, alpha travel
            with self._phase("ghost_vel"):
                pending = self._refresh_ghost_start(["vx", "vy", "vz", "u", "alpha"]) if self.vel_dirty else None
            tag = "ghost_vel"
            with self._phase("compute"):
                be.density()
        self.pos_dirty = self.vel_dirty = False
        if self.P > 1 and not self.gravity and not self.variable:
            # ... and so do the forces of the particles that cannot see a ghost; the rest follows once the ghost
            # fields have arrived and the EOS of the ghosts is refreshed
            with self._phase("compute"):
                be.forces_interior()
            with self._phase(tag):
                self._refresh_ghost_finish(pending)
            with self._phase("compute"):
                be.refresh_eos()
                be.forces_boundary()
        elif self.P > 1:
            with self._phase(tag):
                self._refresh_ghost_finish(pending)
            with self._phase("compute"):
                be.refresh_eos()
                be.forces()
        else:
            with self._phase("compute"):
                be.forces()
        with self._phase("reduce"):
            self._reduce(before_drift)

    def _step(self):
        """one iteration of simulate()'s loop body, [F]:889-916, dt and t on the backend"""
        be = self.be
        self.evaluate()
        with self._phase("compute"):
            if hasattr(be, "kick_drift"):
                be.kick_drift()
            else:
                be.kick()
                be.drift()
        self.pos_dirty = True
        self.sources_valid = False
        self.pred_for_drift = True       # the reduction of the evaluation above predicted where this drift takes everybody
        self.since_migrate += 1
        self.evaluate(before_drift=False)     # a kick follows, no drift: nobody reads a predicted box
        with self._phase("compute"):
            if hasattr(be, "kick_dt_candidate"):
                be.kick_dt_candidate()
            else:
                be.kick()
                be.dt_candidate_local()      # get_next_timestep's local part, [F]:845-851; reduced with the next exchange
        self.vel_dirty = True
        self.dt_pending = True
        if self.variable:
            with self._phase("compute"):
                be.update_h()                # calc_smoothing, [V]:1152; the ghosts' h is stale now: full exchange next
            self.pos_dirty = True
            self.vel_dirty = False
        if self.sink_creation and self.P > 1:
            with self._phase("accrete"):
                cands = self._all_gather(be.sink_candidate()).to(self.dev)         # [P, 9]; row 0 = particle number or +inf
                be.add_sink_checked(cands[torch.argmin(cands[:, 0])])
        if self.accrete and self.P > 1:
            with self._phase("accrete"):
                self._accrete_and_cull()

    def _accrete_and_cull(self):
        """every rank decides for its own particles on the shared octree; the sums over the accreted particles are
        all-gathered and added in rank order, so every rank updates the (replicated) sinks identically"""
        be = self.be
        off = sum(self.counts_all[:self.rank])
        allp = self._all_gather(be.accrete_mark(off)).to(self.dev)
        removed, keep = be.accrete_apply(allp)
        if removed:
            self.gid = self.gid[keep]
        self.n_owned = int(self.gid.numel())
        self.stats_removed += removed
        self.counts_all = None
        self.sources_valid = False
        self.pos_dirty = True          # the ghosts were dropped with the accreted particles: exchange before the next pass
        self.vel_dirty = False

    def _finish_dt(self):
        """reduce a pending dt candidate now (end of a run): t += dt and [F]:855-858 on every rank"""
        if not self.dt_pending:
            return
        with self._phase("reduce"):
            allp = self._all_gather(self.be.pack_partials()).to(self.dev)
            # the sink accelerations in the blocks are the totals every rank already holds: keep them
            # (summing them again would multiply by P), only the dt part of the blocks is applied
            mine = allp[self.rank:self.rank + 1].clone()
            mine[0, DT_SLOT] = allp[:, DT_SLOT].min()
            self.be.apply_partials(mine, True)
            self.dt_pending = False

    def run(self, nsteps: int, dt: float) -> float:
        self.be.set_dt(dt, self.t)
        for _ in range(nsteps):
            self._step()
        self._finish_dt()
        dt, self.t = self.be.get_dt()
        return dt

    def step(self, dt: float) -> float:
        """one step; returns the next dt"""
        return self.run(1, dt)

    def gather_state(self) -> dict:
        """owned state + gid of this rank as numpy (for checks and saves)"""
        if self.in_backend:
            self._pull_owned()
            self.in_backend = True      # the backend copy stays valid
        out = {k: self.owned[i].cpu().numpy() for i, k in enumerate(STATE)}
        out["gid"] = self.gid.cpu().numpy()
        return out
