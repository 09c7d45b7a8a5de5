"""Row-partitioned Lanczos over several GPUs of one node: one process per GPU,
collectives through torch.distributed (backend "nccl" = RCCL over xGMI).

The reference has no multi-GPU code; the only thing to agree with is the
single-GPU result.  The matrix is cut into contiguous row slabs.  Per iteration
the ranks exchange the current Lanczos vector (halo exchange with the two
neighbouring slabs when the matrix is banded, all-gather otherwise) and
all-reduce two small buffers: alpha, then the reorthogonalisation coefficients
together with w.w (the norm after the pass follows from them).  Without
reorthogonalisation the basis stays unnormalised and ONE reduction of two
doubles per iteration is enough.  Everything else runs inside each rank's engine through the
phase-level C ABI (include/lanczos_hip.h).

The host logic here (partition, column remap, exchange plan, iteration order)
does not touch the GPU itself; it drives any object with the phase interface,
which is how the CPU tests exercise it with the gloo backend.
"""
from __future__ import annotations

import os
import time
from dataclasses import dataclass

import numpy as np


def partition_rows(n: int, world: int) -> list[int]:
    """Contiguous slabs: rank p owns rows [b[p], b[p+1])."""
    return [(p * n) // world for p in range(world + 1)]


def partition_rows_by_nnz(row_head: np.ndarray, world: int) -> list[int]:
    """Contiguous slabs with about equal nnz (for irregular matrices)."""
    n = row_head.shape[0] - 1
    nnz = int(row_head[-1])
    cuts = [0]
    for p in range(1, world):
        cuts.append(int(np.searchsorted(row_head, (p * nnz) // world, side="left")))
    cuts.append(n)
    for p in range(1, world + 1):
        cuts[p] = max(cuts[p], cuts[p - 1])
    return cuts


@dataclass
class SlabPlan:
    """How one rank's slab addresses the vector it multiplies with."""
    mode: str            # "halo" or "allgather"
    n_local: int
    n_ext: int           # length of the SpMV input buffer
    row_begin: int
    halo_lo: int         # entries needed from the end of the previous slab
    halo_hi: int         # entries needed from the start of the next slab
    column_index: np.ndarray  # remapped into [0, n_ext)
    remote_base: int = -1     # halo form: index of the first remote entry (n_local, or past the padding
                              # of the vector itself when in_place)
    in_place: bool = False    # the remote entries live in the tail of the basis vector the SpMV reads


def needed_halo(cols: np.ndarray, row_begin: int, row_end: int) -> tuple[int, int]:
    """(how far below row_begin, how far beyond row_end) the slab's columns reach."""
    if cols.size == 0:
        return 0, 0
    lo = int(cols.min())
    hi = int(cols.max())
    return max(row_begin - lo, 0), max(hi - (row_end - 1), 0)


def plan_slab(cols: np.ndarray, bounds: list[int], rank: int, mode: str, remote_base: int = -1) -> SlabPlan:
    """Remap global column indices for `mode`.

    halo:      x_ext = [own slab | tail of slab rank-1 (halo_lo) | head of slab rank+1 (halo_hi)];
               with remote_base >= n_local the two remote parts start there instead of at n_local
               (the gap stays unaddressed): the layout of a basis vector with the remote entries
               in its own tail
    allgather: x_ext = the whole vector in slab order, every slab padded to the longest one
    """
    world = len(bounds) - 1
    r0, r1 = bounds[rank], bounds[rank + 1]
    n_local = r1 - r0
    if mode == "halo":
        base = n_local if remote_base < 0 else remote_base
        assert base >= n_local
        hlo, hhi = needed_halo(cols, r0, r1)
        out = np.empty_like(cols)
        own = (cols >= r0) & (cols < r1)
        below = cols < r0
        above = cols >= r1
        out[own] = cols[own] - r0
        out[below] = base + (cols[below] - (r0 - hlo))
        out[above] = base + hlo + (cols[above] - r1)
        return SlabPlan("halo", n_local, base + hlo + hhi, r0, hlo, hhi, out, base, remote_base >= 0)
    pad = max(bounds[p + 1] - bounds[p] for p in range(world))
    owner = np.searchsorted(np.asarray(bounds[1:]), cols, side="right")
    out = owner * pad + (cols - np.asarray(bounds)[owner])
    return SlabPlan("allgather", n_local, pad * world, r0, 0, 0, out.astype(np.int64))


def halo_feasible(hlo: int, hhi: int, bounds: list[int], rank: int) -> bool:
    """The halo form needs every remote column to live in an adjacent slab."""
    world = len(bounds) - 1
    if hlo and (rank == 0 or hlo > bounds[rank] - bounds[rank - 1]):
        return False
    if hhi and (rank == world - 1 or hhi > bounds[rank + 2] - bounds[rank + 1]):
        return False
    return True


class Comm:
    """torch.distributed on tensors as they are (nccl on GPU tensors, gloo on CPU tensors)."""

    def __init__(self, dist, torch, device="cpu"):
        self.dist, self.torch, self.device = dist, torch, device
        self.rank, self.world = dist.get_rank(), dist.get_world_size()

    def _ctl_device(self):
        """Where small control tensors live: what the backend can move."""
        return self.device

    def all_gather_ints(self, values):
        """Every rank's list of ints (setup only)."""
        if self.world == 1:
            return [list(values)]
        t = self.torch.tensor(list(values), dtype=self.torch.int64, device=self._ctl_device())
        out = [self.torch.zeros_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return [[int(x) for x in o.cpu()] for o in out]

    def all_agree(self, flag: bool) -> bool:
        """True only if `flag` is true on every rank (setup only)."""
        return all(v[0] for v in self.all_gather_ints([1 if flag else 0]))

    def all_reduce_sum(self, t):
        if self.world > 1:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)

    def all_gather_into(self, out, inp):
        if self.world > 1:
            self.dist.all_gather_into_tensor(out, inp)
        else:
            out.copy_(inp)

    def neighbour_exchange(self, send_lo, recv_lo, send_hi, recv_hi):
        """send_lo goes to rank-1, send_hi to rank+1; recv_lo comes from rank-1, recv_hi from rank+1.
        Any of them may be None."""
        self.neighbour_exchange_finish(self.neighbour_exchange_start(send_lo, recv_lo, send_hi, recv_hi))

    def neighbour_exchange_finish(self, works):
        """Orders the current stream behind the transfers (the host does not wait)."""
        for r in works:
            r.wait()

    def neighbour_exchange_start(self, send_lo, recv_lo, send_hi, recv_hi):
        """Starts the transfers behind what the current stream holds so far and returns at once;
        kernels enqueued before neighbour_exchange_finish run beside them."""
        ops = []
        d = self.dist
        if send_lo is not None:
            ops.append(d.P2POp(d.isend, send_lo, self.rank - 1))
        if recv_hi is not None:
            ops.append(d.P2POp(d.irecv, recv_hi, self.rank + 1))
        if send_hi is not None:
            ops.append(d.P2POp(d.isend, send_hi, self.rank + 1))
        if recv_lo is not None:
            ops.append(d.P2POp(d.irecv, recv_lo, self.rank - 1))
        return d.batch_isend_irecv(ops) if ops else []


class StagedComm(Comm):
    """Same interface, but every payload goes through host memory and the gloo
    backend.  Only for rehearsing several ranks on ONE GPU (RCCL refuses two
    ranks on one device); never used by bench.py."""

    def _stage(self, t):
        return t.detach().to("cpu")

    def _ctl_device(self):
        return "cpu"

    def all_reduce_sum(self, t):
        if self.world > 1:
            c = self._stage(t)
            self.dist.all_reduce(c, op=self.dist.ReduceOp.SUM)
            t.copy_(c)

    def all_gather_into(self, out, inp):
        if self.world > 1:
            c_out = self.torch.empty(out.shape, dtype=out.dtype)
            self.dist.all_gather_into_tensor(c_out, self._stage(inp).contiguous())
            out.copy_(c_out)
        else:
            out.copy_(inp)

    def neighbour_exchange_start(self, send_lo, recv_lo, send_hi, recv_hi):
        # host-staged: complete on return
        d = self.dist
        bufs = {}
        reqs = []
        if send_lo is not None:
            reqs.append(d.isend(self._stage(send_lo).contiguous(), self.rank - 1))
        if send_hi is not None:
            reqs.append(d.isend(self._stage(send_hi).contiguous(), self.rank + 1))
        if recv_hi is not None:
            bufs["hi"] = self.torch.empty(recv_hi.shape, dtype=recv_hi.dtype)
            reqs.append(d.irecv(bufs["hi"], self.rank + 1))
        if recv_lo is not None:
            bufs["lo"] = self.torch.empty(recv_lo.shape, dtype=recv_lo.dtype)
            reqs.append(d.irecv(bufs["lo"], self.rank - 1))
        for r in reqs:
            r.wait()
        if recv_hi is not None:
            recv_hi.copy_(bufs["hi"])
        if recv_lo is not None:
            recv_lo.copy_(bufs["lo"])
        return []


class DistLanczos:
    """One rank's share of a row-partitioned Lanczos run.

    `make_engine(plan, x_ext, max_iter)` returns an object with the phase
    interface of lanczos_amd.Engine (phase_spmv, phase_recur, phase_dots,
    phase_update, phase_scale, export_vector, set_start_scaled, history) whose
    phase methods take a torch tensor `red` living where the engine computes.
    """

    def __init__(self, comm: Comm, torch, device, plan: SlabPlan, bounds, max_iter: int, make_engine,
                 reorth: str = "full", interior=None):
        if reorth == "mgs":
            raise ValueError("row-partitioned runs take the blocked orders (full, full2, cuda_compat, selective, none): "
                             "one basis vector at a time would cost one reduction over ranks per vector")
        self.comm, self.torch, self.device = comm, torch, device
        self.plan, self.bounds, self.max_iter, self.reorth = plan, bounds, max_iter, reorth
        self.rank, self.world = comm.rank, comm.world
        self.red = torch.zeros(max_iter + 8, dtype=torch.float64, device=device)
        # in-place halo form: the basis lives here, in a tensor the collectives
        # can address, and the remote entries are received straight into the tail of the vector the
        # SpMV reads -- no gather buffer, no copy of the vector per iteration
        self.in_place = plan.mode == "halo" and plan.in_place
        if self.in_place:
            self.nvec = 3 if reorth == "none" else max_iter
            self.ld = -(-plan.n_ext // 2048) * 2048
            self.basis = torch.zeros(self.nvec, self.ld, dtype=torch.float64, device=device)
            self.x_ext = None
            self.engine = make_engine(plan, self.basis, max_iter)
        else:
            self.x_ext = torch.zeros(plan.n_ext, dtype=torch.float64, device=device)
            self.engine = make_engine(plan, self.x_ext, max_iter)
        # rows [interior[0], interior[1]) read nothing remote: their product can run beside the exchange
        self.interior = interior if (interior and self.in_place and getattr(self.engine, "can_overlap", False)) else None
        self.k = 0
        self.reorth_steps = 0
        self._views = {}
        self.red1, self.red2 = self.red[:1], self.red[:2]
        # no reorthogonalisation: unnormalised basis, one reduction of two doubles per iteration
        self.lazy = reorth == "none" and hasattr(self.engine, "phase_recur_lazy")
        self._closed_k = 0
        if reorth == "selective":
            # every rank holds the same alpha and beta, so every rank's tracker decides alike
            from . import OmegaTracker
            self.tracker = OmegaTracker(max_iter)
            self.ab = torch.zeros(2, dtype=torch.float64, device=device)
            self.last_beta = 0.0
        if plan.mode == "halo":
            # what the neighbours need from this slab
            need = comm.all_gather_ints([plan.halo_lo, plan.halo_hi])
            self.send_lo = need[self.rank - 1][1] if self.rank > 0 else 0               # head, for rank-1
            self.send_hi = need[self.rank + 1][0] if self.rank < self.world - 1 else 0  # tail, for rank+1
            assert self.send_lo <= plan.n_local and self.send_hi <= plan.n_local
        else:
            self.pad = plan.n_ext // self.world

    # -- start vector ---------------------------------------------------------
    def set_start(self, v0_local: np.ndarray):
        """v0_local: this slab's entries of the global start vector; normalised globally."""
        t = self.torch.tensor([float(np.dot(v0_local, v0_local))], dtype=self.torch.float64, device=self.device)
        self.comm.all_reduce_sum(t)
        norm2 = float(t.cpu()[0])
        self.engine.set_start_scaled(v0_local, 1.0 / np.sqrt(norm2))
        self.k = 0
        self.reorth_steps = 0
        self._closed_k = 0
        if self.reorth == "selective":
            self.tracker.reset()

    # -- the vector exchange --------------------------------------------------
    def exchange(self, start_only=False):
        p, n = self.plan, self.plan.n_local
        if p.mode == "allgather":
            mine = self.x_ext[self.rank * self.pad: self.rank * self.pad + n]
            self.engine.export_vector(self.k, mine)
            if self.world > 1:
                self.comm.all_gather_into(self.x_ext, self.x_ext[self.rank * self.pad:(self.rank + 1) * self.pad])
            return
        if self.in_place:
            x, base = self.basis[self.k % 3 if self.nvec == 3 else self.k], p.remote_base
        else:
            x, base = self.x_ext, n
            self.engine.export_vector(self.k, x[:n])
        if self.world == 1:
            return []
        # the four views of a basis slot never change: built once per slot
        slot = (self.k % 3 if self.nvec == 3 else self.k) if self.in_place else -1
        views = self._views.get(slot)
        if views is None:
            views = (x[:self.send_lo] if self.send_lo else None,
                     x[base:base + p.halo_lo] if p.halo_lo else None,
                     x[n - self.send_hi:n] if self.send_hi else None,
                     x[base + p.halo_lo:base + p.halo_lo + p.halo_hi] if p.halo_hi else None)
            self._views[slot] = views
        send_lo, recv_lo, send_hi, recv_hi = views
        works = self.comm.neighbour_exchange_start(send_lo, recv_lo, send_hi, recv_hi)
        if start_only:
            return works
        self.comm.neighbour_exchange_finish(works)
        return []

    # -- one iteration (reference src/lanczos.c:109-190, split at the reductions) --
    def step(self):
        e, red = self.engine, self.red
        if self.interior:
            works = self.exchange(start_only=True)        # the transfers run beside the product
            e.phase_spmv_begin(*self.interior)
            self.comm.neighbour_exchange_finish(works)
            e.phase_spmv_end(red)
        else:
            self.exchange()
            e.phase_spmv(red)
        if self.lazy:
            self.comm.all_reduce_sum(self.red2)           # u_k . A u_k and |u_k|^2 together
            e.phase_recur_lazy(red)
            self.k += 1
            return
        self.comm.all_reduce_sum(self.red1)               # alpha
        if self.reorth == "selective":
            self.ab[0:1].copy_(red[:1])
        n_red, is_norm = e.phase_recur(red)
        self.comm.all_reduce_sum(red[:n_red])             # reorth coefficients and w.w together, or beta^2
        if self.reorth == "selective":
            # the one host round trip per step: alpha_k and ||w||^2 for the decision rule
            self.ab[1:2].copy_(red[:1])
            alpha, norm2 = self.ab.cpu().tolist()
            if self.k > 0:
                self.tracker.set_beta(self.k - 1, self.last_beta)
            need, _ = self.tracker.step(self.k, alpha, float(np.sqrt(norm2)))
            if need:
                self.reorth_steps += 1
                n2 = e.phase_dots(red)
                self.comm.all_reduce_sum(red[:n2])
                e.phase_update(red)
                self.comm.all_reduce_sum(red[:1])         # beta^2
                norm2 = float(red[:1].cpu()[0])
            self.last_beta = float(np.sqrt(norm2))
        elif not is_norm:
            e.phase_update(red)                           # red[0] = beta^2 over all ranks (from the reduced w.w)
            if self.reorth == "full2":
                n2 = e.phase_dots(red)
                self.comm.all_reduce_sum(red[:n2])
                e.phase_update(red)
                self.comm.all_reduce_sum(red[:1])         # beta^2 after the second pass
        e.phase_scale(red)
        self.k += 1

    def run(self, steps: int):
        for _ in range(steps):
            self.step()

    def history(self):
        if self.lazy and self.k > 0 and self._closed_k != self.k:
            # beta of the last iteration is the norm of the vector it left behind
            self.engine.phase_lazy_norm(self.red)
            self.comm.all_reduce_sum(self.red[:1])
            self.engine.phase_lazy_close(self.red)
            self._closed_k = self.k
        return self.engine.history(self.k)


# ---------------------------------------------------------------------------
# GPU glue
# ---------------------------------------------------------------------------
class _GpuPhaseEngine:
    """lanczos_amd.Engine with tensors in place of raw device addresses."""

    def __init__(self, la, a_local, plan: SlabPlan, x_ext, max_iter, reorth, n_global, stream, spmv="auto",
                 profile=False):
        self.la = la
        mat = la.CrsMatrix(a_local.values, plan.column_index, a_local.row_head_indexes)
        if x_ext.dim() == 2:   # the caller's basis, remote entries in the tails of its vectors
            self.e = la.Engine(mat, max_iter, reorth=reorth, spmv=spmv, profile=profile, stream=stream,
                               n_global=n_global, row_begin=plan.row_begin, n_ext=plan.n_ext,
                               basis=x_ext.data_ptr(), basis_ld=x_ext.shape[1])
        else:
            self.e = la.Engine(mat, max_iter, reorth=reorth, spmv=spmv, profile=profile, stream=stream,
                               n_global=n_global, row_begin=plan.row_begin, n_ext=plan.n_ext, x_ext=x_ext.data_ptr())

    def set_start_scaled(self, v0, scale):
        self.e.set_start_scaled(v0, scale)

    def export_vector(self, k, dst):
        self.e.export_vector(k, dst.data_ptr())

    def phase_spmv(self, red):
        self.e.phase_spmv(red.data_ptr())

    def phase_recur(self, red):
        return self.e.phase_recur(red.data_ptr())

    def phase_dots(self, red):
        return self.e.phase_dots(red.data_ptr())

    def phase_update(self, red):
        self.e.phase_update(red.data_ptr())

    def phase_scale(self, red):
        self.e.phase_scale(red.data_ptr())

    @property
    def can_overlap(self):
        return self.e.can_overlap

    def phase_spmv_begin(self, lo, hi):
        self.e.phase_spmv_begin(lo, hi)

    def phase_spmv_end(self, red):
        self.e.phase_spmv_end(red.data_ptr())

    def phase_recur_lazy(self, red):
        self.e.phase_recur_lazy(red.data_ptr())

    def phase_lazy_norm(self, red):
        self.e.phase_lazy_norm(red.data_ptr())

    def phase_lazy_close(self, red):
        self.e.phase_lazy_close(red.data_ptr())

    def history(self, count):
        return self.e.history(count)


def interior_rows(row_head: np.ndarray, plan: SlabPlan, max_boundary_share: float = 0.25):
    """(first, end) of the rows between those that read remote entries, when the latter are a
    prefix (reading the previous slab) and a suffix (reading the next one) of the slab and few
    enough to be worth computing twice; None otherwise."""
    if plan.mode != "halo" or not plan.in_place:
        return None
    n = plan.n_local
    remote = np.flatnonzero(plan.column_index >= plan.remote_base)
    if remote.size == 0:
        return None
    rows = np.searchsorted(row_head, remote, side="right") - 1
    from_lo = plan.column_index[remote] < plan.remote_base + plan.halo_lo
    first = int(rows[from_lo].max()) + 1 if from_lo.any() else 0
    end = int(rows[~from_lo].min()) if (~from_lo).any() else n
    if first > end or (first + n - end) > max_boundary_share * n:
        return None
    return first, end


def choose_mode(cols, bounds, rank, comm, want="auto"):
    """All ranks must use the same exchange form: halo only if every rank can."""
    if want in ("halo", "allgather"):
        return want
    hlo, hhi = needed_halo(cols, bounds[rank], bounds[rank + 1])
    n_local = bounds[rank + 1] - bounds[rank]
    ok = halo_feasible(hlo, hhi, bounds, rank) and (hlo + hhi) <= n_local // 2 + 1
    return "halo" if comm.all_agree(ok) else "allgather"


def build_gpu_rank(la, torch, comm, a_local, bounds, max_iter, reorth="full", mode="auto", spmv="auto",
                   profile=False, in_place=True, overlap=True):
    """a_local: this rank's slab as a CrsMatrix with GLOBAL column indices."""
    rank = comm.rank
    device = torch.device("cuda", torch.cuda.current_device())
    # switches for comparing the exchange forms on a real node (same value on every rank)
    in_place = in_place and os.environ.get("LANCZOS_DIST_NO_INPLACE", "0") != "1"
    overlap = overlap and os.environ.get("LANCZOS_DIST_NO_OVERLAP", "0") != "1"
    mode = choose_mode(a_local.column_index, bounds, rank, comm, mode)
    n_local = bounds[rank + 1] - bounds[rank]
    plan = plan_slab(a_local.column_index, bounds, rank, mode, la.gather_base(n_local) if in_place else -1)
    # One side stream carries the engine's kernels AND torch's collectives/copies, so
    # they are ordered without host syncs.  It must not be the legacy default stream:
    # its handle is 0, which the C ABI reads as "create a private stream".
    side = torch.cuda.Stream()
    torch.cuda.set_stream(side)
    stream = side.cuda_stream
    assert stream != 0
    n_global = bounds[-1]

    def make_engine(plan_, x_ext, max_iter_):
        return _GpuPhaseEngine(la, a_local, plan_, x_ext, max_iter_, reorth, n_global, stream, spmv, profile)

    # every rank takes the same decision, or the collectives of the two step forms would not pair up
    inner = interior_rows(a_local.row_head_indexes, plan) if overlap else None
    if not comm.all_agree(inner is not None or comm.world == 1):
        inner = None
    runner = DistLanczos(comm, torch, device, plan, bounds, max_iter, make_engine, reorth, inner)
    runner.stream = side  # keeps the stream alive as long as the engine launches on it
    return runner


def _timed_dist_run(torch, dist, runner, v0, steps, warm):
    """max over ranks of the wall time of `steps` iterations from the start vector; of the kernels only the product is
    bracketed by events (its time goes into the line)"""
    runner.set_start(v0)
    runner.run(warm)
    runner.engine.e.profile_select(("spmv",))
    runner.set_start(v0)
    runner.engine.e.profile_reset()
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    runner.run(steps)
    torch.cuda.synchronize()
    dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def bench_distributed(args, workloads, algorithmic_bytes, hbm_peak, keeper=None):
    """bench.py for --gpus N > 1 (launched by torch.distributed.run, one rank per GPU)."""
    import torch
    import torch.distributed as dist

    import lanczos_amd as la

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with "
                         f"python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus}")
    torch.cuda.set_device(local_rank)
    dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    world = dist.get_world_size()   # what the line reports as n_gpus: the ranks RCCL connected
    comm = Comm(dist, torch, torch.device("cuda", local_rank))
    d, N = workloads[args.workload]
    n = N ** d
    bounds = partition_rows(n, world)
    a_local = la.synth_laplacian_rows(d, N, bounds[rank], bounds[rank + 1])
    nnz_t = torch.tensor([float(a_local.length)], dtype=torch.float64, device="cuda")
    dist.all_reduce(nnz_t)
    nnz = int(nnz_t.item())
    K, W = args.steps, args.warmup
    runner = build_gpu_rank(la, torch, comm, a_local, bounds, max(K, W) + 1, reorth=args.reorth, spmv=args.spmv,
                            profile=True)
    v0 = la.gaussian_fill(123456789, bounds[rank], bounds[rank + 1] - bounds[rank])
    runner.set_start(v0)
    runner.run(W)
    # only the product and the kernels one of which dominates are bracketed by events in the timed run (two events around
    # every launch of every class cost a run of short kernels several per cent)
    timed_classes = ("spmv", "recur") if args.reorth in ("none", "selective") else ("spmv", "dots", "update")
    runner.engine.e.profile_select(timed_classes)
    runner.set_start(v0)
    runner.engine.e.profile_reset()
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    runner.run(K)
    torch.cuda.synchronize()
    dist.barrier()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    alpha, beta = runner.history()
    prof = runner.engine.e.profile()
    # achieved SpMV bandwidth of the timed run: every rank's kernel on its own slab (the bytes of the slab's format over
    # the kernel's HIP-event time), summed over the ranks
    mine = prof["spmv"]["bytes"] / (prof["spmv"]["ms"] * 1e-3) / 1e9 if prof["spmv"]["ms"] > 0 else 0.0
    bw = torch.tensor([mine], dtype=torch.float64, device="cuda")
    dist.all_reduce(bw)
    spmv_gbps_all = float(bw.item())
    out = None
    if rank == 0:
        bytes_ = algorithmic_bytes(n, nnz)
        ritz = la.ritz_values(alpha, beta, K, 4)
        name = max(timed_classes, key=lambda k_: prof[k_]["ms"])
        p = prof[name]
        achieved = p["bytes"] / (p["ms"] * 1e-3) / 1e9 if p["ms"] > 0 else 0.0
        out = {
            "metric": "Lanczos iters/sec",
            "value": K / dt,
            "unit": "iters/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": dt / K * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"3-D 7-point Laplacian {N}^3 (n={n}, nnz={nnz}), {K} Lanczos steps from a fixed "
                            f"N(0,1) start vector, reorth={args.reorth}, row-partitioned over {world} GPUs "
                            f"({runner.plan.mode} exchange + {1 if runner.lazy else 2} all-reduce(s) per step)",
                "reorth": args.reorth,
                "exchange": runner.plan.mode,
                "exchange_in_place": bool(runner.in_place),
                "exchange_overlapped_with_product": runner.interior is not None,
                "parallelism": f"row-partition x{world}",
            },
            "roofline": {
                "bound": "hbm", "kernel": name, "achieved": achieved, "peak": hbm_peak, "unit": "GB/s",
                "frac": achieved / hbm_peak, "traffic": None, "launches": p["launches"],
                "note": "rank 0's slab; per-GPU figure",
            },
            "spmv_hbm_GBps": spmv_gbps_all,
            "spmv_hbm_GBps_per_gpu": spmv_gbps_all / world,
            "spmv_hbm_kernel": runner.engine.e.spmv_name,
            "crs_charged_bytes_per_iter_no_reorth": bytes_["iter_charged"],
            "kernel_ms_total_rank0": {k_: prof[k_]["ms"] for k_ in timed_classes},
            "ritz_lowest4": [float(x) for x in ritz],
        }
    # The primary figure stands from here on: rank 0 hands the line to its keeper (a process of its own that
    # prints the last line it was given when this one ends, however it ends).  What follows are secondary
    # sections; should one of them hang on some rank (they have never run on more than one GPU), every rank
    # leaves after a deadline, so the run still reports its metric.
    import tempfile
    import threading
    state = {"done": False, "section": "primary"}
    ctl = dist.new_group(backend="gloo")     # host-side agreement between the ranks: no kernel spins on a GPU for it
    # one node by contract: a file every rank can see calls the secondary sections off for all of them
    abort_path = os.path.join(tempfile.gettempdir(), f"lanczos_bench_abort_{os.environ.get('MASTER_PORT', '0')}")
    if rank == 0 and os.path.exists(abort_path):
        os.remove(abort_path)
    dist.barrier(group=ctl)

    def publish():
        if rank == 0 and out is not None and keeper is not None:
            keeper.publish(out)

    publish()

    def _leave(reason):
        """Ends this rank from whichever thread notices that the secondary sections cannot go on.  Rank 0's
        keeper still prints the primary line (with the reason in it); the exit code says that the run did
        not finish."""
        if state["done"]:
            return
        if rank == 0 and out is not None:
            out["secondary_sections"] = reason
            publish()
        os._exit(3)

    deadline = float(os.environ.get("LANCZOS_BENCH_SECONDARY_DEADLINE_S", "420"))
    timer = threading.Timer(deadline, lambda: _leave(f"abandoned at the {deadline:.0f} s deadline in section "
                                                     f"{state['section']}"))
    timer.daemon = True
    timer.start()

    def _watch():
        while not state["done"]:
            if os.path.exists(abort_path):
                try:
                    why = open(abort_path).read()
                except OSError:
                    why = "called off by another rank"
                _leave(f"abandoned in section {state['section']}: {why}"[:400])
            time.sleep(0.5)

    watcher = threading.Thread(target=_watch, daemon=True)
    watcher.start()
    grace = float(os.environ.get("LANCZOS_BENCH_SECTION_GRACE_S", "30"))

    def section(name, fn):
        """A secondary section.  An error is recorded in the line instead of ending the run, and the ranks
        agree on it (a sum over the host-side group) so that they go on in step.  A rank that failed while
        the others are still inside one of the section's collectives would wait for that agreement for
        ever: it gives them `grace` seconds, then calls the remaining sections off for every rank."""
        state["section"] = name
        err = None
        try:
            fn()
        except Exception as exc:  # noqa: BLE001
            err = f"{type(exc).__name__}: {exc}"[:300]
        call_off = None
        if err is not None:
            def _call_off():
                with open(abort_path, "w") as f:
                    f.write(f"rank {rank} failed ({err}); the other ranks did not leave the section within {grace:.0f} s")
            call_off = threading.Timer(grace, _call_off)
            call_off.daemon = True
            call_off.start()
        failed = torch.tensor([1 if err is not None else 0], dtype=torch.int64)
        dist.all_reduce(failed, group=ctl)
        if call_off is not None:
            call_off.cancel()
        if int(failed.item()) and rank == 0:
            out.setdefault("section_errors", {})[name] = err or f"{int(failed.item())} of {world} ranks failed"
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        publish()

    steps2 = 100
    del runner
    torch.cuda.empty_cache()

    def sec_vector_step():
        # the SpMV + vector step on its own (no reorthogonalisation), same slabs
        r2 = build_gpu_rank(la, torch, comm, a_local, bounds, 4, reorth="none", spmv=args.spmv, profile=True)
        dt2 = _timed_dist_run(torch, dist, r2, v0, steps2, 5)
        r2.history()                       # drains the stream and collects the kernel timings
        p2 = r2.engine.e.profile()
        if rank == 0:
            b = algorithmic_bytes(n, nnz)
            # bytes moved: what the product's and the recurrence's launches on rank 0's slab were charged (the SpMV in
            # the matrix format in use plus the 32 n of the deferred-normalisation recurrence; 10 n + 26 n when the
            # slab steps as two plane walks), times the ranks
            fmt_iter = (p2["spmv"]["bytes"] + p2["recur"]["bytes"]) / steps2 * world
            out["spmv_vector_step"] = {
                "iters_per_s": steps2 / dt2,
                "ms_per_iter": dt2 / steps2 * 1e3,
                "spmv_kernel": r2.engine.e.spmv_name,
                "bytes_moved_per_iter": fmt_iter,
                "frac_of_hbm_peak": fmt_iter * steps2 / dt2 / 1e9 / (hbm_peak * world),
                "crs_charged_bytes_per_iter": b["iter_charged"],
                "crs_charged_bytes_over_time_GBps": b["iter_charged"] * steps2 / dt2 / 1e9,
                "spmv_avg_ms_rank0": p2["spmv"]["ms"] / steps2,
                "exchange": r2.plan.mode,
                "all_reduces_per_step": 1 if r2.lazy else 2,
                "exchange_overlapped_with_product": r2.interior is not None,
            }

    def sec_big_grid():
        # strong scaling on the large grid (SpMV + vector step): the matrix whose 300-vector basis
        # does not fit one GPU; every rank generates its own slab
        big = args.big_grid
        nb = big ** 3
        bb = partition_rows(nb, world)
        slab = la.synth_laplacian_rows(3, big, bb[rank], bb[rank + 1])
        nz = torch.tensor([float(slab.length)], dtype=torch.float64, device="cuda")
        dist.all_reduce(nz)
        nnz_b = int(nz.item())
        r4 = build_gpu_rank(la, torch, comm, slab, bb, 4, reorth="none", spmv=args.spmv, profile=True)
        v4 = la.gaussian_fill(123456789, bb[rank], bb[rank + 1] - bb[rank])
        dt4 = _timed_dist_run(torch, dist, r4, v4, steps2, 5)
        r4.history()
        p4 = r4.engine.e.profile()
        if rank == 0:
            b = algorithmic_bytes(nb, nnz_b)
            out["strong_scaling_laplacian"] = {
                "workload": f"3-D 7-point Laplacian {big}^3 (n={nb}, nnz={nnz_b}), SpMV + vector step, reorth=none, "
                            f"row-partitioned over {world} GPUs",
                "steps": steps2,
                "iters_per_s": steps2 / dt4,
                "ms_per_iter": dt4 / steps2 * 1e3,
                "spmv_kernel": r4.engine.e.spmv_name,
                "bytes_moved_per_iter": (p4["spmv"]["bytes"] + p4["recur"]["bytes"]) / steps2 * world,
                "frac_of_hbm_peak": (p4["spmv"]["bytes"] + p4["recur"]["bytes"]) / dt4 / 1e9 / hbm_peak,
                "crs_charged_bytes_per_iter": b["iter_charged"],
                "crs_charged_bytes_over_time_GBps": b["iter_charged"] * steps2 / dt4 / 1e9,
                "spmv_avg_ms_rank0": p4["spmv"]["ms"] / steps2,
                "exchange": r4.plan.mode,
                "exchange_overlapped_with_product": r4.interior is not None,
            }

    def sec_random():
        # irregular rows: random symmetric matrix, all-gather exchange.  Every rank builds its own slab (the
        # rows hold ~21 entries each with little spread, so equal row counts are equal entry counts to a
        # fraction of a per cent)
        nr = args.rand_rows
        rb = partition_rows(nr, world)
        mine = la.synth_random_symmetric_rows(nr, 10, 42, rb[rank], rb[rank + 1])
        nz3 = torch.tensor([float(mine.length)], dtype=torch.float64, device="cuda")
        dist.all_reduce(nz3)
        nnz_r = int(nz3.item())
        r3 = build_gpu_rank(la, torch, comm, mine, rb, 4, reorth="none", spmv=args.spmv, profile=True)
        v3 = la.gaussian_fill(123456789, rb[rank], rb[rank + 1] - rb[rank])
        dt3 = _timed_dist_run(torch, dist, r3, v3, steps2, 5)
        r3.history()
        p3 = r3.engine.e.profile()
        if rank == 0:
            b = algorithmic_bytes(nr, nnz_r)
            out["random_symmetric"] = {
                "workload": f"random symmetric n={nr}, nnz={nnz_r} ({nnz_r / nr:.1f} per row), seed 42, "
                            f"equal row slabs over {world} GPUs",
                "steps": steps2,
                "iters_per_s": steps2 / dt3,
                "ms_per_iter": dt3 / steps2 * 1e3,
                "algorithmic_bytes_per_iter": b["iter_moved"],
                "frac_of_hbm_peak": b["iter_moved"] * steps2 / dt3 / 1e9 / (hbm_peak * world),
                "spmv_avg_ms_rank0": p3["spmv"]["ms"] / steps2,
                "exchange": r3.plan.mode,
            }

    def sec_converge():
        # time to solution: selective reorthogonalisation until the lowest eigenvalue stands
        conv = args.converge_steps
        r5 = build_gpu_rank(la, torch, comm, a_local, bounds, conv + 1, reorth="selective", spmv=args.spmv)
        dt5 = _timed_dist_run(torch, dist, r5, v0, conv, 2)
        a5, b5 = r5.history()
        if rank == 0:
            exact = 3.0 * (2.0 - 2.0 * np.cos(np.pi / (N + 1)))
            lowest = float(la.ritz_values(a5, b5, conv, 1)[0])
            out["converged_run"] = {
                "reorth": "selective", "steps": conv, "seconds": dt5, "iters_per_s": conv / dt5,
                "steps_that_reorthogonalised": r5.reorth_steps,
                "lowest_ritz": lowest, "lowest_exact": exact, "rel_err": abs(lowest - exact) / exact,
            }

    def sec_in_process():
        # The other multi-GPU path on the same hardware: ONE process (rank 0) drives all the GPUs of the
        # node through the library's own RCCL calls (lanczos_hip_multi_*, what LANCZOS_HIP_NGPU gives the
        # C driver), while the other ranks wait on the host.  Same matrix, start vector and step counts as
        # the primary figure.
        gl = ctl                                     # a host-side barrier: no kernel spins on the waiting GPUs
        if rank == 0:
            try:
                full = la.synth_laplacian(d, N)
                res = {}
                runs = [(args.reorth, K, W)] + ([("none", steps2, 5)] if args.reorth != "none" else [])
                for mode, steps_m, warm in runs:
                    with la.MultiEngine(full, max(steps_m, warm) + 1, world, reorth=mode) as m:
                        m.set_start_seed(123456789)
                        m.run(warm)
                        m.set_start_seed(123456789)
                        t0 = time.perf_counter()
                        a_m, b_m = m.run(steps_m)     # returns when every slab's stream has drained
                        dt_m = time.perf_counter() - t0
                        info = m.info()
                    res[mode] = {"steps": steps_m, "iters_per_s": steps_m / dt_m, "ms_per_step": dt_m / steps_m * 1e3,
                                 "exchange": "halo" if info["halo"] else "allgather",
                                 "exchange_overlapped_with_product": bool(info["overlapped"])}
                    if mode == args.reorth:
                        r_m = la.ritz_values(a_m, b_m, steps_m, 4)
                        res[mode]["ritz_rel_diff_vs_primary"] = float(np.max(np.abs(r_m - ritz) / np.abs(ritz)))
                out["in_process_rccl"] = dict(res, gpus=world, note="one host process, ncclCommInitAll over the "
                                              "node's GPUs; the other ranks idle on a host barrier")
            except Exception as exc:  # noqa: BLE001
                out["in_process_rccl"] = {"error": f"{type(exc).__name__}: {exc}"[:300]}
            torch.cuda.set_device(local_rank)
        dist.barrier(group=gl)

    section("spmv_vector_step", sec_vector_step)
    if getattr(args, "big_grid", 0) > 0:
        section("strong_scaling_laplacian", sec_big_grid)
    if getattr(args, "rand_rows", 0) > 0:
        section("random_symmetric", sec_random)
    if getattr(args, "converge_steps", 0) > 0 and d == 3:
        section("converged_run", sec_converge)
    if not getattr(args, "no_in_process", False):
        section("in_process_rccl", sec_in_process)
    state["done"] = True
    timer.cancel()
    dist.barrier()
    dist.destroy_process_group()
    return out
