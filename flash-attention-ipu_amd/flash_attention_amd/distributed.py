"""Multi-GPU forward: shard the independent `batch*heads` groups across ranks, gather O only.

The reference has no distributed layer (it only *describes* head sharding, "no communication between
chips", demo notebook line 904); groups are a pure batch dimension of every op in the forward
(FA.cpp:186, grouped matmuls / per-row reductions), so rank r of W simply owns G/W groups and runs the same
kernel on them.  The single exchange step is the gather of the output over xGMI (`torch.distributed` backend
"nccl" is RCCL on ROCm; "gloo" on CPU tensors is used by the CPU tests).  Three ways to move it:

  "allgather"  one RCCL all-gather of [G/W, L, D] per rank (every rank ends up with the full O);
  "direct"     the same result as W-1 grouped point-to-point sends and receives per rank: on MI355X every pair of
               GPUs has its own xGMI link, so each link carries exactly one shard per direction and no ring
               forwards anything (a ring all-gather is bound by ONE link: ~7x the bytes per link);
  "root"       gather to rank 0 only (what "a gather of the output" needs at minimum): 1/W of the total traffic,
               the same per-link load on rank 0's seven links.

`SlabGather` overlaps the exchange with compute inside one call: the local shard is cut into slabs and the gather
of slab s runs (on RCCL's stream) under the kernel of slab s + 1.
"""

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist

from ._impl import _native
from ._impl.flash_attention_qkv_packed import flash_attention_qkv_packed

GATHER_MODES = ("allgather", "direct", "root")
# the exchanges that leave the FULL [G, L, D] on every rank - the only ones an automatic selection may return: "root" moves 1/W of the
# traffic and completes rank 0's buffer alone, so it is not interchangeable with the other two (and would always win a timing contest)
FULL_GATHER_MODES = ("allgather", "direct")


def shard_range(num_groups: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Groups [lo, hi) owned by `rank`.  G must divide evenly so the all-gather is one flat buffer."""
    if num_groups % world_size != 0:
        raise ValueError(f"batch*heads = {num_groups} must be divisible by the world size {world_size}")
    per = num_groups // world_size
    return rank * per, (rank + 1) * per


class _Works:
    """wait() on several point-to-point requests (the handle `gather_output(..., async_op=True)` returns)."""

    def __init__(self, works):
        self.works = list(works)

    def wait(self):
        for w in self.works:
            w.wait()
        return True


def gather_output(out_local: torch.Tensor, group=None, out: Optional[torch.Tensor] = None, async_op: bool = False,
                  mode: str = "allgather"):
    """Gather of the per-rank outputs [G/W, L, D] -> [G, L, D] (rank-major = group order).

    mode "allgather" / "direct": every rank receives everything; "root": only rank 0's buffer is complete (the
    other ranks' buffers hold their own shard).  Returns `out`, or `(out, work)` with async_op=True."""
    if mode not in GATHER_MODES:
        raise ValueError(f"mode must be one of {GATHER_MODES}")
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    out_local = out_local.contiguous()
    n = out_local.shape[0]
    if out is None:
        out = torch.empty((world * n, *out_local.shape[1:]), dtype=out_local.dtype, device=out_local.device)
    if mode == "allgather":
        work = dist.all_gather_into_tensor(out, out_local, group=group, async_op=async_op)
        return (out, work) if async_op else out
    # point-to-point forms: my own slot is a local copy, every other slot is filled by its owner
    out[rank * n:(rank + 1) * n].copy_(out_local)
    ops: List[dist.P2POp] = []
    if mode == "direct":
        for step in range(1, world):   # rank r talks to r+step / r-step: every link busy in both directions at once
            dst, src = (rank + step) % world, (rank - step) % world
            ops.append(dist.P2POp(dist.isend, out_local, dist.get_global_rank(group, dst) if group else dst, group))
            ops.append(dist.P2POp(dist.irecv, out[src * n:(src + 1) * n], dist.get_global_rank(group, src) if group else src, group))
    elif rank == 0:
        for src in range(1, world):
            ops.append(dist.P2POp(dist.irecv, out[src * n:(src + 1) * n], dist.get_global_rank(group, src) if group else src, group))
    else:
        ops.append(dist.P2POp(dist.isend, out_local, dist.get_global_rank(group, 0) if group else 0, group))
    works = _Works(dist.batch_isend_irecv(ops) if ops else [])
    if async_op:
        return out, works
    works.wait()
    return out


def select_gather_mode(out_local: torch.Tensor, step=None, group=None, reps: int = 3, clock=None, modes=FULL_GATHER_MODES,
                       also_time=()):
    """Measure the ways to move O on THIS machine and agree on the fastest of `modes`: (mode, table).

    `modes` defaults to the two exchanges after which EVERY rank holds the full [G, L, D] (what `flash_attention_qkv_packed_sharded` and
    `SlabGather` promise); the returned mode is always one of them.  `also_time` (e.g. ("root",)) adds informational rows to the table
    that are timed the same way but never selected: "root" leaves W - 1 ranks with their own shard only and moves 1/W of the bytes, so
    ranking it against the full gathers would compare different results.  Passing modes=GATHER_MODES is the explicit opt-in to let
    "root" win (a caller that needs O on rank 0 only).

    xGMI gives every GPU pair a link of its own, so the direct point-to-point gather should beat a ring all-gather -
    but RCCL's all-gather is not necessarily a ring, and which one wins is a property of the node, so it is timed, not assumed.
    Per mode: the exchange alone (`reps` gathers of `out_local`, no kernel) and - when `step` is given, a callable that runs the
    forward and returns a fresh [G/W, L, D] - kernel + exchange per step with the gather of step s under the kernel of step s + 1
    (double-buffered, as bench.py and SlabGather overlap them).  Every rank runs the same sequence (each gather is a collective);
    the times are all-reduced to their MAXIMUM over ranks, so every rank sees the same table and takes the same minimum: ms_per_step
    when `step` is given, else exchange_only_ms; ties go to the earlier entry of `modes`.
    table[mode] = {"exchange_only_ms", "ms_per_step" (None without `step`)}.  `clock` (default time.perf_counter) is a test hook."""
    import time

    clock = clock or time.perf_counter
    world = dist.get_world_size(group)
    cuda = out_local.is_cuda
    sync = torch.cuda.synchronize if cuda else (lambda: None)
    bufs = [torch.empty((world * out_local.shape[0], *out_local.shape[1:]), dtype=out_local.dtype, device=out_local.device) for _ in range(2)]
    reps = max(1, int(reps))
    times = []
    modes = tuple(modes)
    timed = modes + tuple(m for m in also_time if m not in modes)
    for mode in timed:
        gather_output(out_local, group=group, out=bufs[0], mode=mode)            # warm-up (connections, first-use setup)
        sync()
        dist.barrier(group)
        t0 = clock()
        for _ in range(reps):
            gather_output(out_local, group=group, out=bufs[0], mode=mode)
        sync()
        x_ms = (clock() - t0) / reps * 1e3
        s_ms = 0.0
        if step is not None:
            dist.barrier(group)
            pend = [None, None]
            t0 = clock()
            for i in range(reps):
                b = i & 1
                if pend[b] is not None:
                    pend[b][1].wait()
                o_ = step()
                pend[b] = (o_, gather_output(o_, group=group, out=bufs[b], async_op=True, mode=mode)[1])   # (o_ stays alive until its gather completes)
            for pb in pend:
                if pb is not None:
                    pb[1].wait()
            sync()
            s_ms = (clock() - t0) / reps * 1e3
        times += [x_ms, s_ms]
    tm = torch.tensor(times, dtype=torch.float64, device=out_local.device)
    dist.all_reduce(tm, op=dist.ReduceOp.MAX, group=group)
    tm = tm.tolist()
    table = {m: {"exchange_only_ms": tm[2 * i], "ms_per_step": tm[2 * i + 1] if step is not None else None, "eligible": m in modes}
             for i, m in enumerate(timed)}
    key = "ms_per_step" if step is not None else "exchange_only_ms"
    best = min(modes, key=lambda m: (table[m][key], modes.index(m)))
    return best, table


def flash_attention_qkv_packed_sharded(qkv_local: torch.Tensor, num_chunks_q: int, num_chunks_kv: int,
                                       group=None, gather: bool = True, mode: str = "allgather") -> torch.Tensor:
    """Each rank passes ITS shard `qkv_local` [3, G/W, L, D]; returns the full [G, L, D] on every rank
    (or the local [G/W, L, D] with gather=False).  No collective touches Q, K or V.
    mode "root" is the one exception to "on every rank": only rank 0's result is complete, the other ranks' buffers hold their own
    shard at its place and uninitialised memory elsewhere."""
    out_local = flash_attention_qkv_packed(qkv_local, num_chunks_q, num_chunks_kv)
    if not gather or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return out_local
    return gather_output(out_local, group=group, mode=mode)


class SlabGather:
    """Forward of one rank's shard in `slabs` pieces with the gather of slab s overlapped with the kernel of
    slab s + 1 (the exchange of BASELINE config 5: 268 MB of O per rank against ~2 ms of compute).

    Ownership is slab-cyclic so that the gathered buffer IS the global group order: with Gs = G_local / slabs
    groups per slab, global group g belongs to rank (g // Gs) % W, slab g // (Gs * W); the gathered tensor is
    [slabs * W * Gs, L, D] = [G, L, D].  (Groups are independent - FA.cpp:186 - so any partition is valid.)
    Buffers are allocated once; `run` may be called repeatedly (a step of bench.py).

    Stream contract: the kernels run on the caller's current stream, the exchanges on the backend's.  `run()` by default makes
    the current stream wait for every exchange it started before it returns (`Work.wait()` on an RCCL work orders the CURRENT
    STREAM behind the collective; it does not block the host), so whatever the caller enqueues next on that stream sees the
    complete result.  `run(wait=False)` returns while the last exchanges are still in flight - for a caller that overlaps them
    with its own next kernels (bench.py) - and the result must then not be read before `wait()`.  The gathered result lives in one
    of TWO buffers used alternately: a tensor returned by `run` stays valid until the run after the next one starts."""

    def __init__(self, groups_local: int, L: int, D: int, dtype, device, slabs: int = 4, group=None,
                 mode: str = "allgather", gather: bool = True, gather_at_world1: bool = False):
        if groups_local % slabs != 0:
            raise ValueError(f"local groups {groups_local} must be divisible by slabs={slabs}")
        self.group, self.mode, self.slabs = group, mode, slabs
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # (gather_at_world1: run the exchange calls even when this rank is alone - a copy - so that a one-GPU box exercises the backend's path)
        self.gather = gather and (self.world > 1 or (gather_at_world1 and dist.is_initialized()))
        self.gs = groups_local // slabs
        self.local = torch.empty((slabs, self.gs, L, D), dtype=dtype, device=device)
        self.fulls = [torch.empty((slabs, self.world * self.gs, L, D), dtype=dtype, device=device) for _ in range(2)] if self.gather else None
        self.turn = 0
        self.pending = [None] * slabs

    def owned_groups(self, rank: int) -> torch.Tensor:
        """Global group ids of `rank`'s shard, in the order of its local [G_local] axis."""
        s = torch.arange(self.slabs)[:, None]
        i = torch.arange(self.gs)[None, :]
        return ((s * self.world + rank) * self.gs + i).reshape(-1)

    def run(self, qkv_local: torch.Tensor, num_chunks_q: int, num_chunks_kv: int, wait: bool = True) -> torch.Tensor:
        """qkv_local: contiguous [3, G_local, L, D] on this rank's GPU.  Returns [G, L, D] (every rank, or rank 0
        with mode "root"), or this rank's [G_local, L, D] when there is nothing to gather.  See the class docstring for
        the stream contract (`wait`)."""
        _, gl, L, D = qkv_local.shape
        full = self.fulls[self.turn] if self.gather else None
        self.turn ^= 1
        gs = self.gs
        q, k, v = qkv_local[0], qkv_local[1], qkv_local[2]
        st = (L * D, 0, D)
        for s in range(self.slabs):
            if self.pending[s] is not None:        # slab s of the previous call: its gather still reads local[s]
                self.pending[s].wait()
                self.pending[s] = None
            sl = slice(s * gs, (s + 1) * gs)
            if qkv_local.is_cuda:   # no copy: the slab's q, k, v are contiguous [Gs, L, D] pieces of the shard
                _native.fwd_strided(q[sl], k[sl], v[sl], self.local[s], None, gs, L, D, 1, st, st, st, st,
                                    num_chunks_q, num_chunks_kv, 1.0)
            else:                   # CPU tensors: the op's own CPU branch (fa.py:47-53), as in the packed op
                self.local[s].copy_(flash_attention_qkv_packed(qkv_local[:, sl], num_chunks_q, num_chunks_kv))
            if self.gather:
                _, self.pending[s] = gather_output(self.local[s], group=self.group, out=full[s], async_op=True,
                                                   mode=self.mode)
        if wait:
            self.wait()
        return full.view(-1, L, D) if self.gather else self.local.view(gl, L, D)

    def wait(self):
        for s in range(self.slabs):
            if self.pending[s] is not None:
                self.pending[s].wait()
                self.pending[s] = None
