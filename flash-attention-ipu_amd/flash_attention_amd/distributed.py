"""Multi-GPU forward: shard the independent `batch*heads` groups across ranks, gather O only.

The reference has no distributed layer (it only *describes* head sharding, "no communication between
chips", demo notebook line 904); groups are a pure batch dimension of every op in the forward
(FA.cpp:186, grouped matmuls / per-row reductions), so rank r of W simply owns groups
[r*G/W, (r+1)*G/W) and runs the same kernel on them.  The single exchange step is the gather of the
output: one RCCL all-gather of [G/W, L, D] per rank over xGMI (`torch.distributed` backend "nccl" is
RCCL on ROCm; "gloo" on CPU tensors is used by the CPU tests).
"""

from typing import Optional, Tuple

import torch
import torch.distributed as dist

from ._impl.flash_attention_qkv_packed import flash_attention_qkv_packed


def shard_range(num_groups: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Groups [lo, hi) owned by `rank`.  G must divide evenly so the all-gather is one flat buffer."""
    if num_groups % world_size != 0:
        raise ValueError(f"batch*heads = {num_groups} must be divisible by the world size {world_size}")
    per = num_groups // world_size
    return rank * per, (rank + 1) * per


def gather_output(out_local: torch.Tensor, group=None, out: Optional[torch.Tensor] = None, async_op: bool = False):
    """all-gather of the per-rank outputs [G/W, L, D] -> [G, L, D] (rank-major = group order)."""
    world = dist.get_world_size(group)
    out_local = out_local.contiguous()
    if out is None:
        out = torch.empty((world * out_local.shape[0], *out_local.shape[1:]), dtype=out_local.dtype,
                          device=out_local.device)
    work = dist.all_gather_into_tensor(out, out_local, group=group, async_op=async_op)
    return (out, work) if async_op else out


def flash_attention_qkv_packed_sharded(qkv_local: torch.Tensor, num_chunks_q: int, num_chunks_kv: int,
                                       group=None, gather: bool = True) -> torch.Tensor:
    """Each rank passes ITS shard `qkv_local` [3, G/W, L, D]; returns the full [G, L, D] on every rank
    (or the local [G/W, L, D] with gather=False).  No collective touches Q, K or V."""
    out_local = flash_attention_qkv_packed(qkv_local, num_chunks_q, num_chunks_kv)
    if not gather or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return out_local
    return gather_output(out_local, group=group)
