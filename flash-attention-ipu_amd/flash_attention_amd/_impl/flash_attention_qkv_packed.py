"""`flash_attention_qkv_packed` - same name, signature, checks and messages as the reference op
(flash_attention_ipu/_impl/flash_attention_qkv_packed.py:7-55), served on MI355X by the HIP kernels
behind the C ABI in include/flash_attention_amd.h."""

import torch

from . import _native, ops

PAD_HEAD_DIM_MIN_ROWS = 4096   # query rows (groups x L) from which a head dim other than 64 / 128 is padded to one


def flash_attention_qkv_packed(qkv: torch.Tensor, num_chunks_q: int, num_chunks_kv: int) -> torch.Tensor:
    """
    Memory-efficient causally masked multi-head attention from a packed qkv tensor.

    Computes `softmax(Q @ K.T + causal_mask, dim=-1) @ V` (no 1/sqrt(D) scale, exactly as the
    reference op) without materialising the attention matrix.

    qkv -- shape (3, N, L, D)
    returns -- shape (N, L, D), same dtype and device

    num_chunks_q / num_chunks_kv keep the reference's meaning (L is cut into that many query / key
    chunks, FA.cpp:193-195) and its divisibility checks; the result does not depend on them.  On the
    GPU the chunk length is the upper bound of the hardware tile (see fa_plan()).

    Device dispatch mirrors fa.py:38-53: a GPU tensor runs the HIP kernel (there is no fallback: the
    native library must be loaded); a CPU tensor runs the reference's own dense CPU expression.
    """
    if qkv.ndim != 4:
        raise ValueError("flash_attention_qkv_packed expects qkv input to have 4 dimensions")
    if qkv.shape[0] != 3:
        raise ValueError("flash_attention_qkv_packed expects qkv input to have size 3 at dimension 0")
    if qkv.shape[2] % num_chunks_q != 0:
        raise ValueError(
            "flash_attention_qkv_packed expects qkv size at dimension 2 to be divisible by num_chunks_q"
        )
    if qkv.shape[2] % num_chunks_kv != 0:
        raise ValueError(
            "flash_attention_qkv_packed expects qkv size at dimension 2 to be divisible by num_chunks_kv"
        )

    if qkv.is_cuda:
        if qkv.dtype not in _native.FA_DTYPE:
            raise TypeError(f"flash_attention_qkv_packed: unsupported dtype {qkv.dtype} on GPU")
        # HIP forward; HIP backward through autograd (an opaque torch.library custom op under torch.compile)
        D = qkv.shape[3]
        sixteen = qkv.dtype in (torch.float16, torch.bfloat16)
        native = D in (64, 128) or (sixteen and D <= 256 and D % 8 == 0)    # (zero-extended inside the MFMA kernels: no copies)
        if not native and D < (256 if sixteen else 128) and qkv.shape[1] * qkv.shape[2] >= PAD_HEAD_DIM_MIN_ROWS and not torch.compiler.is_compiling():
            # a head dim the MFMA kernels do not take (not a multiple of 8, or fp32 off 64 / 128), on a call too large for the generic
            # (scalar fp32) kernel: zero columns change neither Q K^T nor the first D columns of P V (the op applies no 1 / sqrt(D)
            # scale, FA.cpp:254), so the call runs padded to a head dim they do take and the result is sliced; autograd sees
            # pad / slice.  (D > 256 - fp32: D > 128 - stays on the generic kernel.)
            Dp = (D + 7) // 8 * 8 if sixteen else (64 if D < 64 else 128)
            padded = torch.nn.functional.pad(qkv, (0, Dp - D))
            return ops.attention_qkv_packed(padded, int(num_chunks_q), int(num_chunks_kv))[..., :D].contiguous()
        return ops.attention_qkv_packed(qkv, int(num_chunks_q), int(num_chunks_kv))

    # CPU tensors: the op's CPU branch (fa.py:47-53) - dense attention with an additive -10000 mask.
    q, k, v = qkv
    mask = torch.triu(torch.full((q.shape[1], q.shape[1]), -10000), 1)
    attn = torch.nn.functional.softmax(q @ k.permute(0, 2, 1) + mask, dim=-1)
    return attn @ v


def flash_attention_qkv_packed_with_lse(qkv: torch.Tensor, num_chunks_q: int = 1, num_chunks_kv: int = 1):
    """GPU only: (out [N,L,D], logsumexp [N,L] fp32) - the reference's AttentionOutputWithStash
    (flash_attention_qkv_packed.hpp:10-13; logSumExp = m + log(l), FA.cpp:316)."""
    if not qkv.is_cuda:
        raise ValueError("flash_attention_qkv_packed_with_lse needs a GPU tensor")
    if qkv.ndim != 4 or qkv.shape[0] != 3:
        raise ValueError("flash_attention_qkv_packed expects qkv of shape (3, N, L, D)")
    if qkv.dtype not in _native.FA_DTYPE:
        raise TypeError(f"flash_attention_qkv_packed: unsupported dtype {qkv.dtype} on GPU")
    return ops.fwd_qkv_packed(qkv, int(num_chunks_q), int(num_chunks_kv))


__all__ = ["flash_attention_qkv_packed"]
