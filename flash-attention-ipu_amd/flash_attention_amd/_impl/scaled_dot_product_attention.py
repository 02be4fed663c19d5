"""Override of torch.nn.functional.scaled_dot_product_attention for causal self-attention on MI355X.

Same mechanism, guards, messages and chunk heuristic as the reference override
(flash_attention_ipu/_impl/scaled_dot_product_attention.py:11-61).  Differences, all on purpose:
  * "on device" means `query.is_cuda` (ROCm tensors) instead of poptorch.isRunningOnIpu();
  * q, k, v go to the kernel as three strided views and the D**-0.5 scale as a scalar, instead of
    `torch.stack([q * D**-0.5, k, v])` (sdpa.py:55), which would move 2x the QKV bytes;
  * `attn_mask is not None` replaces the reference's `if attn_mask:` (which raises RuntimeError for
    multi-element masks instead of NotImplementedError);
  * the modern `scale=` / `enable_gqa=` keywords are accepted so current callers do not break.
"""

from math import ceil, floor, log2
from typing import Callable, Optional

import torch

from . import _native
from .utils import patch_function


def chunk_heuristic(L: int, D: int):
    """(num_chunks_q, num_chunks_kv) exactly as sdpa.py:52-53: keep a score block no larger than q."""
    ratio = max(L, D) // D
    return int(2 ** (ceil(log2(ratio) / 2))), int(2 ** (floor(log2(ratio) / 2)))


def _group_view(x: torch.Tensor):
    """[*batch, L, D] -> (tensor, heads, (outer stride, inner stride, row stride)) without copying
    when the batch dims can be addressed as one or two strides."""
    if x.stride(-1) != 1 and x.shape[-1] != 1:
        x = x.contiguous()
    L, D = x.shape[-2:]
    batch = x.shape[:-2]
    try:
        flat = x.view(-1, L, D)            # raises if the batch dims do not collapse
        return flat, 1, (flat.stride(0), 0, flat.stride(1))
    except RuntimeError:
        pass
    if len(batch) == 2:
        return x, batch[1], (x.stride(0), x.stride(1), x.stride(2))
    flat = x.reshape(-1, L, D)             # copies
    return flat, 1, (flat.stride(0), 0, flat.stride(1))


class _SdpaAttention(torch.autograd.Function):
    """Training path of the override: forward with the logsumexp stash, backward through fa_bwd.
    Operands are made [G, L, D]-contiguous here (the no-copy strided path is the inference one)."""

    @staticmethod
    def forward(ctx, query, key, value, softmax_scale, num_chunks_q, num_chunks_kv):
        L, D = query.shape[-2:]
        q, k, v = (t.reshape(-1, L, D).contiguous() for t in (query, key, value))
        G = q.shape[0]
        out = torch.empty_like(q)
        lse = torch.empty((G, L), dtype=torch.float32, device=q.device)
        st = (L * D, 0, D)
        if out.numel() > 0:
            _native.fwd_strided(q, k, v, out, lse, G, L, D, 1, st, st, st, st, num_chunks_q, num_chunks_kv, softmax_scale)
        ctx.save_for_backward(q, k, v, out, lse)
        ctx.softmax_scale = softmax_scale
        ctx.batch_shape = query.shape[:-2]
        return out.view(*query.shape)

    @staticmethod
    def backward(ctx, dout):
        q, k, v, out, lse = ctx.saved_tensors
        G, L, D = q.shape
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        if dq.numel() > 0:
            _native.bwd(q, k, v, out, dout.reshape(G, L, D).contiguous(), lse, dq, dk, dv, G, L, D, 1, ctx.softmax_scale)
        shape = (*ctx.batch_shape, L, D)
        return dq.view(shape), dk.view(shape), dv.view(shape), None, None, None


@patch_function(torch.nn.functional.scaled_dot_product_attention, [torch.nn.functional])
def _scaled_dot_product_attention_amd(
    orig_fn: Callable,
    query: torch.Tensor,
    key: torch.Tensor,
    value: torch.Tensor,
    attn_mask: Optional[torch.Tensor] = None,
    dropout_p: float = 0.0,
    is_causal: bool = False,
    scale: Optional[float] = None,
    enable_gqa: bool = False,
) -> torch.Tensor:
    if attn_mask is not None:
        raise NotImplementedError("flash_attention_ipu does not currently support passing attn_mask")
    if dropout_p > 0.0:
        raise NotImplementedError("flash_attention_ipu does not currently support non-zero dropout_p")
    if not is_causal:
        raise NotImplementedError("flash_attention_ipu does not currently support is_causal=False")
    if query.shape != key.shape:
        raise NotImplementedError(
            "flash_attention_ipu does not currently support Grouped- or Multi-query \
             attention (query.shape != key.shape)"
        )
    if key.shape != value.shape:
        raise NotImplementedError("flash_attention_ipu does not currently support value.shape != key.shape")

    on_device = query.is_cuda and key.is_cuda and value.is_cuda and query.dtype in _native.FA_DTYPE \
        and query.dtype == key.dtype == value.dtype and query.ndim >= 2
    if on_device:
        L, D = query.shape[-2:]
        batch_shape = query.shape[:-2]
        num_chunks_q, num_chunks_kv = chunk_heuristic(L, D)
        softmax_scale = D ** -0.5 if scale is None else scale

        q, hq, qs = _group_view(query)
        k, hk, ks = _group_view(key)
        v, hv, vs = _group_view(value)
        heads = max(hq, hk, hv)
        # a 1-stride view is also a valid (batch, head) view when another operand needs two strides
        if heads > 1:
            def two_level(t, h, s):
                if h == heads:
                    return s
                return (s[0] * heads, s[0], s[2])
            qs, ks, vs = two_level(q, hq, qs), two_level(k, hk, ks), two_level(v, hv, vs)
        G = 1
        for n in batch_shape:
            G *= n
        if heads > 1 and query.ndim == 4 and query.stride(1) < query.stride(2):
            # head-inner input ([B, L, H, D] viewed as [B, H, L, D]): give the output the same layout,
            # so the caller's usual .transpose(1, 2).contiguous() is free
            B, H = batch_shape
            out = torch.empty((B, L, H, D), dtype=query.dtype, device=query.device).transpose(1, 2)
            os_ = (out.stride(0), out.stride(1), out.stride(2))
        else:
            out = torch.empty((*batch_shape, L, D), dtype=query.dtype, device=query.device)
            os_ = (L * D * heads, L * D, D) if heads > 1 else (L * D, 0, D)
        needs_grad = torch.is_grad_enabled() and (query.requires_grad or key.requires_grad or value.requires_grad)
        if needs_grad:
            return _SdpaAttention.apply(query, key, value, softmax_scale, num_chunks_q, num_chunks_kv)
        if out.numel() > 0:
            _native.fwd_strided(q, k, v, out, None, G, L, D, heads, qs, ks, vs, os_,
                                num_chunks_q, num_chunks_kv, softmax_scale)
        return out

    extra = {}
    if scale is not None:
        extra["scale"] = scale
    if enable_gqa:
        extra["enable_gqa"] = enable_gqa
    return orig_fn(query, key, value, attn_mask, dropout_p, is_causal, **extra)
