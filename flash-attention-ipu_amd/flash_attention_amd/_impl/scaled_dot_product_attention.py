"""Override of torch.nn.functional.scaled_dot_product_attention for causal self-attention on MI355X.

Same mechanism, guards, messages and chunk heuristic as the reference override
(flash_attention_ipu/_impl/scaled_dot_product_attention.py:11-61).  Differences, all on purpose:
  * "on device" means `query.is_cuda` (ROCm tensors) instead of poptorch.isRunningOnIpu();
  * q, k, v go to the kernel as three strided views and the D**-0.5 scale as a scalar, instead of
    `torch.stack([q * D**-0.5, k, v])` (sdpa.py:55), which would move 2x the QKV bytes; the kernels are
    reached through `torch.library` custom ops (ops.py), so autograd and torch.compile see opaque ops;
  * `attn_mask is not None` replaces the reference's `if attn_mask:` (which raises RuntimeError for
    multi-element masks instead of NotImplementedError);
  * the modern `scale=` / `enable_gqa=` keywords are accepted so current callers do not break;
  * on the GPU the four modes the reference rejects (sdpa.py:21-37) are served instead of raising: is_causal=False, grouped
    K/V heads (with enable_gqa=True, as stock SDPA demands), `attn_mask` (boolean or additive, anything broadcastable to
    [B, H, L, L]) and `dropout_p > 0` - the last two by the mask / dropout kernels (fa_fwd_ex / fa_bwd_ex) for f16 / bf16 at
    every head dim up to 256 that is a multiple of 8 (forward and backward; eager, under HIP-graph capture and under torch.compile), by
    stock SDPA (`orig_fn`) for what those kernels do not take.  Off the GPU they raise exactly like the reference.
"""

import logging
from math import ceil, floor, log2
from typing import Callable, Optional

import torch

from . import _native, ops
from .utils import patch_function


logger = logging.getLogger(__name__)
_HANDOFF_REASONS = set()


def _handoff(reason: str) -> None:
    """A GPU call leaves for stock SDPA (`orig_fn`): say why, ONCE per distinct reason (the reference logs its patch, utils.py:35, and raises
    where this override hands over - "why is my model slow" must be answerable from the log).  Silent while tracing: a log call breaks the graph."""
    if reason not in _HANDOFF_REASONS and not torch.compiler.is_compiling():
        _HANDOFF_REASONS.add(reason)
        logger.info("flash_attention_amd: scaled_dot_product_attention call handed to stock SDPA: %s", reason)


def chunk_heuristic(L: int, D: int):
    """(num_chunks_q, num_chunks_kv) exactly as sdpa.py:52-53: keep a score block no larger than q."""
    ratio = max(L, D) // D
    return int(2 ** (ceil(log2(ratio) / 2))), int(2 ** (floor(log2(ratio) / 2)))


# Head dims other than 64 / 128.  The reference's op has no head-dim restriction (FA.cpp:175-195; its own tests run D = 2,
# t_cpp:245-251).  The C ABI serves
#   * f16 / bf16 with D a multiple of 8 up to 256 through the MFMA kernels themselves: 64, 128, 192 and 256 natively, the others ZERO-EXTENDED
#     at the next of the four inside the kernels (round 4, `FwdParams::Dr`: out-of-range LDS-DMA chunks and predicated register loads
#     read the missing columns as zeros, stores stop at D - no copies, forward and backward; masks and dropout included);
#   * everything else (D > 256, D not a multiple of 8, fp32 off 64 / 128) through the generic kernel: one workgroup per query row, scalar
#     fp32 arithmetic, 1-2 TFLOP/s - a correctness path that an `import flash_attention_amd.auto` must not silently put
#     under a model.  For calls of at least GENERIC_KERNEL_MAX_ROWS query rows the override therefore pads such a D to the next head dim the
#     MFMA kernels take (zero columns change neither Q K^T nor the first D columns of P V; the scale stays D^-1/2 of the TRUE head dim;
#     autograd sees pad / slice) and hands what cannot be padded to one (D > 256; fp32 above 128) to stock SDPA (`orig_fn`).
# Smaller calls (the reference's own tiny test shapes) stay on the generic kernel: exact fp32 arithmetic, no copies.
GENERIC_KERNEL_MAX_ROWS = 4096
MFMA_HEAD_DIM_MAX = 256       # 16-bit operands (fa_capi.cpp: kernel_head_dim); fp32: 64 and 128 only


def _mfma_head_dim(D: int, dtype):
    """(served natively?, head dim to pad to or None): what the MFMA kernels make of a D-wide call"""
    sixteen = dtype in (torch.float16, torch.bfloat16)
    if D in (64, 128) or (sixteen and D <= MFMA_HEAD_DIM_MAX and D % 8 == 0):
        return True, D
    if D > (MFMA_HEAD_DIM_MAX if sixteen else 128):
        return False, None
    if sixteen:
        return False, (D + 7) // 8 * 8            # the next multiple of 8: the kernels zero-extend from there
    return False, 64 if D < 64 else 128           # fp32: the fp32 MFMA kernels take 64 and 128 only


def _gqa_ok(query: torch.Tensor, key: torch.Tensor, value: torch.Tensor) -> bool:
    """[B, H, L, D] query with [B, H_kv, L, D] key / value, H a multiple of H_kv (grouped- / multi-query)."""
    return (query.ndim == 4 and key.ndim == 4 and key.shape == value.shape and query.shape[0] == key.shape[0]
            and query.shape[2:] == key.shape[2:] and key.shape[1] >= 1 and query.shape[1] % key.shape[1] == 0)


def _masked_or_dropped(query, key, value, attn_mask, dropout_p, is_causal, scale):
    """attn_mask and / or dropout through the mask / dropout kernels; None when this call is not one they take (the caller then
    hands it to stock SDPA): operands other than f16 / bf16, head dims above 256, more than two batch dimensions, a mask that
    wants a gradient or does not broadcast to [B, H, L, L], attn_mask together with is_causal (torch's own error), dropout traced
    before the device-resident generator state exists."""
    needs_grad = torch.is_grad_enabled() and (query.requires_grad or key.requires_grad or value.requires_grad)
    if not ops.supports_extra(query, needs_grad) or query.ndim not in (3, 4):
        return None
    if torch.compiler.is_compiling() and dropout_p > 0.0 and ops.graph_dropout_state(query.device) is None:
        # torch.compile: the kernels take (seed, offset) from the DEVICE-resident generator state (below), which must exist before the trace -
        # an H2D copy cannot be traced.  One eager dropout call on the device (a warm-up step) or flash_attention_amd.init_graph_dropout_state()
        # creates it; until then a traced dropout call is stock SDPA's (once it exists the dictionary guard fails and the call is re-traced).
        return None
    if attn_mask is not None and (is_causal or attn_mask.requires_grad or not attn_mask.is_cuda):
        return None
    if not (0.0 <= dropout_p < 1.0):
        return None
    q4, k4, v4 = (t.unsqueeze(1) if t.ndim == 3 else t for t in (query, key, value))
    B, H, L, D = q4.shape
    if q4.numel() == 0 or k4.numel() == 0:
        return None
    mask = None
    if attn_mask is not None:
        m = attn_mask
        if m.dtype != torch.bool:
            if not m.dtype.is_floating_point:
                return None
            m = m.to(query.dtype)
        if m.ndim > 4 or (query.ndim == 3 and m.ndim > 3):
            return None
        if m.ndim >= 1 and m.stride(-1) != 1 and m.shape[-1] > 1:
            m = m.contiguous()
        if query.ndim == 3 and m.ndim == 3:
            m = m.unsqueeze(1)
        try:
            m = m.expand(B, H, L, L)
        except RuntimeError:
            return None
        if m.stride(-1) != 1:      # (a mask broadcast along the keys: materialise that row once)
            m = m.contiguous()
        mask = m      # (boolean masks go down as they are: one byte per element; no dtype view - inductor cannot lower bool.view(uint8))
    softmax_scale = D ** -0.5 if scale is None else scale
    seed, offset, state = 0, 0, None
    if dropout_p > 0.0:
        if torch.compiler.is_compiling():
            # traced: the state tensor is a graph INPUT; snapshot + advance are one opaque op that mutates it (ops.advance_dropout_state) -
            # every call of the compiled graph draws fresh decisions, its backward re-reads the snapshot its forward saved
            state = ops.traced_dropout_state(query.device, B * H * L * L)
        elif torch.cuda.is_current_stream_capturing():     # graph capture: (seed, offset) from device memory, fresh on every replay
            state = ops.capture_dropout_state(query.device, B * H * L * L)
        else:
            seed, offset = ops.next_dropout_seed(query.device, B * H * L * L)
    out = ops.attention_extra(q4, k4, v4, mask, softmax_scale, bool(is_causal), float(dropout_p), seed, offset, state)
    return out.squeeze(1) if query.ndim == 3 else out


def _unpad(out: torch.Tensor, D_true: int) -> torch.Tensor:
    """the first D_true columns of a result computed at a padded head dim (contiguous, as stock SDPA's result is)"""
    return out if out.shape[-1] == D_true else out[..., :D_true].contiguous()


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
    on_device = query.is_cuda and key.is_cuda and value.is_cuda and query.dtype in _native.FA_DTYPE \
        and query.dtype == key.dtype == value.dtype and query.ndim >= 2
    # The reference's guards, same order and messages (sdpa.py:21-41).  Two of them are WIDENED on the GPU
    # (SURVEY 8f-4): is_causal=False and grouped-/multi-query shapes are served by the HIP kernels instead of
    # raising; off the GPU they raise exactly like the reference.
    if attn_mask is not None and not on_device:
        raise NotImplementedError("flash_attention_ipu does not currently support passing attn_mask")
    if dropout_p > 0.0 and not on_device:
        raise NotImplementedError("flash_attention_ipu does not currently support non-zero dropout_p")
    if not is_causal and not on_device:
        raise NotImplementedError("flash_attention_ipu does not currently support is_causal=False")
    # Different sequence lengths with the same batch / head layout (a decode step against a key/value cache, cross-attention): the reference
    # raises for every query.shape != key.shape (sdpa.py:33-37) and so does this override off the GPU; ON the GPU such a call is stock SDPA's
    # (no kernel here takes L_q != L_kv), so that `import flash_attention_amd.auto` does not break a model's generation loop.
    if (on_device and key.shape == value.shape and query.ndim == key.ndim and query.shape[:-2] == key.shape[:-2]
            and query.shape[-1] == key.shape[-1] and query.shape[-2] != key.shape[-2]):
        kw = {}
        if scale is not None:
            kw["scale"] = scale
        if enable_gqa:
            kw["enable_gqa"] = enable_gqa
        _handoff("query and key sequence lengths differ (decode step / cross-attention): no kernel here takes L_q != L_kv")
        return orig_fn(query, key, value, attn_mask, dropout_p, is_causal, **kw)
    gqa = query.shape != key.shape
    # (grouped K/V heads are served only when the caller asks for them, as stock SDPA does: without enable_gqa=True a
    # query.shape != key.shape call raises like the reference instead of silently computing something stock SDPA rejects)
    if gqa and not (on_device and enable_gqa and _gqa_ok(query, key, value)):
        raise NotImplementedError(
            "flash_attention_ipu does not currently support Grouped- or Multi-query \
             attention (query.shape != key.shape)"
        )
    if key.shape != value.shape:
        raise NotImplementedError("flash_attention_ipu does not currently support value.shape != key.shape")

    extra = {}
    if scale is not None:
        extra["scale"] = scale
    if enable_gqa:
        extra["enable_gqa"] = enable_gqa
    # the kernels assume a positive finite scale (exp2 of scale * log2(e) * s with a fixed reference max); anything
    # else is stock SDPA's business
    if on_device and scale is not None and not (scale > 0.0 and scale < float("inf")):
        on_device = False
        _handoff("scale is not a positive finite number")
    D_true = query.shape[-1]
    native_ok, Dp = _mfma_head_dim(D_true, query.dtype) if on_device else (True, None)
    if on_device and not native_ok and query.numel() > 0 and query.numel() // D_true >= GENERIC_KERNEL_MAX_ROWS:
        if Dp is None or torch.compiler.is_compiling():
            on_device = False       # (no MFMA kernel for this head dim: stock SDPA, not the scalar generic kernel)
            _handoff(f"head dim {D_true} ({query.dtype}) is beyond the MFMA kernels (16-bit: multiples of 8 up to {MFMA_HEAD_DIM_MAX}; fp32: 64, 128)")
        else:
            if scale is None:
                scale = D_true ** -0.5
            query, key, value = (torch.nn.functional.pad(t, (0, Dp - D_true)) for t in (query, key, value))
    if on_device and (attn_mask is not None or dropout_p > 0.0):
        out = _masked_or_dropped(query, key, value, attn_mask, dropout_p, is_causal, scale)
        if out is not None:
            return _unpad(out, D_true)
        on_device = False       # (a combination the mask / dropout kernels do not take: stock SDPA serves it)
        _handoff("attn_mask / dropout_p in a combination the mask kernels do not take (head dim above 256, fp32, attn_mask with is_causal, a mask "
                 "that wants a gradient or does not broadcast to [B, H, L, L], dropout traced before init_graph_dropout_state())")
        if query.shape[-1] != D_true:
            query, key, value = (t[..., :D_true] for t in (query, key, value))
    if on_device:
        L, D = query.shape[-2:]
        batch_shape = query.shape[:-2]
        num_chunks_q, num_chunks_kv = chunk_heuristic(L, D)
        softmax_scale = D ** -0.5 if scale is None else scale
        causal = bool(is_causal)

        needs_grad = torch.is_grad_enabled() and (query.requires_grad or key.requires_grad or value.requires_grad)
        if not needs_grad and query.ndim == 4 and query.numel() > 0:
            # inference on [B, H, L, D] views (e.g. slices of a fused [B, T, 3C] projection): no copies at all;
            # grouped-query K/V heads are resolved inside the kernel
            return _unpad(ops.attention_bhld(query, key, value, softmax_scale, num_chunks_q, num_chunks_kv, causal), D_true)
        if gqa:   # training with grouped K/V heads: resolved inside the forward AND the backward kernels (no expanded K/V)
            return _unpad(ops.attention_gqa(query, key, value, softmax_scale, num_chunks_q, num_chunks_kv, causal), D_true)
        if query.ndim == 4 and query.numel() > 0 and not torch.compiler.is_compiling():
            # eager training on the [B, H, L, D] views as they are: no reshape (for the head-inner slices of a fused projection that was three
            # copies), result and gradients in the operands' layout, cached stride blocks (ops._BhldAttentionFn)
            return _unpad(ops.attention_bhld_train(query, key, value, softmax_scale, num_chunks_q, num_chunks_kv, causal), D_true)
        out = ops.attention(query.reshape(-1, L, D), key.reshape(-1, L, D), value.reshape(-1, L, D),
                            softmax_scale, num_chunks_q, num_chunks_kv, causal)
        return _unpad(out.view(*batch_shape, L, D), D_true)

    return orig_fn(query, key, value, attn_mask, dropout_p, is_causal, **extra)
