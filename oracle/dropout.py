"""numpy restatement of the product's dropout generator and a dense checker for masked / dropped attention (ORACLE - test
infrastructure; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this).

The reference rejects `attn_mask` and `dropout_p > 0` (flash_attention_ipu/_impl/scaled_dot_product_attention.py:21-28), so there
is no reference behaviour to restate: the yardstick is torch's own definition of scaled_dot_product_attention (softmax of
scale * q k^T + mask, dropout of the weights with the survivors scaled by 1 / (1 - p), times v) evaluated densely in fp64, with
the keep decisions taken by the SAME counter-based function the kernels use (csrc/fa_extras.h) - restated here independently
in numpy uint32 arithmetic:

    mix32(x):  x ^= x >> 16;  x *= 0x7feb352d;  x ^= x >> 15;  x *= 0x846ca68b;  x ^= x >> 16          ("lowbias32")
    key0 = the 64-bit seed and offset folded: mix32 chained over (seed_lo ^ 0x9E3779B9, seed_hi, offset_lo, offset_hi)
    kg   = mix32(key0 ^ g * 0x9E3779B1)
    kt   = mix32(kg + ((row >> 16) << 16 | col >> 16))
    keep(g, row, col) = mix32(kt ^ ((row & 0xffff) << 16 | (col & 0xffff))) >= floor(p * 2^32)
"""

import numpy as np
import torch

from .dense_attention import MASK_VALUE

_M32 = np.uint64(0xFFFFFFFF)


def _mix32(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.uint64) & _M32
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7FEB352D)) & _M32
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846CA68B)) & _M32
    x ^= x >> np.uint64(16)
    return x


def dropout_key0(seed: int, offset: int) -> int:
    k = _mix32(np.array([(seed & 0xFFFFFFFF) ^ 0x9E3779B9], dtype=np.uint64))
    k = _mix32(k ^ np.uint64((seed >> 32) & 0xFFFFFFFF))
    k = _mix32(k ^ np.uint64(offset & 0xFFFFFFFF))
    k = _mix32(k ^ np.uint64((offset >> 32) & 0xFFFFFFFF))
    return int(k[0])


def dropout_threshold(p: float) -> int:
    t = float(np.float32(p)) * 4294967296.0
    return 4294967295 if t >= 4294967295.0 else int(t)


def dropout_keep(seed: int, offset: int, groups, rows, cols, p: float) -> torch.Tensor:
    """keep[g, r, c] for group ids `groups`, row ids `rows`, column ids `cols` (1-d integer sequences) -> bool tensor."""
    g = np.asarray(groups, dtype=np.uint64)[:, None, None]
    r = np.asarray(rows, dtype=np.uint64)[None, :, None]
    c = np.asarray(cols, dtype=np.uint64)[None, None, :]
    kg = _mix32(np.uint64(dropout_key0(seed, offset)) ^ ((g * np.uint64(0x9E3779B1)) & _M32))
    kt = _mix32((kg + (((r >> np.uint64(16)) << np.uint64(16)) | (c >> np.uint64(16)))) & _M32)
    h = _mix32(kt ^ (((r & np.uint64(0xFFFF)) << np.uint64(16)) | (c & np.uint64(0xFFFF))))
    return torch.from_numpy(h >= np.uint64(dropout_threshold(p)))


def dense_attention_extra(q, k, v, scale=1.0, causal=False, attn_mask=None, dropout_p=0.0, seed=0, offset=0,
                          compute_dtype=torch.float64):
    """(out [B, H, L, D], lse [B, H, L]) of q [B, H, L, D] against k, v [B, H_kv, L, D] with torch's SDPA semantics:
    attn_mask boolean (True = the key takes part) or additive, broadcastable to [B, H, L, L]; the causal mask as in
    dense_attention_lse; dropout decisions from dropout_keep with group id b * H + h.  Rows without any admitted key give NaN."""
    q, k, v = q.to(compute_dtype), k.to(compute_dtype), v.to(compute_dtype)
    B, H, L, _ = q.shape
    Hkv = k.shape[1]
    if Hkv != H:
        k = k.repeat_interleave(H // Hkv, dim=1)
        v = v.repeat_interleave(H // Hkv, dim=1)
    s = (q @ k.transpose(-1, -2)) * scale
    if causal:
        s = s + torch.triu(torch.full((L, L), MASK_VALUE, dtype=compute_dtype), 1)
    if attn_mask is not None:
        if attn_mask.dtype == torch.bool:
            s = s.masked_fill(~attn_mask.expand(B, H, L, L), float("-inf"))
        else:
            s = s + attn_mask.to(compute_dtype)
    m = s.max(dim=-1, keepdim=True).values
    m_safe = torch.where(torch.isinf(m), torch.zeros_like(m), m)
    p = torch.exp(s - m_safe)
    l = p.sum(dim=-1, keepdim=True)
    w = p / l
    if dropout_p > 0.0:
        keep = dropout_keep(seed, offset, range(B * H), range(L), range(L), dropout_p).view(B, H, L, L)
        w = torch.where(keep, w / (1.0 - float(np.float32(dropout_p))), torch.zeros_like(w))
    return w @ v, (m_safe + torch.log(l)).squeeze(-1)
