"""Dense CPU restatement of the reference's attention forward (ORACLE - test infrastructure).

Follows, line for line in meaning (not in text):
  * flash_attention_ipu/_impl/flash_attention_qkv_packed.py:48-53  (CPU branch of the op:
    q,k,v = qkv; mask = triu(full((L,L), -10000), 1); softmax(q @ k^T + mask) @ v)
  * flash_attention_ipu/cpp/vanilla_attention.cpp:33-66             (same arithmetic on device:
    S = QK^T; S += triu(-10000, 1); rowmax; sub; exp; rowsum; div; @V)
  * flash_attention_ipu/cpp/flash_attention_qkv_packed.cpp:316      (logSumExp = m + log(l))
  * flash_attention_ipu/_impl/scaled_dot_product_attention.py:52-53 (chunk heuristic)

Semantics: NO 1/sqrt(D) scale inside the op; the causal mask is an additive -10000 (not -inf).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""

from math import ceil, floor, log2

import numpy as np
import torch

MASK_VALUE = -10000.0  # fa.py:49, vanilla_attention.cpp:53, FA.cpp:141


def dense_causal_attention(qkv: torch.Tensor, compute_dtype=None, slab: int = 0) -> torch.Tensor:
    """out[g] = softmax(q[g] @ k[g]^T + triu(-10000, 1)) @ v[g]   (fa.py:48-53).

    qkv: [3, G, L, D] CPU tensor.  `compute_dtype` (e.g. torch.float64 / torch.float32) upcasts the
    operands first - used to evaluate the oracle on the *same 16-bit-rounded inputs* the GPU sees.
    `slab` > 0 processes that many groups at a time to bound the [slab, L, L] score tensor.
    Returns [G, L, D] in compute_dtype (or qkv.dtype when compute_dtype is None).
    """
    assert qkv.ndim == 4 and qkv.shape[0] == 3
    if compute_dtype is not None:
        qkv = qkv.to(compute_dtype)
    q, k, v = qkv
    G, L, _ = q.shape
    mask = torch.triu(torch.full((L, L), MASK_VALUE, dtype=q.dtype), 1)
    if slab <= 0:
        slab = G
    outs = []
    for g0 in range(0, G, slab):
        s = q[g0:g0 + slab] @ k[g0:g0 + slab].permute(0, 2, 1) + mask
        p = torch.nn.functional.softmax(s, dim=-1)
        outs.append(p @ v[g0:g0 + slab])
    return torch.cat(outs, 0)


def dense_causal_attention_lse(qkv: torch.Tensor, scale: float = 1.0, compute_dtype=torch.float64):
    """(out, lse) with lse[g, r] = log sum_c exp(scale*S[r, c] + mask[r, c])   (FA.cpp:316).

    `scale` multiplies the logits (1.0 for the packed op; D**-0.5 models sdpa.py:55's fold).
    """
    qkv = qkv.to(compute_dtype)
    q, k, v = qkv
    L = q.shape[1]
    mask = torch.triu(torch.full((L, L), MASK_VALUE, dtype=q.dtype), 1)
    s = (q @ k.permute(0, 2, 1)) * scale + mask
    m = s.max(dim=-1, keepdim=True).values
    p = torch.exp(s - m)
    l = p.sum(dim=-1, keepdim=True)
    out = (p / l) @ v
    lse = (m + torch.log(l)).squeeze(-1)
    return out, lse


def sdpa_chunk_heuristic(L: int, D: int):
    """(num_chunks_q, num_chunks_kv) exactly as sdpa.py:52-53 chooses them."""
    r = max(L, D) // D
    return int(2 ** (ceil(log2(r) / 2))), int(2 ** (floor(log2(r) / 2)))


def frozen_randn(seed: int, shape) -> torch.Tensor:
    """Frozen-stream N(0,1) generator: identical on every box and torch version."""
    a = np.random.RandomState(seed).standard_normal(size=tuple(shape)).astype(np.float32)
    return torch.from_numpy(a)
