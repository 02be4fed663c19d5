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


def dense_attention_lse(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, scale: float = 1.0, causal: bool = True,
                        compute_dtype=torch.float64):
    """Checker for the WIDENED modes (is_causal=False and grouped-/multi-query attention, which the
    reference rejects - scaled_dot_product_attention.py:29-37 - so there is no reference behaviour to restate):
    the same dense arithmetic as dense_causal_attention_lse (fa.py:48-53), with the mask optional and with
    q [B, H, L, D] against k, v [B, H_kv, L, D], query head h reading key/value head h // (H / H_kv)
    (the semantics of torch SDPA's enable_gqa).  Returns (out [B, H, L, D], lse [B, H, L])."""
    q, k, v = q.to(compute_dtype), k.to(compute_dtype), v.to(compute_dtype)
    H, Hkv, L = q.shape[1], k.shape[1], q.shape[2]
    assert H % Hkv == 0
    if Hkv != H:
        k = k.repeat_interleave(H // Hkv, dim=1)
        v = v.repeat_interleave(H // Hkv, dim=1)
    s = (q @ k.transpose(-1, -2)) * scale
    if causal:
        s = s + torch.triu(torch.full((L, L), MASK_VALUE, dtype=q.dtype), 1)
    m = s.max(dim=-1, keepdim=True).values
    p = torch.exp(s - m)
    l = p.sum(dim=-1, keepdim=True)
    return (p / l) @ v, (m + torch.log(l)).squeeze(-1)


def dense_causal_attention_rows(qkv: torch.Tensor, rows, scale: float = 1.0, compute_dtype=torch.float64):
    """Rows `rows` of dense_causal_attention_lse's output for EVERY group, without the [L, L] score matrix:
    out[g, i] = softmax(scale * q[g, rows[i]] @ k[g]^T + mask[rows[i]]) @ v[g]  (fa.py:48-53 restricted to a few
    query rows) - the oracle for full-size configs (L = 8192 .. 32768) that a dense evaluation cannot reach.
    qkv: [3, G, L, D] CPU tensor; returns (out [G, len(rows), D], lse [G, len(rows)])."""
    rows = torch.as_tensor(rows, dtype=torch.long)
    L = qkv.shape[2]
    q = qkv[0][:, rows].to(compute_dtype)
    outs, lses = [], []
    cols = torch.arange(L)[None, :]
    mask = torch.where(cols > rows[:, None], torch.tensor(MASK_VALUE, dtype=compute_dtype), torch.tensor(0.0, dtype=compute_dtype))
    for g in range(qkv.shape[1]):
        s = (q[g] @ qkv[1, g].to(compute_dtype).T) * scale + mask
        m = s.max(dim=-1, keepdim=True).values
        p = torch.exp(s - m)
        l = p.sum(dim=-1, keepdim=True)
        outs.append((p / l) @ qkv[2, g].to(compute_dtype))
        lses.append((m + torch.log(l)).squeeze(-1))
    return torch.stack(outs), torch.stack(lses)


def dense_causal_attention_grads_blocked(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, dout: torch.Tensor, scale: float = 1.0,
                                         slab: int = 2048):
    """Gradients of ONE group of the causal op - O = softmax(scale Q K^T + mask) V, fa.py:48-53 - evaluated in fp64 by query-row
    slabs, so that the reference test sizes a dense autograd cannot hold are reachable (t_fa.py:56-93 goes to L = 16384: a dense
    [L, L] fp64 matrix is 2 GB there, autograd keeps several).  The closed forms are those of the reference's own grad graph
    (flashAttentionQKVPackedGrad, FA.cpp:375-492): with P the softmax rows of a slab,  Dsum = rowsum(dO o O),
    dS = P o (dO V^T - Dsum),  dQ = scale dS K,  dK += scale dS^T Q,  dV += P^T dO.
    q, k, v, dout: [L, D] CPU tensors (any float dtype); returns (out, dq, dk, dv) in fp64.  tests/test_oracle.py pins it against
    autograd through dense_causal_attention at sizes both can run."""
    q, k, v, dout = (t.to(torch.float64) for t in (q, k, v, dout))
    L, D = q.shape
    out = torch.empty(L, D, dtype=torch.float64)
    dq = torch.empty(L, D, dtype=torch.float64)
    dk = torch.zeros(L, D, dtype=torch.float64)
    dv = torch.zeros(L, D, dtype=torch.float64)
    cols = torch.arange(L)[None, :]
    for r0 in range(0, L, slab):
        r1 = min(L, r0 + slab)
        n = r1                       # causal: keys 0 .. r1 - 1 are all a slab's rows can see
        s = (q[r0:r1] @ k[:n].T) * scale
        s += torch.where(cols[:, :n] > torch.arange(r0, r1)[:, None], torch.tensor(MASK_VALUE, dtype=torch.float64), torch.tensor(0.0, dtype=torch.float64))
        p = torch.softmax(s, dim=-1)
        del s
        o = p @ v[:n]
        out[r0:r1] = o
        dp = dout[r0:r1] @ v[:n].T
        dsum = (dout[r0:r1] * o).sum(dim=-1, keepdim=True)
        dv[:n] += p.T @ dout[r0:r1]
        ds = p * (dp - dsum)
        del p, dp
        dq[r0:r1] = (ds @ k[:n]) * scale
        dk[:n] += (ds.T @ q[r0:r1]) * scale
        del ds
    return out, dq, dk, dv


def sdpa_chunk_heuristic(L: int, D: int):
    """(num_chunks_q, num_chunks_kv) exactly as sdpa.py:52-53 chooses them."""
    r = max(L, D) // D
    return int(2 ** (ceil(log2(r) / 2))), int(2 ** (floor(log2(r) / 2)))


def frozen_randn(seed: int, shape) -> torch.Tensor:
    """Frozen-stream N(0,1) generator: identical on every box and torch version."""
    a = np.random.RandomState(seed).standard_normal(size=tuple(shape)).astype(np.float32)
    return torch.from_numpy(a)
