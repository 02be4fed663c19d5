"""GPU tests of the modes the reference REJECTS and this build serves (the guards at
flash_attention_ipu/_impl/scaled_dot_product_attention.py:29-37): is_causal=False and grouped-/multi-query
attention, forward and backward, plus the host-side edge cases around them (softmax scale <= 0, kv chunk -> kv step).
The checker is oracle.dense_attention_lse (the reference's dense arithmetic, fa.py:48-53, with the mask optional and
K/V heads expanded) evaluated in fp64 on the same 16-bit-rounded inputs; tolerances as in test_gpu_parity.py."""

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
SEED = 1123581321
MAXABS = {torch.float16: 1e-2, torch.bfloat16: 3e-2, torch.float32: 1e-4}
# logsumexp stash: the softmax denominator is the sum of the 16-bit-rounded P the PV product uses (formed on the
# matrix pipe), so a row dominated by one key carries that key's rounding: <= 2^-8 relative for bf16 (log -> 3.9e-3)
LSE_ATOL = 6e-3


def _check(out, ref, dtype, what):
    out = out.detach().cpu().double()
    err = (out - ref).abs()
    assert torch.isfinite(out).all(), f"{what}: non-finite output"
    if dtype != torch.float32:
        bad = err > (1e-2 + 1e-2 * ref.abs())
        assert not bool(bad.any()), f"{what}: {int(bad.sum())} elements out of tolerance, max|delta|={err.max().item():.3e}"
    assert err.max().item() <= MAXABS[dtype], f"{what}: max|delta|={err.max().item():.3e}"


def _fwd(q, k, v, scale, causal, lse=True):
    """fa_fwd through the C ABI on [B, H, L, D] q and [B, H_kv, L, D] k, v."""
    from flash_attention_amd._impl import _native

    B, H, L, D = q.shape
    out = torch.empty_like(q)
    lse_t = torch.empty(B * H, L, device=q.device, dtype=torch.float32) if lse else None
    st = lambda t: (t.stride(0), t.stride(1), t.stride(2))  # noqa: E731
    _native.fwd_strided(q, k, v, out, lse_t, B * H, L, D, H, st(q), st(k), st(v), st(out), 1, 1, scale, causal,
                        heads_kv=k.shape[1])
    return out, lse_t


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("B,H,L,D", [(1, 2, 32, 64), (2, 3, 200, 64), (1, 5, 257, 64), (2, 2, 1000, 128), (1, 3, 70, 24), (1, 1, 2304, 64)])
def test_noncausal_forward_vs_oracle(dtype, B, H, L, D):
    from oracle import dense_attention_lse, frozen_randn

    x = frozen_randn(SEED + L + D, (3, B, H, L, D)).to(dtype)
    scale = D ** -0.5
    ref, ref_lse = dense_attention_lse(x[0], x[1], x[2], scale=scale, causal=False)
    q, k, v = x.cuda()
    out, lse = _fwd(q, k, v, scale, False)
    _check(out, ref, dtype, f"non-causal {dtype} {B}x{H}x{L}x{D}")
    torch.testing.assert_close(lse.view(B, H, L).cpu().double(), ref_lse, atol=LSE_ATOL, rtol=1e-4)


@pytest.mark.parametrize("kernel,qb,waves", [("dma", 1, 4), ("dma", 2, 4), ("dma", 1, 8), ("dma", 1, 1),
                                             ("pipe", 2, 4), ("pipe", 1, 4), ("pipe", 1, 2), ("pipe16", 1, 4), ("pipe16", 2, 4), ("split2", 1, 4), ("split4", 1, 4)])
def test_noncausal_every_kernel_variant(knobs, kernel, qb, waves):
    """is_causal=False through every MFMA kernel (the mask mode is a template parameter of the pipelined kernels: no
    dead block in the last step, only keys past L masked), ragged L, unscaled bf16 logits."""
    from oracle import dense_attention_lse, frozen_randn

    if kernel.startswith("split"):      # the dma kernel with 2 | 4 kv split groups per workgroup
        knobs.set("split", int(kernel[5:]))
    else:
        knobs.set("kernel", kernel)
        knobs.set("qb", qb)
        knobs.set("waves", waves)
    B, H, L, D = 1, 3, 600, 64
    x = (frozen_randn(SEED + 5, (3, B, H, L, D)) * 0.6).to(torch.bfloat16)
    ref, ref_lse = dense_attention_lse(x[0], x[1], x[2], scale=1.0, causal=False)
    q, k, v = x.cuda()
    out, lse = _fwd(q, k, v, 1.0, False)
    _check(out, ref, torch.bfloat16, f"non-causal kernel={kernel} qb={qb} waves={waves}")
    torch.testing.assert_close(lse.view(B, H, L).cpu().double(), ref_lse, atol=LSE_ATOL, rtol=1e-4)


@pytest.mark.parametrize("kernel,waves", [("pipe", 4), ("pipe", 8), ("pipe16", 4), ("pipe16", 8), ("dma", 8)])
@pytest.mark.parametrize("L", [600, 4100])
def test_noncausal_head_dim_128_pipelined_kernels(knobs, kernel, waves, L):
    """Unmasked attention at head_dim 128 through the pipelined kernels (32x32x16 and 16x16x32 MFMA shapes)."""
    from oracle import dense_attention_lse, frozen_randn

    knobs.set("kernel", kernel)
    knobs.set("waves", waves)
    B, H, D = 1, 2, 128
    x = frozen_randn(SEED + 6 + L, (3, B, H, L, D)).to(torch.bfloat16)
    scale = D ** -0.5
    ref, ref_lse = dense_attention_lse(x[0], x[1], x[2], scale=scale, causal=False)
    q, k, v = x.cuda()
    out, lse = _fwd(q, k, v, scale, False)
    _check(out, ref, torch.bfloat16, f"non-causal D=128 kernel={kernel} waves={waves} L={L}")
    torch.testing.assert_close(lse.view(B, H, L).cpu().double(), ref_lse, atol=LSE_ATOL, rtol=1e-4)


@pytest.mark.parametrize("causal", [True, False])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("H,Hkv", [(8, 2), (6, 1), (4, 4)])
def test_grouped_query_forward_vs_oracle(dtype, causal, H, Hkv):
    """Query head h reads key/value head h // (H / H_kv) inside the kernel; K and V are never expanded."""
    from oracle import dense_attention_lse, frozen_randn

    B, L, D = 2, 300, 64
    q = frozen_randn(SEED + H, (B, H, L, D)).to(dtype)
    k = frozen_randn(SEED + H + 1, (B, Hkv, L, D)).to(dtype)
    v = frozen_randn(SEED + H + 2, (B, Hkv, L, D)).to(dtype)
    scale = D ** -0.5
    ref, ref_lse = dense_attention_lse(q, k, v, scale=scale, causal=causal)
    out, lse = _fwd(q.cuda(), k.cuda(), v.cuda(), scale, causal)
    _check(out, ref, dtype, f"gqa {dtype} H={H} Hkv={Hkv} causal={causal}")
    torch.testing.assert_close(lse.view(B, H, L).cpu().double(), ref_lse, atol=LSE_ATOL, rtol=1e-4)


def test_grouped_query_long_sequence_pipelined_kernel():
    """GQA through the default kernel of the headline regime (L = 4096, D = 64: pipelined, two query blocks)."""
    from flash_attention_amd._impl import _native
    from oracle import dense_attention_lse

    B, H, Hkv, L, D = 4, 16, 4, 4096, 64
    assert _native.plan(B * H, L, D, torch.bfloat16, 8, 8, softmax_scale=D ** -0.5).kernel in (2, 4)   # (pipe | pipe16: the measured table decides)
    gen = torch.Generator(device="cuda").manual_seed(3)
    q = torch.randn(B, H, L, D, device="cuda", generator=gen).to(torch.bfloat16)
    k = torch.randn(B, Hkv, L, D, device="cuda", generator=gen).to(torch.bfloat16)
    v = torch.randn(B, Hkv, L, D, device="cuda", generator=gen).to(torch.bfloat16)
    out = F.scaled_dot_product_attention(q, k, v, is_causal=True, enable_gqa=True)
    for b, h in ((0, 0), (3, 15), (2, 9)):
        ref, _ = dense_attention_lse(q[b:b + 1, h:h + 1].cpu(), k[b:b + 1, h // 4:h // 4 + 1].cpu(),
                                     v[b:b + 1, h // 4:h // 4 + 1].cpu(), scale=D ** -0.5, causal=True)
        _check(out[b:b + 1, h:h + 1], ref, torch.bfloat16, f"gqa L=4096 ({b},{h})")


def test_sdpa_override_serves_the_widened_modes():
    """On the GPU the override no longer raises for is_causal=False / grouped-query shapes (attn_mask and dropout:
    tests/test_gpu_extra.py); results agree with the fp64 oracle and with stock SDPA."""
    import flash_attention_amd.auto  # noqa: F401
    from oracle import dense_attention_lse, frozen_randn

    patched = F.scaled_dot_product_attention
    stock = patched.__wrapped__
    B, H, Hkv, L, D = 2, 6, 2, 384, 64
    q = frozen_randn(SEED + 40, (B, H, L, D)).to(torch.bfloat16).cuda()
    k = frozen_randn(SEED + 41, (B, Hkv, L, D)).to(torch.bfloat16).cuda()
    v = frozen_randn(SEED + 42, (B, Hkv, L, D)).to(torch.bfloat16).cuda()
    kf, vf = k.repeat_interleave(H // Hkv, 1), v.repeat_interleave(H // Hkv, 1)
    for causal in (True, False):
        ref, _ = dense_attention_lse(q.cpu(), k.cpu(), v.cpu(), scale=D ** -0.5, causal=causal)
        _check(patched(q, k, v, is_causal=causal, enable_gqa=True), ref, torch.bfloat16, f"override gqa causal={causal}")
        _check(patched(q, kf, vf, is_causal=causal), ref, torch.bfloat16, f"override mha causal={causal}")
        _check(stock(q, kf, vf, is_causal=causal), ref, torch.bfloat16, f"stock sdpa causal={causal}")   # same yardstick
    with pytest.raises(NotImplementedError, match="Grouped- or Multi-query"):   # grouped K/V heads are served only on request, as stock SDPA does
        patched(q, k, v, is_causal=True)
    with pytest.raises(NotImplementedError, match="Grouped- or Multi-query"):   # 6 query heads over 4 kv heads: not a grouping
        patched(q, k.repeat_interleave(2, 1), v.repeat_interleave(2, 1), is_causal=True)


def test_different_sequence_lengths_go_to_stock_sdpa_on_the_gpu_and_raise_off_it(caplog):
    """A decode step against a key/value cache (L_q = 1) and cross-attention (L_q != L_kv): the reference rejects every query.shape !=
    key.shape (scaled_dot_product_attention.py:33-37); off the GPU so does the override, on the GPU the call is stock SDPA's - importing
    the override must not break a generation loop.  The hand-off is logged, once per distinct reason (the reference logs its patch, utils.py:35)."""
    import logging

    import flash_attention_amd.auto  # noqa: F401
    from flash_attention_amd._impl import _native
    from flash_attention_amd._impl import scaled_dot_product_attention as ov

    ov._HANDOFF_REASONS.clear()
    caplog.set_level(logging.INFO, logger=ov.logger.name)

    patched = F.scaled_dot_product_attention
    stock = patched.__wrapped__
    gen = torch.Generator(device="cuda").manual_seed(5)
    k, v = (torch.randn(2, 4, 300, 64, device="cuda", generator=gen).bfloat16() for _ in range(2))
    calls = []
    real = _native.fwd_strided
    _native.fwd_strided = lambda *a, **kw: (calls.append(1), real(*a, **kw))[1]
    try:
        for Lq, causal in ((1, False), (7, False), (300 - 8, True)):
            q = torch.randn(2, 4, Lq, 64, device="cuda", generator=gen).bfloat16()
            assert torch.equal(patched(q, k, v, is_causal=causal), stock(q, k, v, is_causal=causal))
    finally:
        _native.fwd_strided = real
    assert calls == []
    said = [r.getMessage() for r in caplog.records if "handed to stock SDPA" in r.getMessage()]
    assert len(said) == 1 and "L_q != L_kv" in said[0], said          # three calls, one reason, one line
    with pytest.raises(NotImplementedError, match="Grouped- or Multi-query"):
        patched(torch.zeros(2, 4, 1, 64), k.cpu(), v.cpu(), is_causal=True)


@pytest.mark.parametrize("scale", [0.0, -0.125])
def test_sdpa_nonpositive_scale_goes_to_stock_sdpa(scale):
    """A softmax scale <= 0 is outside what the kernels compute (fixed reference max, exp2 of scale*log2e*s): the
    override hands it to stock SDPA (valid softmax), and the C ABI refuses it (FA_ERR_BAD_SCALE)."""
    import flash_attention_amd.auto  # noqa: F401
    from flash_attention_amd._impl import _native
    from oracle import frozen_randn

    patched = F.scaled_dot_product_attention
    q, k, v = frozen_randn(SEED + 50, (3, 2, 2, 128, 64)).to(torch.bfloat16).cuda()
    out = patched(q, k, v, is_causal=True, scale=scale)
    ref = patched.__wrapped__(q, k, v, is_causal=True, scale=scale)
    assert torch.isfinite(out).all() and torch.equal(out, ref)
    with pytest.raises(ValueError, match="softmax_scale must be finite and > 0"):
        _fwd(q, k, v, scale, True)


@pytest.mark.parametrize("causal", [True, False])
@pytest.mark.parametrize("dtype,D", [(torch.bfloat16, 64), (torch.float16, 128), (torch.float32, 64), (torch.bfloat16, 24)])
def test_backward_causal_and_noncausal_vs_oracle_autograd(monkeypatch, dtype, D, causal):
    """dQ, dK, dV through the SDPA override (MFMA backward kernels, generic kernels for fp32 / odd head dims) against
    autograd through the fp64 oracle, for both mask modes; grouped K/V heads (H = 4 over H_kv = 2) are resolved inside the
    forward and backward kernels (fa_fwd / fa_bwd `heads_kv`)."""
    import flash_attention_amd.auto  # noqa: F401
    from flash_attention_amd._impl import ops
    from oracle import dense_attention_lse, frozen_randn

    monkeypatch.setattr(ops, "GQA_NATIVE_MIN_WORKGROUPS", 0)      # (the kernels' own head mapping, whatever the grid size)
    B, H, Hkv, L = 2, 4, 2, 200
    q0 = (frozen_randn(SEED + 60, (B, H, L, D)) * 0.7).to(dtype)
    k0 = (frozen_randn(SEED + 61, (B, Hkv, L, D)) * 0.7).to(dtype)
    v0 = frozen_randn(SEED + 62, (B, Hkv, L, D)).to(dtype)
    g0 = frozen_randn(SEED + 63, (B, H, L, D)).to(dtype)
    qr, kr, vr = (t.double().requires_grad_(True) for t in (q0, k0, v0))
    ref, _ = dense_attention_lse(qr, kr, vr, scale=D ** -0.5, causal=causal)
    ref.backward(g0.double())
    q, k, v = (t.cuda().requires_grad_(True) for t in (q0, k0, v0))
    out = F.scaled_dot_product_attention(q, k, v, is_causal=causal, enable_gqa=True)
    out.backward(g0.cuda())
    tol = {torch.float16: 4e-3, torch.bfloat16: 1.5e-2, torch.float32: 1e-5}[dtype]
    for name, got, want in (("out", out, ref.detach()), ("dq", q.grad, qr.grad), ("dk", k.grad, kr.grad), ("dv", v.grad, vr.grad)):
        got = got.detach().cpu().double()
        err = (got - want).abs()
        floor = tol * max(1e-3, want.abs().max().item())
        assert torch.isfinite(got).all(), name
        assert bool((err <= floor + 2e-2 * want.abs()).all()), f"{name} {dtype} D={D} causal={causal}: max|d|={err.max().item():.3e}"


@pytest.mark.parametrize("causal", [True, False])
@pytest.mark.parametrize("dtype,D,tol", [(torch.bfloat16, 64, 1.5e-2), (torch.bfloat16, 128, 1.5e-2), (torch.float32, 128, 1e-5), (torch.float16, 40, 4e-3)])
@pytest.mark.parametrize("H,Hkv", [(8, 1), (6, 2), (8, 8)])
@pytest.mark.parametrize("native", [True, False])
def test_grouped_query_backward_inside_the_kernels(monkeypatch, native, dtype, D, tol, H, Hkv, causal):
    """Grouped- / multi-query training: dQ with the K/V head h // (H / H_kv), dK and dV of a shared head summed over its query
    heads inside one workgroup (MFMA kernels for 16-bit and fp32 at head_dim 64 / 128, generic kernels at head_dim 40) -
    against autograd through the fp64 oracle on the unexpanded tensors; head-inner (projection-style) q layout, ragged L
    spanning several query tiles."""
    import flash_attention_amd.auto  # noqa: F401
    from flash_attention_amd._impl import ops
    from oracle import dense_attention_lse, frozen_randn

    monkeypatch.setattr(ops, "GQA_NATIVE_MIN_WORKGROUPS", 0 if native else 1 << 30)   # both routes of attention_gqa
    B, L = 2, 320          # (divisible by the chunk heuristic, 2.5 q tiles of 128 rows)
    q0 = (frozen_randn(SEED + 80 + H, (B, L, H, D)) * 0.7).to(dtype).transpose(1, 2)      # [B, H, L, D] view of [B, L, H, D]
    k0 = (frozen_randn(SEED + 81 + H, (B, Hkv, L, D)) * 0.7).to(dtype)
    v0 = frozen_randn(SEED + 82 + H, (B, Hkv, L, D)).to(dtype)
    g0 = frozen_randn(SEED + 83 + H, (B, H, L, D)).to(dtype)
    qr, kr, vr = (t.double().requires_grad_(True) for t in (q0, k0, v0))
    ref, _ = dense_attention_lse(qr, kr, vr, scale=D ** -0.5, causal=causal)
    ref.backward(g0.double())
    q, k, v = (t.cuda().requires_grad_(True) for t in (q0, k0, v0))
    out = F.scaled_dot_product_attention(q, k, v, is_causal=causal, enable_gqa=True)
    out.backward(g0.cuda())
    assert k.grad.shape == k0.shape and v.grad.shape == v0.shape and q.grad.shape == q0.shape
    for name, got, want in (("out", out, ref.detach()), ("dq", q.grad, qr.grad), ("dk", k.grad, kr.grad), ("dv", v.grad, vr.grad)):
        got = got.detach().cpu().double()
        err = (got - want).abs()
        floor = tol * max(1e-3, want.abs().max().item())
        assert torch.isfinite(got).all(), name
        assert bool((err <= floor + 2e-2 * want.abs()).all()), f"{name} {dtype} D={D} H={H}/{Hkv} causal={causal}: max|d|={err.max().item():.3e}"


def test_grouped_query_training_under_torch_compile(monkeypatch):
    """The grouped-query training path as torch.compile sees it: custom ops flash_attention_amd::{fwd_gqa, bwd_gqa} with fake
    implementations and registered autograd, fullgraph."""
    from flash_attention_amd._impl import ops
    from oracle import frozen_randn

    B, H, Hkv, L, D = 1, 4, 2, 192, 64
    q0, k0, v0 = (frozen_randn(SEED + 90 + i, (B, h, L, D)).to(torch.bfloat16) for i, h in enumerate((H, Hkv, Hkv)))
    monkeypatch.setattr(ops, "GQA_NATIVE_MIN_WORKGROUPS", 0)

    def f(q, k, v):
        return ops.attention_gqa(q, k, v, D ** -0.5, 2, 2, True).float().square().sum()

    grads = []
    for fn in (f, torch.compile(f, fullgraph=True)):
        q, k, v = (t.cuda().requires_grad_(True) for t in (q0, k0, v0))
        fn(q, k, v).backward()
        grads.append((q.grad, k.grad, v.grad))
    for a, b in zip(*grads):
        assert torch.equal(a, b)


def test_kv_chunk_below_64_keys_runs_32_key_steps(knobs):
    """num_chunks_kv is honoured where it is a real tile size: a kv chunk of 32 keys selects the 32-key pipelined
    steps; the result agrees with the default plan within rounding (the reference's chunk-independence)."""
    import flash_attention_amd as fa
    from flash_attention_amd._impl import _native
    from oracle import dense_causal_attention_lse, frozen_randn

    G, L, D = 3, 1024, 64
    knobs.set("table", 0)       # (the built-in rules: 64-key tiles at this shape unless the kv chunk asks for less)
    assert _native.plan(G, L, D, torch.bfloat16, 4, 32).block_kv == 32 and _native.plan(G, L, D, torch.bfloat16, 4, 4).block_kv == 64
    qkv = frozen_randn(SEED + 70, (3, G, L, D)).to(torch.bfloat16)
    ref, _ = dense_causal_attention_lse(qkv)
    for nk in (4, 32):
        _check(fa.flash_attention_qkv_packed(qkv.cuda(), 4, nk), ref, torch.bfloat16, f"num_chunks_kv={nk}")
