"""Head dims other than 64 / 128 (round-3 verdict, item 4).  The reference's op has no head-dim restriction
(flash_attention_ipu/cpp/flash_attention_qkv_packed.cpp:175-195; its own tests run D = 2, tests/cpp/test_flash_attention.cpp:245-251)
and the C ABI serves every D <= 1024: f16 / bf16 at every multiple of 8 up to 256 through the MFMA kernels (64, 128, 192 and 256 natively, the
others ZERO-EXTENDED inside the kernels: out-of-range LDS-DMA chunks and predicated loads read the missing columns as zeros, stores stop
at the true head dim - forward and backward; masks and dropout up to 128), everything else through the generic kernel (one workgroup per
query row, scalar fp32 arithmetic, 1-2 TFLOP/s).  `import flash_attention_amd.auto` must not put that kernel under a model: large calls it
would get run padded to a head dim the MFMA kernels take or, where there is none (D > 256), go to stock SDPA."""

import time

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
SEED = 1123581321


def _time(fn, n=20):
    fn()
    torch.cuda.synchronize()
    best = float("inf")
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best


@pytest.mark.parametrize("D", [32, 80, 96, 160, 192, 256])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_override_at_other_head_dims_matches_the_oracle_and_is_not_slower_than_stock(D, dtype):
    """`F.scaled_dot_product_attention` on [8, 16, 2048, D] through the override: forward against the fp64 oracle, gradients against
    autograd through the oracle on sampled groups, and the forward's time against stock SDPA (`__wrapped__`): <= 1.2x."""
    import flash_attention_amd.auto  # noqa: F401
    from oracle import dense_attention_lse

    B, H, L = 8, 16, 2048
    gen = torch.Generator(device="cuda").manual_seed(SEED + D)
    q, k, v = (torch.randn(B, H, L, D, device="cuda", generator=gen).to(dtype).requires_grad_(True) for _ in range(3))
    out = F.scaled_dot_product_attention(q, k, v, is_causal=True)
    assert out.shape == (B, H, L, D) and out.dtype == dtype and out.is_contiguous()
    dout = torch.randn(B, H, L, D, device="cuda", generator=gen).to(dtype)
    gq, gk, gv = torch.autograd.grad(out, (q, k, v), dout)
    for b, h in ((0, 0), (B - 1, H - 1), (3, 7)):
        qd, kd, vd = (t[b, h][None, None].detach().cpu().double().requires_grad_(True) for t in (q, k, v))
        ref, _ = dense_attention_lse(qd, kd, vd, scale=D ** -0.5, causal=True)
        err = (out[b, h].detach().cpu().double() - ref[0, 0].detach()).abs()
        assert bool((err <= 1e-2 + 1e-2 * ref[0, 0].detach().abs()).all()), f"D={D} group ({b},{h}): max|d|={err.max().item():.3e}"
        rq, rk, rv = torch.autograd.grad(ref, (qd, kd, vd), dout[b, h][None, None].cpu().double())
        for got, want, name in ((gq, rq, "dq"), (gk, rk, "dk"), (gv, rv, "dv")):
            g = got[b, h].detach().cpu().double()
            tol = (1.5e-2 if dtype == torch.bfloat16 else 4e-3) * max(1e-3, want.abs().max().item())
            bad = (g - want[0, 0]).abs() > tol + 2e-2 * want[0, 0].abs()
            assert not bool(bad.any()), f"D={D} {name} group ({b},{h}): {int(bad.sum())} out of tolerance"
    with torch.no_grad():
        t_ours = _time(lambda: F.scaled_dot_product_attention(q, k, v, is_causal=True))
        t_stock = _time(lambda: F.scaled_dot_product_attention.__wrapped__(q, k, v, is_causal=True))
    print(f"[head dim {D} {dtype}] override {t_ours * 1e3:.1f} us, stock SDPA {t_stock * 1e3:.1f} us")
    assert t_ours <= 1.2 * t_stock, f"D={D}: override {t_ours:.3f} ms vs stock {t_stock:.3f} ms"


def test_head_dims_above_256_go_to_stock_sdpa_and_tiny_calls_stay_on_the_generic_kernel(monkeypatch):
    import flash_attention_amd.auto  # noqa: F401
    from flash_attention_amd._impl import _native
    from oracle import dense_attention_lse

    calls = []
    real = _native.fwd_strided
    monkeypatch.setattr(_native, "fwd_strided", lambda *a, **kw: (calls.append(a[7]), real(*a, **kw))[1])
    q, k, v = (torch.randn(2, 4, 1024, 320, device="cuda").bfloat16() for _ in range(3))
    out = F.scaled_dot_product_attention(q, k, v, is_causal=True)
    ref = F.scaled_dot_product_attention.__wrapped__(q, k, v, is_causal=True)
    assert calls == [] and torch.equal(out, ref), "a large D = 320 call must be stock SDPA's"
    # head_dim 256 is the MFMA kernels' (plan: the pipelined kernel, 4 waves) ...
    q, k, v = (torch.randn(2, 4, 1024, 256, device="cuda").bfloat16() for _ in range(3))
    out = F.scaled_dot_product_attention(q, k, v, is_causal=True)
    assert calls == [256] and _native.plan(8, 1024, 256, torch.bfloat16, 1, 1, softmax_scale=1 / 16).kernel == 2
    ref = F.scaled_dot_product_attention.__wrapped__(q.float(), k.float(), v.float(), is_causal=True)
    torch.testing.assert_close(out.float(), ref, atol=1e-2, rtol=1e-2)
    # ... and so is a masked D = 256 call, with or without a gradient wanted (fa_fwd_ex / fa_bwd_ex: not stock SDPA's bits)
    calls.clear()
    mask = torch.ones(1024, 1024, device="cuda", dtype=torch.bool).tril()
    stock = F.scaled_dot_product_attention.__wrapped__
    out = F.scaled_dot_product_attention(q, k, v, attn_mask=mask)
    assert calls == [] and not torch.equal(out, stock(q, k, v, attn_mask=mask))
    torch.testing.assert_close(out.float(), stock(q.float(), k.float(), v.float(), attn_mask=mask), atol=1e-2, rtol=1e-2)
    qg = q.clone().requires_grad_(True)
    outg = F.scaled_dot_product_attention(qg, k, v, attn_mask=mask)
    assert torch.equal(outg.detach(), out)
    outg.sum().backward()
    assert torch.isfinite(qg.grad.float()).all()
    # the reference's own tiny shape family (t_cpp:245-251: L = 6, D = 2): the generic kernel, exact fp32 arithmetic
    q, k, v = (torch.randn(4, 6, 2, device="cuda") for _ in range(3))
    out = F.scaled_dot_product_attention(q, k, v, is_causal=True)
    assert calls == [2]
    ref, _ = dense_attention_lse(q.cpu().double()[None], k.cpu().double()[None], v.cpu().double()[None], scale=2 ** -0.5, causal=True)
    torch.testing.assert_close(out.cpu().double(), ref[0], atol=1e-4, rtol=1e-4)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("D", [8, 24, 40, 72, 96, 120, 136, 192, 200, 256])
@pytest.mark.parametrize("B,H,Hkv,L,causal", [(2, 3, 3, 200, True), (1, 4, 2, 333, True), (2, 2, 2, 1024, False), (1, 2, 1, 1500, True)])
def test_zero_extended_head_dims_through_the_mfma_kernels(dtype, D, B, H, Hkv, L, causal):
    """fa_fwd / fa_bwd on operands whose rows are D elements wide, D a multiple of 8 up to 256 other than 64 / 128: the MFMA
    kernels run them at head_dim 64 / 128 / 256 with the missing columns read as zeros (`FwdParams::Dr`) - no padded copies.  Out, logsumexp
    and all three gradients against the fp64 oracle; ragged L, grouped K/V heads, causal and unmasked."""
    from flash_attention_amd._impl import _native, ops
    from oracle import dense_attention_lse, frozen_randn

    p = _native.plan(B * H, L, D, dtype, 1, 1, softmax_scale=D ** -0.5, causal=causal)
    assert p.kernel in (0, 2, 4, 6), "a 16-bit head dim that is a multiple of 8 up to 256 must be served by an MFMA kernel"
    q = (frozen_randn(SEED + L + D, (B, H, L, D)) * 0.8).to(dtype)
    k = (frozen_randn(SEED + L + D + 1, (B, Hkv, L, D)) * 0.8).to(dtype)
    v = (frozen_randn(SEED + L + D + 2, (B, Hkv, L, D)) * 0.8).to(dtype)
    g = frozen_randn(SEED + L + D + 3, (B, H, L, D)).to(dtype)
    scale = D ** -0.5
    qr, kr, vr = (t.double().requires_grad_(True) for t in (q, k, v))
    ref, ref_lse = dense_attention_lse(qr, kr, vr, scale=scale, causal=causal)
    ref.backward(g.double())
    qc, kc, vc = (t.cuda().requires_grad_(True) for t in (q, k, v))
    out = ops.attention_gqa(qc, kc, vc, scale, 1, 1, causal) if Hkv != H else \
        ops.attention(qc.reshape(B * H, L, D), kc.reshape(B * H, L, D), vc.reshape(B * H, L, D), scale, 1, 1, causal).view(B, H, L, D)
    err = (out.detach().cpu().double() - ref.detach()).abs()
    assert bool((err <= 1e-2 + 1e-2 * ref.detach().abs()).all()), f"D={D} {dtype}: max|d|={err.max().item():.3e}"
    out.backward(g.cuda())
    for got, want, name in ((qc.grad, qr.grad, "dq"), (kc.grad, kr.grad, "dk"), (vc.grad, vr.grad, "dv")):
        gg = got.detach().cpu().double()
        assert torch.isfinite(gg).all()
        tol = (1.5e-2 if dtype == torch.bfloat16 else 4e-3) * max(1e-3, want.abs().max().item())
        bad = (gg - want).abs() > tol + 2e-2 * want.abs()
        assert not bool(bad.any()), f"D={D} {dtype} {name}: {int(bad.sum())}/{bad.numel()} out of tolerance"
    # the logsumexp stash and a neighbour's memory: a [.., D] tensor followed directly by another must not be touched past its rows
    buf = torch.full((B * H * L * D + 64,), 7.0, dtype=dtype, device="cuda")
    o2 = buf[:B * H * L * D].view(B * H, L, D)
    lse = torch.empty(B * H, L, device="cuda", dtype=torch.float32)
    if Hkv == H:
        st = (L * D, 0, D)
        _native.fwd_strided(qc.detach().reshape(B * H, L, D), kc.detach().reshape(B * H, L, D), vc.detach().reshape(B * H, L, D), o2, lse,
                            B * H, L, D, 1, st, st, st, st, 1, 1, scale, causal)
        torch.cuda.synchronize()
        assert bool((buf[B * H * L * D:] == 7.0).all()), "stores past the true head dim"
        torch.testing.assert_close(lse.cpu().double(), ref_lse.detach().reshape(B * H, L), atol=6e-3, rtol=1e-4)
        assert torch.equal(o2.view(B, H, L, D), out.detach())


@pytest.mark.parametrize("kernel,waves,qb", [("pipe", 4, 1), ("pipe", 4, 2), ("pipe", 8, 1), ("pipe16", 4, 1), ("pipe16", 4, 2), ("pipe16", 8, 1), ("dma", 8, 1), ("dma", 4, 2),
                                             ("kh", 4, 1), ("kh2", 4, 1)])
@pytest.mark.parametrize("D,G,L", [(96, 2, 1000), (72, 3, 2304), (40, 2, 1000), (24, 9, 600), (56, 2, 2304), (120, 1, 4608)])
def test_zero_extended_head_dims_through_every_kernel_that_reads_them(knobs, kernel, waves, qb, D, G, L):
    """`FwdParams::Dr` in the software-pipelined kernels too (32x32x16 and 16x16x32 forms, one and two query blocks per wave, 4 and 8
    waves) and in the 8-wave / two-block forms of the 64-key-tile kernel: forced variants against the oracle, out and logsumexp -
    what a long-context call at head_dim 80 / 96 gets from the measured table."""
    from flash_attention_amd._impl import _native
    from oracle import dense_causal_attention_lse, frozen_randn

    Dk = 64 if D < 64 else 128
    if (qb == 2 and Dk == 128) or (waves == 8 and Dk == 64 and kernel != "dma") or (kernel == "dma" and qb == 2 and Dk == 128):
        pytest.skip("no such form at this head dim")
    if kernel.startswith("kh"):         # 64-row workgroups of (row block, key half) waves, alone or with two kv split groups
        knobs.set("kernel", "kh")
        if kernel == "kh2":
            knobs.set("split", 2)
    else:
        knobs.set("kernel", kernel)
        knobs.set("waves", waves)
        knobs.set("qb", qb)
    dtype = torch.bfloat16
    p = _native.plan(G, L, D, dtype, 1, 1)
    assert p.kernel == {"dma": 0, "pipe": 2, "pipe16": 4, "kh": 6, "kh2": 6}[kernel], (p.kernel, p.waves, p.qb)
    x = (frozen_randn(SEED + 37 * L + D, (3, G, L, D)) * 0.8).to(dtype)
    ref, ref_lse = dense_causal_attention_lse(x, scale=D ** -0.5, compute_dtype=torch.float64)
    q, k, v = x.cuda()
    buf = torch.full((G * L * D + 64,), 7.0, dtype=dtype, device="cuda")
    out = buf[:G * L * D].view(G, L, D)
    lse = torch.empty(G, L, device="cuda", dtype=torch.float32)
    st = (L * D, 0, D)
    _native.fwd_strided(q, k, v, out, lse, G, L, D, 1, st, st, st, st, 1, 1, D ** -0.5, True)
    torch.cuda.synchronize()
    assert bool((buf[G * L * D:] == 7.0).all()), "stores past the true head dim"
    err = (out.cpu().double() - ref).abs()
    assert bool((err <= 1e-2 + 1e-2 * ref.abs()).all()), f"{kernel} {waves}w qb={qb} D={D}: max|d|={err.max().item():.3e}"
    torch.testing.assert_close(lse.cpu().double(), ref_lse, atol=6e-3, rtol=1e-4)


def test_golden_g7_head_dims_from_the_reference():
    """tests/golden/g7_head_dims.npz: outputs of the reference's own Python (generate_golden.py) at D = 32, 96 (zero-extended here), 192, 256
    (native here), G = 2, L = 512, unscaled packed op.  fp32 inputs run the generic kernel (exact arithmetic: 1e-4); the 16-bit calls run the
    MFMA kernels and sit within the reference 16-bit CPU path's own rounding noise (it rounds the scores to 16 bits) + this repo's tolerance."""
    import pathlib

    import numpy as np

    import flash_attention_amd as fa
    from flash_attention_amd._impl import _native
    from oracle import frozen_randn

    g = np.load(pathlib.Path(__file__).parent / "golden" / "g7_head_dims.npz")
    G, L = (int(x) for x in g["shape"])
    rows = g["rowidx"]
    for D in (int(d) for d in g["dims"]):
        x = frozen_randn(SEED + 7 + D, (3, G, L, D)) * 0.5
        o32 = fa.flash_attention_qkv_packed(x.cuda(), 2, 2).cpu().double()
        assert (o32[:, rows, :] - torch.from_numpy(g[f"D{D}_f32_rows"])).abs().max().item() <= 1e-4
        assert abs(o32.sum().item() - float(g[f"D{D}_f32_sum"])) <= 1e-1
        for dn, dt, bound in (("f16", torch.float16, 2e-2), ("bf16", torch.bfloat16, 1e-1)):
            assert _native.plan(G, L, D, dt, 2, 2).kernel in (0, 2, 4, 6)
            o = fa.flash_attention_qkv_packed(x.to(dt).cuda(), 2, 2).cpu().double()
            assert (o[:, rows, :] - torch.from_numpy(g[f"D{D}_{dn}_rows"])).abs().max().item() <= bound, (D, dn)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("waves", [4, 8, "pipe"])
@pytest.mark.parametrize("D,G,L,causal", [(192, 2, 1000, True), (152, 3, 2304, True), (256, 2, 1000, True), (208, 1, 2304, False), (192, 9, 300, False), (256, 9, 300, False),
                                          (232, 2, 1333, True)])
def test_head_dims_192_and_256_in_both_workgroup_forms(knobs, dtype, waves, D, G, L, causal):
    """head_dim 192 / 256 (and what runs zero-extended at them) through the 64-key-tile kernel with 4 and with 8 waves (192: the rule's
    choice from 256 workgroups on; 256: no 8-wave form - it would spill 33 registers, and no shipped kernel uses scratch - a forced 8
    is served by 4), unscaled f16 with the exact row max included: out and logsumexp against the fp64 oracle, no store past a row."""
    from flash_attention_amd._impl import _native
    from oracle import dense_attention_lse, frozen_randn

    pipe = waves == "pipe"      # the software-pipelined kernel's 4-wave form: head_dim 256 (and 200 .. 248 at it), skip-max arithmetic only
    if pipe and D <= 192:
        pytest.skip("no pipelined form at head_dim 192")
    scale = D ** -0.5 if (dtype == torch.bfloat16 or pipe) else 0.5     # (f16 at scale 0.5 > 1/4: the exact-max arithmetic)
    if pipe:
        knobs.set("kernel", "pipe")
    else:
        knobs.set("kernel", "dma")
        knobs.set("waves", waves)
    p = _native.plan(G, L, D, dtype, 1, 1, softmax_scale=scale, causal=causal)
    w_eff = 4 if (not pipe and waves == 8 and D > 192) else waves
    assert (p.kernel, p.waves, p.block_q, p.qb) == ((2, 4, 128, 1) if pipe else (0, w_eff, 32 * w_eff, 1))
    x = (frozen_randn(SEED + 41 * L + D, (3, G, L, D)) * (0.8 if dtype == torch.bfloat16 else 0.3)).to(dtype)
    ref, ref_lse = dense_attention_lse(x[0][None].double(), x[1][None].double(), x[2][None].double(), scale=scale, causal=causal)
    q, k, v = x.cuda()
    buf = torch.full((G * L * D + 64,), 7.0, dtype=dtype, device="cuda")
    out = buf[:G * L * D].view(G, L, D)
    lse = torch.empty(G, L, device="cuda", dtype=torch.float32)
    st = (L * D, 0, D)
    _native.fwd_strided(q, k, v, out, lse, G, L, D, 1, st, st, st, st, 1, 1, scale, causal)
    torch.cuda.synchronize()
    assert bool((buf[G * L * D:] == 7.0).all()), "stores past the true head dim"
    err = (out.cpu().double() - ref[0]).abs()
    assert bool((err <= 1e-2 + 1e-2 * ref[0].abs()).all()), f"{waves} D={D} {dtype}: max|d|={err.max().item():.3e}"
    torch.testing.assert_close(lse.cpu().double(), ref_lse[0], atol=6e-3, rtol=1e-4)


def test_mask_and_dropout_at_head_dim_96():
    """fa_fwd_ex / fa_bwd_ex at a zero-extended head dim: additive bias + dropout + causal, out and gradients against oracle/dropout.py."""
    from flash_attention_amd._impl import ops
    from oracle import dense_attention_extra, frozen_randn

    B, H, L, D, dtype, p_drop = 2, 2, 320, 96, torch.bfloat16, 0.2
    q, k, v = ((frozen_randn(SEED + 40 + i, (B, H, L, D)) * 0.6).to(dtype) for i in range(3))
    g = frozen_randn(SEED + 44, (B, H, L, D)).to(dtype)
    mask = (0.5 * torch.randn(1, H, L, L, generator=torch.Generator().manual_seed(3))).to(dtype)
    qr, kr, vr = (t.double().requires_grad_(True) for t in (q, k, v))
    ref, _ = dense_attention_extra(qr, kr, vr, D ** -0.5, True, mask, p_drop, 11, 22)
    ref.backward(g.double())
    qc, kc, vc = (t.cuda().requires_grad_(True) for t in (q, k, v))
    out = ops.attention_extra(qc, kc, vc, mask.cuda().expand(B, H, L, L), D ** -0.5, True, p_drop, 11, 22)
    err = (out.detach().cpu().double() - ref.detach()).abs()
    assert bool((err <= 1e-2 + 1e-2 * ref.detach().abs()).all()), f"max|d|={err.max().item():.3e}"
    out.backward(g.cuda())
    for got, want, name in ((qc.grad, qr.grad, "dq"), (kc.grad, kr.grad, "dk"), (vc.grad, vr.grad, "dv")):
        gg = got.detach().cpu().double()
        bad = (gg - want).abs() > 1.5e-2 * max(1e-3, want.abs().max().item()) + 2e-2 * want.abs()
        assert torch.isfinite(gg).all() and not bool(bad.any()), f"{name}: {int(bad.sum())} out of tolerance"


@pytest.mark.parametrize("D,dtype", [(100, torch.bfloat16), (196, torch.bfloat16), (96, torch.float32)])
def test_padded_head_dims(D, dtype):
    """What the kernels do not zero-extend - a head dim that is no multiple of 8, fp32 off 64 / 128 - runs PADDED on a large call
    (override and packed op): forward against the oracle, gradients finite and shaped."""
    import flash_attention_amd as fa
    from oracle import dense_causal_attention, frozen_randn

    G, L = 64, 512
    qkv = (frozen_randn(SEED + D, (3, G, L, D)) * 0.4).to(dtype)
    dev = qkv.cuda().requires_grad_(True)
    out = fa.flash_attention_qkv_packed(dev, 2, 2)
    assert out.shape == (G, L, D) and out.is_contiguous()
    ref = dense_causal_attention(qkv[:, :3], compute_dtype=torch.float64)
    err = (out[:3].detach().cpu().double() - ref).abs()
    assert bool((err <= (1e-2 + 1e-2 * ref.abs() if dtype != torch.float32 else 2e-4 + 1e-4 * ref.abs())).all()), f"max|d|={err.max().item():.3e}"
    (gq,) = torch.autograd.grad(out, dev, torch.ones_like(out))
    assert gq.shape == dev.shape and torch.isfinite(gq).all()


@pytest.mark.parametrize("D", [32, 96])
def test_packed_op_at_other_head_dims(D):
    """flash_attention_qkv_packed([3, 128, 1024, D]) bf16: served by the MFMA kernel zero-extended (no copies), oracle-checked incl.
    gradients, and timed against the same call explicitly padded to the MFMA head dim (what round 4 did first: three copies more).
    (The generic kernel such a call used to reach measured 1.2 TFLOP/s at D = 96 and 2.0 at D = 32: DESIGN.md 3.3b.)"""
    import flash_attention_amd as fa
    from flash_attention_amd._impl import ops
    from oracle import dense_causal_attention, frozen_randn

    G, L = 128, 1024
    qkv = (frozen_randn(SEED + D, (3, G, L, D)) * 0.4).to(torch.bfloat16)
    dev = qkv.cuda().requires_grad_(True)
    out = fa.flash_attention_qkv_packed(dev, 4, 4)
    assert out.shape == (G, L, D) and out.is_contiguous()
    ref = dense_causal_attention(qkv[:, :4], compute_dtype=torch.float64)
    err = (out[:4].detach().cpu().double() - ref).abs()
    assert bool((err <= 1e-2 + 1e-2 * ref.abs()).all()), f"max|d|={err.max().item():.3e}"
    (g,) = torch.autograd.grad(out, dev, torch.ones_like(out))
    assert g.shape == dev.shape and torch.isfinite(g).all()
    Dp = 64 if D < 64 else 128

    def padded():
        return ops.fwd_qkv_packed(F.pad(dev.detach(), (0, Dp - D)), 4, 4)[0][..., :D].contiguous()

    err = (padded()[:4].cpu().double() - ref).abs()
    assert bool((err <= 1e-2 + 1e-2 * ref.abs()).all())
    with torch.no_grad():
        t_native = _time(lambda: fa.flash_attention_qkv_packed(dev, 4, 4), n=10)
        t_pad = _time(padded, n=10)
    flops = 2.0 * G * L * L * D
    print(f"[packed D={D}] zero-extended in the kernel {t_native * 1e3:.1f} us = {flops / t_native / 1e9:.1f} TFLOP/s; "
          f"explicitly padded to {Dp} {t_pad * 1e3:.1f} us = {flops / t_pad / 1e9:.1f} TFLOP/s")
    assert t_native < 1.1 * t_pad
