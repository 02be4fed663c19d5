"""GPU tests of the backward kernels (flashAttentionQKVPackedGrad, FA.cpp:348-512) through
autograd + the C ABI, against autograd through the fp64 oracle on the same 16-bit-rounded inputs.

The reference's own backward bounds are loose: max|delta| <= 1e-4 fp32, <= 1.0 fp16 at G=8,L=2048,D=128
(t_cpp:271-295) and atol/rtol 1e-2 on grads in the Python tests (t_fa.py:88-93, t_sdpa.py:96-113).
Gradients are sums of many signed 16-bit-rounded products, so their error scales with the tensor's
magnitude rather than with each element: the check is |delta| <= tol*max|ref| + 2e-2*|ref| with
tol = 4e-3 (fp16) / 1.5e-2 (bf16: 3 fewer mantissa bits)."""

import pytest
import torch

pytestmark = pytest.mark.gpu
SEED = 1123581321


def _oracle_grads(qkv16, dout16, scale):
    from oracle.dense_attention import MASK_VALUE

    x = qkv16.double().requires_grad_(True)
    q, k, v = x
    L = q.shape[1]
    mask = torch.triu(torch.full((L, L), MASK_VALUE, dtype=torch.float64), 1)
    p = torch.softmax((q @ k.transpose(1, 2)) * scale + mask, dim=-1)
    out = p @ v
    out.backward(dout16.double())
    return out.detach(), x.grad.detach()


TOL = {torch.float16: 4e-3, torch.bfloat16: 1.5e-2}


def _close(got, ref, what):
    tol = TOL[got.dtype]
    got = got.detach().cpu().double()
    err = (got - ref).abs()
    floor = tol * max(1e-3, ref.abs().max().item())
    bad = err > floor + 2e-2 * ref.abs()
    assert torch.isfinite(got).all(), what
    assert not bool(bad.any()), f"{what}: {int(bad.sum())}/{bad.numel()} out of tolerance, max|d|={err.max().item():.3e}, ref max {ref.abs().max().item():.2f}"


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("G,L,D", [(2, 64, 64), (1, 128, 64), (3, 200, 64), (2, 257, 64), (2, 512, 128), (1, 1000, 128), (2, 1024, 64)])
def test_packed_backward_vs_oracle(dtype, G, L, D):
    import flash_attention_amd as fa
    from oracle import frozen_randn

    qkv = (frozen_randn(SEED + L + D, (3, G, L, D)) * 0.5).to(dtype)   # logits std ~ sqrt(D)/4: softmax not one-hot
    dout = frozen_randn(SEED + 1, (G, L, D)).to(dtype)
    ref_out, ref_grad = _oracle_grads(qkv, dout, 1.0)
    x = qkv.cuda().requires_grad_(True)
    out = fa.flash_attention_qkv_packed(x, 1, 1)
    out.backward(dout.cuda())
    _close(out, ref_out, "out")
    for i, name in enumerate("qkv"):
        _close(x.grad[i], ref_grad[i], f"d{name} {dtype} {G}x{L}x{D}")


@pytest.mark.parametrize("kernel", ["pipe", "pipe16"])
def test_backward_after_every_forward_kernel(knobs, kernel):
    """The backward consumes the forward's `out` and log-sum-exp: the same gradients whichever forward kernel
    produced them (the default at these sizes is the 64-key-tile kernel, covered above)."""
    import flash_attention_amd as fa
    from oracle import frozen_randn

    knobs.set("FA_AMD_KERNEL", kernel)
    G, L, D, dtype = 2, 1024, 64, torch.bfloat16
    qkv = (frozen_randn(SEED + L + D, (3, G, L, D)) * 0.5).to(dtype)
    dout = frozen_randn(SEED + 1, (G, L, D)).to(dtype)
    ref_out, ref_grad = _oracle_grads(qkv, dout, 1.0)
    x = qkv.cuda().requires_grad_(True)
    out = fa.flash_attention_qkv_packed(x, 1, 1)
    out.backward(dout.cuda())
    _close(out, ref_out, f"out ({kernel})")
    for i, name in enumerate("qkv"):
        _close(x.grad[i], ref_grad[i], f"d{name} after the {kernel} forward")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_sdpa_backward_vs_stock(dtype):
    """t_sdpa.py:96-113: dq, dk, dv of the patched SDPA vs stock SDPA math (fp64 here), batch shape (2,3)."""
    import torch.nn.functional as F

    import flash_attention_amd.auto  # noqa: F401
    from oracle import frozen_randn

    L, D = 256, 64
    q, k, v = frozen_randn(SEED + 5, (3, 2, 3, L, D)).to(dtype)
    dout = frozen_randn(SEED + 6, (2, 3, L, D)).to(dtype)
    ref_out, ref_grad = _oracle_grads(torch.stack([q, k, v]).reshape(3, 6, L, D), dout.reshape(6, L, D), D ** -0.5)
    qc, kc, vc = (t.cuda().requires_grad_(True) for t in (q, k, v))
    out = F.scaled_dot_product_attention(qc, kc, vc, is_causal=True)
    out.backward(dout.cuda())
    _close(out.reshape(6, L, D), ref_out, "sdpa out")
    for t, r, name in zip((qc, kc, vc), ref_grad, "qkv"):
        _close(t.grad.reshape(6, L, D), r, f"sdpa d{name}")


def test_backward_is_deterministic_and_group_independent():
    import flash_attention_amd as fa

    gen = torch.Generator(device="cuda").manual_seed(3)
    qkv = (torch.randn(3, 16, 2048, 64, device="cuda", generator=gen) * 0.5).to(torch.bfloat16)
    dout = torch.randn(16, 2048, 64, device="cuda", generator=gen).to(torch.bfloat16)

    def grads(x):
        x = x.clone().requires_grad_(True)
        fa.flash_attention_qkv_packed(x, 1, 1).backward(dout)
        return x.grad

    g1, g2 = grads(qkv), grads(qkv)
    assert torch.equal(g1, g2)                                  # no atomics: bitwise reproducible
    perm = torch.randperm(16, device="cuda", generator=gen)
    x = qkv[:, perm].clone().requires_grad_(True)
    fa.flash_attention_qkv_packed(x, 1, 1).backward(dout[perm])
    assert torch.equal(x.grad, g1[:, perm])                     # groups never interact
    assert torch.isfinite(g1).all()
    # dV of the last key only sees the last query row: dV[L-1] = P[L-1,L-1] * dO[L-1]
    assert torch.all(g1[2, :, -1].abs() <= dout[:, -1].abs().float() * 1.01 + 1e-3)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("G,L,D", [(4, 6, 2), (2, 33, 7), (2, 70, 24), (8, 256, 128), (2, 300, 80), (1, 1200, 32)])
def test_generic_backward(knobs, dtype, G, L, D):
    """Shapes/dtypes only the generic backward kernels serve, incl. the reference's fp32 C++ cases
    4x6x2 and 8x256x128 (t_cpp:273-287, bound 1e-4 there)."""
    import flash_attention_amd as fa
    from oracle import frozen_randn

    if dtype != torch.float32 and D in (64, 128):
        pytest.skip("served by the MFMA kernels")
    knobs.set("f32_generic", 1)      # fp32 at head_dim 64 / 128 has MFMA kernels of its own (test below)
    qkv = (frozen_randn(SEED + L + D, (3, G, L, D)) * (0.5 if D > 16 else 1.0)).to(dtype)
    dout = frozen_randn(SEED + 2, (G, L, D)).to(dtype)
    ref_out, ref_grad = _oracle_grads(qkv, dout, 1.0)
    x = qkv.cuda().requires_grad_(True)
    out = fa.flash_attention_qkv_packed(x, 1, 1)
    out.backward(dout.cuda())
    if dtype == torch.float32:
        assert (x.grad.cpu().double() - ref_grad).abs().max().item() <= 1e-4 * max(1.0, ref_grad.abs().max().item())
    else:
        for i, name in enumerate("qkv"):
            _close(x.grad[i], ref_grad[i], f"generic d{name} {dtype} {G}x{L}x{D}")


@pytest.mark.parametrize("causal", [True, False])
@pytest.mark.parametrize("G,L,D", [(8, 256, 128), (3, 200, 64), (2, 33, 64), (1, 1000, 128), (2, 2048, 64)])
def test_fp32_mfma_backward(knobs, G, L, D, causal):
    """fp32 gradients at head_dim 64 / 128: the v_mfma_f32_32x32x2_f32 kernels (fa_bwd_f32.hip) against autograd
    through the fp64 oracle with the reference's fp32 bound (max|delta| <= 1e-4, t_cpp:273-287, scaled by the
    gradient's magnitude) and against the generic kernels they replace; causal and unmasked."""
    import torch.nn.functional as F

    import flash_attention_amd.auto  # noqa: F401
    from oracle import dense_attention_lse, frozen_randn

    q0, k0, v0 = (frozen_randn(SEED + L + D, (3, 1, G, L, D)) * 0.5)
    g0 = frozen_randn(SEED + 3, (1, G, L, D))
    scale = D ** -0.5
    qr, kr, vr = (t.double().requires_grad_(True) for t in (q0, k0, v0))
    ref, _ = dense_attention_lse(qr, kr, vr, scale=scale, causal=causal)
    ref.backward(g0.double())

    def run():
        q, k, v = (t.cuda().requires_grad_(True) for t in (q0, k0, v0))
        out = F.scaled_dot_product_attention(q, k, v, is_causal=causal)
        out.backward(g0.cuda())
        return out.detach(), q.grad, k.grad, v.grad

    got = run()
    knobs.set("f32_generic", 1)
    gen = run()
    for name, a, b, want in zip(("out", "dq", "dk", "dv"), got, gen, (ref.detach(), qr.grad, kr.grad, vr.grad)):
        bound = 1e-4 * max(1.0, want.abs().max().item())
        assert torch.isfinite(a).all(), name
        assert (a.cpu().double() - want).abs().max().item() <= bound, (name, (a.cpu().double() - want).abs().max().item(), bound)
        assert (a - b).abs().max().item() <= 2 * bound, (name, "vs generic kernels")


def test_eager_fast_path_and_custom_ops_agree_bitwise():
    """Eager calls run plain autograd.Functions, torch.compile traces the torch.library custom ops; both call the same
    implementations, so outputs and gradients must be identical bit for bit (packed op and three-operand op)."""
    from flash_attention_amd._impl import ops
    from oracle import frozen_randn

    qkv = (frozen_randn(SEED + 90, (3, 3, 300, 64)) * 0.5).to(torch.bfloat16).cuda()
    dout = frozen_randn(SEED + 91, (3, 300, 64)).to(torch.bfloat16).cuda()

    def grads(fn):
        x = qkv.clone().requires_grad_(True)
        out = fn(x)
        out.backward(dout)
        return out.detach(), x.grad

    o1, g1 = grads(lambda x: ops.attention_qkv_packed(x, 1, 1))
    o2, g2 = grads(lambda x: ops.fwd_qkv_packed(x, 1, 1)[0])
    assert torch.equal(o1, o2) and torch.equal(g1, g2)
    for causal in (True, False):
        o3, g3 = grads(lambda x: ops.attention(x[0], x[1], x[2], 0.125, 1, 1, causal))
        o4, g4 = grads(lambda x: ops.fwd(x[0], x[1], x[2], 0.125, 1, 1, causal)[0])
        assert torch.equal(o3, o4) and torch.equal(g3, g4)
    q4 = qkv[0].view(1, 3, 300, 64)
    assert torch.equal(ops.attention_bhld(q4, q4, q4, 0.125, 1, 1, True), ops.fwd_bhld(q4, q4, q4, 0.125, 1, 1, True))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_gradients_from_the_kernels_lse_vs_the_oracles_lse(dtype):
    """The forward's logsumexp stash is formed from the sum of the 16-bit-rounded P (the denominator the PV product really used):
    up to 2^-8 relative (bf16) away from the reference definition m + log(l) (FA.cpp:316).  The backward recomputes P = exp(S - lse)
    from it, so this quantifies what that costs: gradients computed by the SAME backward kernels from the kernel's stash and from
    the fp64 oracle's logsumexp, both against autograd through the oracle.  The stash's error must not be what bounds them."""
    import flash_attention_amd as fa
    from flash_attention_amd._impl import _native
    from flash_attention_amd._impl.flash_attention_qkv_packed import flash_attention_qkv_packed_with_lse
    from oracle import dense_causal_attention_lse, frozen_randn

    G, L, D = 4, 1024, 64
    qkv = (frozen_randn(SEED + 77, (3, G, L, D)) * 0.5).to(dtype)
    dout = frozen_randn(SEED + 78, (G, L, D)).to(dtype)
    _, ref_grad = _oracle_grads(qkv, dout, 1.0)
    _, ref_lse = dense_causal_attention_lse(qkv, scale=1.0)
    x = qkv.cuda()
    out, lse = flash_attention_qkv_packed_with_lse(x, 1, 1)
    lse_err = (lse.cpu().double() - ref_lse).abs().max().item()
    grads = {}
    for name, stash in (("kernel", lse), ("oracle", ref_lse.float().cuda())):
        dqkv = torch.empty_like(x)
        _native.bwd(x[0], x[1], x[2], out, dout.cuda(), stash, dqkv[0], dqkv[1], dqkv[2], G, L, D, 1, 1.0, True)
        grads[name] = dqkv.cpu().double()
    scale = ref_grad.abs().max().item()
    e_k = (grads["kernel"] - ref_grad).abs().max().item() / scale
    e_o = (grads["oracle"] - ref_grad).abs().max().item() / scale
    rms_k = ((grads["kernel"] - ref_grad).square().mean().sqrt() / ref_grad.square().mean().sqrt()).item()
    rms_o = ((grads["oracle"] - ref_grad).square().mean().sqrt() / ref_grad.square().mean().sqrt()).item()
    print(f"\\n{dtype}: max|dLSE| = {lse_err:.2e}; gradient error / max|grad|: kernel stash {e_k:.3e} (rms {rms_k:.3e}), oracle stash {e_o:.3e} (rms {rms_o:.3e})")
    assert lse_err <= (6e-3 if dtype == torch.bfloat16 else 2e-3)
    # the gradients' own 16-bit rounding dominates: the kernel's stash may cost at most a quarter on top of the exact stash's error
    assert rms_k <= 1.25 * rms_o + 1e-4, (rms_k, rms_o)
    assert e_k <= 1.5 * e_o + 1e-3, (e_k, e_o)
    for i, n in enumerate("qkv"):
        _close(grads["kernel"][i].to(dtype), ref_grad[i], f"d{n} from the kernel's stash")


def _blocked_check(x, dout, groups, scale, what, slab=2048):
    """gradients of `groups` of a packed launch against the blocked fp64 oracle (one [slab, L] score slab at a time)"""
    from oracle import dense_causal_attention_grads_blocked

    for g in groups:
        q, k, v = (x.detach()[i, g].cpu() for i in range(3))
        _, dq, dk, dv = dense_causal_attention_grads_blocked(q, k, v, dout[g].cpu(), scale=scale, slab=slab)
        for got, want, name in ((x.grad[0, g], dq, "dq"), (x.grad[1, g], dk, "dk"), (x.grad[2, g], dv, "dv")):
            _close(got, want, f"{what} group {g} {name}")


def test_backward_at_the_benchmarked_shape_cfg2():
    """`secondary.cfg2_bwd` of bench.py times fa_bwd on packed qkv [3, 128, 4096, 64] bf16: the same launch here, three sampled groups
    against the fp64 oracle (the dK/dV workgroups sweep 32+ query tiles per key block only from this length on) + determinism and
    group-permutation equivariance of the WHOLE launch (round-3 verdict: the backward was timed where it was not checked)."""
    import flash_attention_amd as fa

    G, L, D = 128, 4096, 64
    gen = torch.Generator(device="cuda").manual_seed(11)
    qkv = (torch.randn(3, G, L, D, device="cuda", generator=gen) * 0.5).to(torch.bfloat16)
    dout = torch.randn(G, L, D, device="cuda", generator=gen).to(torch.bfloat16)
    x = qkv.clone().requires_grad_(True)
    fa.flash_attention_qkv_packed(x, 8, 8).backward(dout)
    assert torch.isfinite(x.grad).all()
    _blocked_check(x, dout, (0, 77, G - 1), 1.0, "cfg2_bwd")
    x2 = qkv.clone().requires_grad_(True)
    fa.flash_attention_qkv_packed(x2, 8, 8).backward(dout)
    assert torch.equal(x2.grad, x.grad), "no atomics: bitwise reproducible"
    perm = torch.randperm(G, device="cuda", generator=gen)
    x3 = qkv[:, perm].clone().requires_grad_(True)
    fa.flash_attention_qkv_packed(x3, 8, 8).backward(dout[perm])
    assert torch.equal(x3.grad, x.grad[:, perm]), "groups never interact"


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("L,nq,nk", [(4096, 8, 4), (16384, 16, 8)])
def test_backward_at_the_reference_python_test_sizes(dtype, L, nq, nk):
    """t_fa.py:56-93: N = 4, D = 128, L in {256, 1024, 4096, 16384}, out AND grads (the two short lengths are covered above and in
    test_packed_backward_vs_oracle).  One group through the blocked fp64 oracle at L = 16384 (268 MB per score slab), two at 4096;
    chunk counts as the reference's heuristic picks them (sdpa.py:52-53)."""
    import flash_attention_amd as fa

    N, D = 4, 128
    gen = torch.Generator(device="cuda").manual_seed(L)
    qkv = (torch.randn(3, N, L, D, device="cuda", generator=gen) * 0.35).to(dtype)      # unscaled logits: std ~ 1.4
    dout = torch.randn(N, L, D, device="cuda", generator=gen).to(dtype)
    x = qkv.clone().requires_grad_(True)
    out = fa.flash_attention_qkv_packed(x, nq, nk)
    out.backward(dout)
    assert torch.isfinite(x.grad).all()
    _blocked_check(x, dout, (N - 1,) if L > 4096 else (0, N - 1), 1.0, f"t_fa {dtype} L={L}")


def test_backward_fp16_at_the_reference_cpp_test_size():
    """t_cpp:289-295: fp16, G = 8, L = 2048, D = 128 backward (the reference bounds max|delta| by 1.0 there; the bound here is the
    suite's own, tol * max|ref| + 2e-2 |ref|) - every group against autograd through the fp64 oracle."""
    import flash_attention_amd as fa
    from oracle import frozen_randn

    G, L, D = 8, 2048, 128
    qkv = (frozen_randn(SEED + 2048, (3, G, L, D)) * 0.35).to(torch.float16)
    dout = frozen_randn(SEED + 2049, (G, L, D)).to(torch.float16)
    x = qkv.cuda().requires_grad_(True)
    out = fa.flash_attention_qkv_packed(x, 4, 4)
    out.backward(dout.cuda())
    _blocked_check(x, dout, range(G), 1.0, "t_cpp fp16 8x2048x128")
    assert (x.grad.float() - 0).abs().max().item() < 1e4


def test_override_training_at_the_headline_shape_on_head_inner_operands():
    """Round 6: the eager training path of the override (`ops._BhldAttentionFn`) at the headline shape `[8,16,4096,64]` on nanoGPT's operand layout -
    q, k, v are head-inner slices of ONE fused [B, T, 3C] projection, read through their strides (no reshape, no copy).  Result and gradients come
    back head-inner; three sampled groups against the blocked fp64 oracle; bit-identical to the same call on contiguous copies of the operands
    (the layout is an addressing matter, not an arithmetic one) and from call to call."""
    import torch.nn.functional as F

    import flash_attention_amd.auto  # noqa: F401

    B, H, L, D = 8, 16, 4096, 64
    gen = torch.Generator(device="cuda").manual_seed(23)
    proj = (torch.randn(B, L, 3 * H * D, device="cuda", generator=gen) * 0.8).to(torch.bfloat16).requires_grad_(True)
    dout = torch.randn(B, H, L, D, device="cuda", generator=gen).to(torch.bfloat16)

    def run(p, contiguous):
        q, k, v = (t.view(B, L, H, D).transpose(1, 2) for t in p.split(H * D, dim=2))
        if contiguous:
            q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        out = F.scaled_dot_product_attention(q, k, v, is_causal=True)
        (g,) = torch.autograd.grad(out, p, dout)
        return out, g

    out, g = run(proj, False)
    assert out.stride() == (L * H * D, D, H * D, 1), "head-inner operands: head-inner result"
    assert torch.isfinite(g).all()
    out2, g2 = run(proj, False)
    assert torch.equal(out, out2) and torch.equal(g, g2)
    outc, gc = run(proj, True)
    assert torch.equal(out, outc) and torch.equal(g, gc), "strided and contiguous operands: the same arithmetic"
    # as a packed launch for the oracle: groups (batch 0 head 0), (batch 3 head 9), (batch 7 head 15)
    qkv = proj.detach().view(B, L, 3, H, D).permute(2, 0, 3, 1, 4).reshape(3, B * H, L, D)
    x = qkv.clone().requires_grad_(True)
    x.grad = g.view(B, L, 3, H, D).permute(2, 0, 3, 1, 4).reshape(3, B * H, L, D).contiguous()
    _blocked_check(x, dout.reshape(B * H, L, D), (0, 3 * H + 9, B * H - 1), D ** -0.5, "override training, head-inner, cfg2 shape")
