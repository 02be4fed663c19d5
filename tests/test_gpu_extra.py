"""GPU tests of attention masks / additive biases and dropout (the `attn_mask` and `dropout_p` modes of
scaled_dot_product_attention that the reference rejects, flash_attention_ipu/_impl/scaled_dot_product_attention.py:21-28).

There is no reference behaviour to restate, so the yardstick is torch's own definition of the op evaluated densely in fp64 on
the same 16-bit-rounded inputs (oracle.dense_attention_extra), with the dropout decisions taken by the numpy restatement of the
kernels' counter-based generator (oracle/dropout.py) - forward AND gradients (autograd through the oracle).  Tolerances are the
forward's (|d| <= 1e-2 + 1e-2 |ref|, t_fa.py:78-79) and the backward's (tests/test_gpu_backward.py).
"""

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
SEED = 1123581321
TOL = {torch.float16: 4e-3, torch.bfloat16: 1.5e-2}
LSE_ATOL = 6e-3


def _fwd_close(out, ref, what):
    out = out.detach().cpu().double()
    nan_ref = torch.isnan(ref)
    assert torch.equal(torch.isnan(out), nan_ref), f"{what}: NaN rows differ (rows without any admitted key must be NaN, no others)"
    err = (out - ref).abs()[~nan_ref]
    bad = err > (1e-2 + 1e-2 * ref.abs()[~nan_ref])
    assert not bool(bad.any()), f"{what}: {int(bad.sum())} elements out of tolerance, max|delta|={err.max().item():.3e}"


def _grad_close(got, ref, what):
    tol = TOL[got.dtype]
    got = got.detach().cpu().double()
    err = (got - ref).abs()
    floor = tol * max(1e-3, ref.abs().max().item())
    bad = err > floor + 2e-2 * ref.abs()
    assert torch.isfinite(got).all(), what
    assert not bool(bad.any()), f"{what}: {int(bad.sum())}/{bad.numel()} out of tolerance, max|d|={err.max().item():.3e}, ref max {ref.abs().max().item():.2f}"


def _masks(kind, B, H, L, dtype, gen):
    """attn_mask variants a caller may pass (each row keeps at least one key unless the variant says otherwise)."""
    if kind == "bool_LL":            # one [L, L] boolean mask for every group: a random pattern with the diagonal kept
        m = torch.rand(L, L, generator=gen) < 0.7
        return m | torch.eye(L, dtype=torch.bool)
    if kind == "bool_padding":       # key-padding mask [B, 1, 1, L]: each batch entry sees a prefix of the keys
        lens = torch.randint(1, L + 1, (B,), generator=gen)
        return (torch.arange(L)[None, :] < lens[:, None]).view(B, 1, 1, L)
    if kind == "bool_BHLL_dead_rows":     # per-head masks, some rows admit NO key at all (-> NaN rows, as torch)
        m = torch.rand(B, H, L, L, generator=gen) < 0.5
        m[:, :, ::7] = False
        m[:, :, 3] = True
        return m
    if kind == "bias_BHLL":          # additive bias per head (ALiBi-like slopes + noise), a few -inf entries
        pos = torch.arange(L)
        b = -(pos[:, None] - pos[None, :]).abs().float()[None, None] * torch.linspace(0.01, 0.2, H)[None, :, None, None]
        b = b + 0.5 * torch.randn(B, H, L, L, generator=gen)
        b[:, :, :, 1::5] = float("-inf")
        b[:, :, :, 0] = 0.0
        return b.to(dtype)
    if kind == "bias_1HLL":          # the same bias for every batch entry (stride 0 on the batch)
        return (0.7 * torch.randn(1, H, L, L, generator=gen)).to(dtype)
    if kind == "bias_row":           # [L] bias along the keys, broadcast over rows
        return torch.randn(L, generator=gen).to(dtype)
    if kind == "bias_head_key":      # [H, 1, L]: a per-key bias that differs from head to head, each head with its own -inf tail
        b = torch.randn(H, 1, L, generator=gen)
        for hh in range(H):
            b[hh, 0, L - min(L - 1, 70 * (hh + 1)):] = float("-inf")
        return b.to(dtype)
    raise AssertionError(kind)


@pytest.mark.parametrize("kind", ["bool_LL", "bool_padding", "bool_BHLL_dead_rows", "bias_BHLL", "bias_1HLL", "bias_row", "bias_head_key"])
@pytest.mark.parametrize("dtype,D", [(torch.bfloat16, 64), (torch.float16, 128), (torch.bfloat16, 192), (torch.float16, 160), (torch.bfloat16, 256), (torch.float16, 232)])
@pytest.mark.parametrize("B,H,Hkv,L", [(2, 3, 3, 200), (1, 4, 2, 256), (2, 2, 1, 333), (1, 2, 2, 1024), (1, 2, 1, 2308)])
def test_masked_forward_and_backward_vs_oracle(kind, dtype, D, B, H, Hkv, L):
    """attn_mask through fa_fwd_ex / fa_bwd_ex: boolean and additive masks, every broadcast pattern, strides that are multiples of four
    elements (mask tiles by LDS-DMA; L = 200, 2308: rows that are not 16-byte aligned) and strides that are not (L = 333: element loads),
    grouped K/V heads (a per-head per-key bias under a shared K/V head), rows without any admitted key, key tiles without any
    admitted key (padding tails), more key blocks and tiles than LDS buffers (L = 2308); head_dim 192 and 160 (zero-extended at 192), 256 and
    232 (zero-extended at 256: the backward's dK/dV as a dV pass and a dK pass)."""
    from flash_attention_amd._impl import ops
    from oracle import dense_attention_extra, frozen_randn

    gen = torch.Generator().manual_seed(SEED + L + H)
    q = (frozen_randn(SEED + L, (B, H, L, D)) * 0.6).to(dtype)
    k = (frozen_randn(SEED + L + 1, (B, Hkv, L, D)) * 0.6).to(dtype)
    v = (frozen_randn(SEED + L + 2, (B, Hkv, L, D)) * 0.6).to(dtype)
    g = frozen_randn(SEED + L + 3, (B, H, L, D)).to(dtype)
    mask = _masks(kind, B, H, L, dtype, gen)
    scale = D ** -0.5
    qr, kr, vr = (t.double().requires_grad_(True) for t in (q, k, v))
    ref, ref_lse = dense_attention_extra(qr, kr, vr, scale, False, mask)
    ok_rows = ~torch.isnan(ref.detach()).any(-1, keepdim=True)
    torch.where(ok_rows, ref, torch.zeros_like(ref)).backward(g.double())      # (NaN rows carry no gradient in the comparison)
    qc, kc, vc = (t.cuda().requires_grad_(True) for t in (q, k, v))
    mc = mask.cuda().expand(B, H, L, L)
    mc = mc.view(torch.uint8) if mc.dtype == torch.bool else mc
    if mc.stride(-1) != 1:
        mc = mc.contiguous()
    out = ops.attention_extra(qc, kc, vc, mc, scale, False)
    _fwd_close(out, ref.detach(), f"masked forward {kind} {dtype} {B}x{H}/{Hkv}x{L}x{D}")
    torch.where(ok_rows.cuda(), out, torch.zeros_like(out)).backward(g.cuda())
    if "dead_rows" not in kind:          # (a dead row's P is 0 in the kernels and NaN in the dense formula: compare live masks only)
        for name, got, want in (("dq", qc.grad, qr.grad), ("dk", kc.grad, kr.grad), ("dv", vc.grad, vr.grad)):
            _grad_close(got, want, f"{name} {kind} {dtype} {B}x{H}/{Hkv}x{L}x{D}")


@pytest.mark.parametrize("kind", ["bool", "bias"])
@pytest.mark.parametrize("dtype,D", [(torch.float16, 64), (torch.bfloat16, 128), (torch.bfloat16, 256), (torch.float16, 192)])
@pytest.mark.parametrize("L,pad,off", [(115, 5, 4), (115, 5, 0), (201, 7, 8), (126, 2, 4), (333, 3, 12)])
def test_mask_view_with_padded_rows_and_a_length_that_ends_inside_a_dword_vs_oracle(kind, dtype, D, L, pad, off):
    """A mask that is a view of a buffer with padded rows: row stride and pointer multiples of four elements (the LDS-DMA tile images), L not - the
    last key of the last row then shares its dword with the element after it.  Found by tests/tools/fuzz_extra.py in round 6: the tile's
    buffer descriptor ended exactly at key L - 1 of the slice's last row, the per-dword range check zeroed that dword, and bias[L-1][L-1] (a
    boolean mask: up to three keys of the last row) was lost in the forward and in all three gradients.  The extent now covers whole dwords."""
    from flash_attention_amd._impl import ops
    from oracle import dense_attention_extra, frozen_randn

    B, H, Hkv = 2, 2, 1
    assert (L + pad) % 4 == 0 and off % 4 == 0 and L % 4 != 0
    gen = torch.Generator().manual_seed(SEED + L + D)
    q = (frozen_randn(SEED + L, (B, H, L, D)) * 0.6).to(dtype)
    k = (frozen_randn(SEED + L + 1, (B, Hkv, L, D)) * 0.6).to(dtype)
    v = (frozen_randn(SEED + L + 2, (B, Hkv, L, D)) * 0.6).to(dtype)
    g = frozen_randn(SEED + L + 3, (B, H, L, D)).to(dtype)
    if kind == "bool":
        big = torch.rand(1, H, L + 1, L + pad + off, generator=gen) < 0.6
        big[..., off] = True                     # (every row keeps its first key)
        big[..., -1, off + L - 3:off + L] = True   # the keys in question are admitted, and carry weight: their scores are the row's largest
    else:
        big = (0.8 * torch.randn(1, H, L + 1, L + pad + off, generator=gen)).to(dtype)
        big[..., -1, off + L - 1] = 3.0          # the element in question matters: the last row's largest term
    view = big[:, :, 1:1 + L, off:off + L]
    scale = D ** -0.5
    qr, kr, vr = (t.double().requires_grad_(True) for t in (q, k, v))
    ref, _ = dense_attention_extra(qr, kr, vr, scale, False, view)
    ref.backward(g.double())
    bc = big.cuda()
    mc = bc[:, :, 1:1 + L, off:off + L].expand(B, H, L, L)
    mc = mc.view(torch.uint8) if mc.dtype == torch.bool else mc
    assert mc.stride(-2) % 4 == 0 and mc.stride(-1) == 1
    qc, kc, vc = (t.cuda().requires_grad_(True) for t in (q, k, v))
    out = ops.attention_extra(qc, kc, vc, mc, scale, False)
    _fwd_close(out, ref.detach(), f"mask view {kind} {dtype} L={L} D={D} pad={pad} off={off}")
    out.backward(g.cuda())
    for name, got, want in (("dq", qc.grad, qr.grad), ("dk", kc.grad, kr.grad), ("dv", vc.grad, vr.grad)):
        _grad_close(got, want, f"{name} mask view {kind} {dtype} L={L} D={D}")


@pytest.mark.parametrize("kind", ["bias_BHLL", "bias_1HLL", "bool_LL"])
@pytest.mark.parametrize("dtype,D", [(torch.bfloat16, 256), (torch.float16, 232), (torch.bfloat16, 128)])
@pytest.mark.parametrize("B,H,Hkv,L", [(1, 2, 1, 1284), (2, 2, 2, 448), (1, 4, 2, 64), (1, 1, 1, 129)])
def test_masked_backward_under_the_causal_mask_vs_oracle(kind, dtype, D, B, H, Hkv, L):
    """fa_bwd_ex with a mask AND the causal mask (the C ABI takes both; the override hands that pair to stock SDPA, as torch rejects it): at
    head_dim 256 a wave of dK/dV keeps the bias of its own 32 keys in a private ring of half-item pieces, and a wave whose keys lie past a query
    tile's rows - or past its first row half - skips those reads but must still request the pieces that follow; one-item and two-item
    sweeps, a ragged last key block, grouped K/V heads."""
    from flash_attention_amd._impl import ops
    from oracle import dense_attention_extra, frozen_randn

    gen = torch.Generator().manual_seed(SEED + 3 * L + H)
    q = (frozen_randn(SEED + L, (B, H, L, D)) * 0.6).to(dtype)
    k = (frozen_randn(SEED + L + 1, (B, Hkv, L, D)) * 0.6).to(dtype)
    v = (frozen_randn(SEED + L + 2, (B, Hkv, L, D)) * 0.6).to(dtype)
    g = frozen_randn(SEED + L + 3, (B, H, L, D)).to(dtype)
    mask = _masks(kind, B, H, L, dtype, gen)
    if mask.dtype == torch.bool:
        mask[..., :1] = True            # (every row admits key 0: under the causal mask a row must keep a key of its past)
    else:
        mask = mask.clone()
        mask[..., 0] = 0.0
    scale = D ** -0.5
    qr, kr, vr = (t.double().requires_grad_(True) for t in (q, k, v))
    ref, _ = dense_attention_extra(qr, kr, vr, scale, True, mask)
    ref.backward(g.double())
    qc, kc, vc = (t.cuda().requires_grad_(True) for t in (q, k, v))
    mc = mask.cuda().expand(B, H, L, L)
    mc = mc.view(torch.uint8) if mc.dtype == torch.bool else mc
    if mc.stride(-1) != 1:
        mc = mc.contiguous()
    out = ops.attention_extra(qc, kc, vc, mc, scale, True)
    _fwd_close(out, ref.detach(), f"masked causal forward {kind} {dtype} {B}x{H}/{Hkv}x{L}x{D}")
    out.backward(g.cuda())
    for name, got, want in (("dq", qc.grad, qr.grad), ("dk", kc.grad, kr.grad), ("dv", vc.grad, vr.grad)):
        _grad_close(got, want, f"{name} causal {kind} {dtype} {B}x{H}/{Hkv}x{L}x{D}")


@pytest.mark.parametrize("dtype,D", [(torch.bfloat16, 64), (torch.float16, 128), (torch.bfloat16, 192)])
def test_key_padding_with_a_batch_entry_that_keeps_no_key_and_tiles_that_keep_none(dtype, D):
    """A key-padding mask whose first batch entry keeps NO key (every tile of it is skipped by the staged-terms path: its rows are NaN with
    lse = -inf, as torch's), whose second keeps 70 keys of 700 (ten of its eleven 64-key tiles are skipped, one is ragged) and whose third keeps
    all: forward against the oracle, gradients of the live entries against autograd through it, the dead entry's gradients finite zeros."""
    from flash_attention_amd._impl import ops
    from oracle import dense_attention_extra, frozen_randn

    B, H, Hkv, L = 3, 4, 2, 700
    q = (frozen_randn(SEED + D, (B, H, L, D)) * 0.6).to(dtype)
    k = (frozen_randn(SEED + D + 1, (B, Hkv, L, D)) * 0.6).to(dtype)
    v = (frozen_randn(SEED + D + 2, (B, Hkv, L, D)) * 0.6).to(dtype)
    g = frozen_randn(SEED + D + 3, (B, H, L, D)).to(dtype)
    lens = torch.tensor([0, 70, L])
    mask = (torch.arange(L)[None, :] < lens[:, None]).view(B, 1, 1, L)
    scale = D ** -0.5
    qr, kr, vr = (t.double().requires_grad_(True) for t in (q, k, v))
    ref, _ = dense_attention_extra(qr, kr, vr, scale, False, mask)
    assert bool(torch.isnan(ref[0]).all()) and not bool(torch.isnan(ref[1:]).any())
    ref[1:].backward(g[1:].double())
    qc, kc, vc = (t.cuda().requires_grad_(True) for t in (q, k, v))
    out = ops.attention_extra(qc, kc, vc, mask.cuda().expand(B, H, L, L).view(torch.uint8), scale, False)
    _fwd_close(out, ref.detach(), f"key padding with an empty entry {dtype} D={D}")
    out[1:].backward(g[1:].cuda())
    for name, got, want in (("dq", qc.grad, qr.grad), ("dk", kc.grad, kr.grad), ("dv", vc.grad, vr.grad)):
        _grad_close(got[1:], want[1:], f"{name} key padding {dtype} D={D}")
        assert bool((got[0] == 0).all()), f"{name}: the entry without keys must get zero gradients"


@pytest.mark.parametrize("kind", ["bool", "bias"])
@pytest.mark.parametrize("causal", [False, True])
def test_mask_that_is_a_strided_view_equals_its_contiguous_copy(kind, causal):
    """The mask is passed by its strides (row stride > L, a batch stride of 0, a base that is only 4-element aligned): the kernels' DMA
    descriptors and lane offsets must address the view, not a dense [L, L] - bit-identical outputs and gradients to the contiguous copy,
    with and without the causal mask underneath."""
    from flash_attention_amd._impl import ops

    torch.manual_seed(5)
    B, H, L, D = 2, 2, 520, 64
    q, k, v = (torch.randn(B, H, L, D, device="cuda", dtype=torch.bfloat16) for _ in range(3))
    g = torch.randn(B, H, L, D, device="cuda", dtype=torch.bfloat16)
    if kind == "bool":
        big = (torch.rand(1, H, L + 3, L + 24, device="cuda") < 0.8)
        big[..., torch.arange(L) + 1, torch.arange(L) + 4] = True
        big = big.view(torch.uint8)
    else:
        big = torch.randn(1, H, L + 3, L + 24, device="cuda", dtype=torch.bfloat16)
    view = big[:, :, 1:1 + L, 4:4 + L].expand(B, H, L, L)
    assert view.stride() == (0, (L + 3) * (L + 24), L + 24, 1)
    res = []
    for m in (view, view.contiguous()):
        qc, kc, vc = (t.clone().requires_grad_(True) for t in (q, k, v))
        out = ops.attention_extra(qc, kc, vc, m, D ** -0.5, causal)
        out.backward(g)
        res.append((out.detach(), qc.grad, kc.grad, vc.grad))
    for a, b in zip(*res):
        assert torch.equal(a, b)


@pytest.mark.parametrize("kind", ["bool_LL", "bool_padding", "bool_BHLL_dead_rows", "bias_BHLL", "bias_1HLL", "bias_head_key"])
@pytest.mark.parametrize("dtype,D", [(torch.bfloat16, 192), (torch.float16, 256), (torch.bfloat16, 256), (torch.bfloat16, 160), (torch.float16, 232)])
@pytest.mark.parametrize("B,H,Hkv,L,causal", [(1, 2, 2, 200, False), (2, 2, 1, 520, False), (1, 2, 1, 1284, True)])
def test_masked_forward_at_head_dims_up_to_256_vs_oracle(kind, dtype, D, B, H, Hkv, L, causal):
    """fa_fwd_ex at head_dim 192 / 256 and at dims that run zero-extended at them (160, 232): the forward's mask kernels take every head dim
    the plain forward takes.  Boolean
    masks and biases through the LDS-DMA tile images (at 256 a bias through wave-private rings of half-tile pieces: two images do not fit beside 132 KB of K / V
    tiles), row-broadcast masks through the staged key terms, with and without the causal mask underneath, grouped K/V heads."""
    from flash_attention_amd._impl import ops
    from oracle import dense_attention_extra, frozen_randn

    gen = torch.Generator().manual_seed(SEED + L + D)
    q = (frozen_randn(SEED + L, (B, H, L, D)) * 0.5).to(dtype)
    k = (frozen_randn(SEED + L + 1, (B, Hkv, L, D)) * 0.5).to(dtype)
    v = (frozen_randn(SEED + L + 2, (B, Hkv, L, D)) * 0.5).to(dtype)
    mask = _masks(kind, B, H, L, dtype, gen)
    if causal and mask.dtype == torch.bool and mask.shape[-2] == L:
        # (under the causal mask a row whose admitted keys all lie in its future has no reference behaviour - torch rejects attn_mask with
        #  is_causal, the oracle's causal term is the reference's finite -10000, the kernels' is -inf: every row that admits a key admits key 0)
        mask[..., :1] |= mask.any(-1, keepdim=True)
    scale = D ** -0.5
    ref, _ = dense_attention_extra(q.double(), k.double(), v.double(), scale, causal, mask)
    mc = mask.cuda().expand(B, H, L, L)
    mc = mc.view(torch.uint8) if mc.dtype == torch.bool else mc
    if mc.stride(-1) != 1:
        mc = mc.contiguous()
    assert ops.supports_extra(q, needs_grad=False) and ops.supports_extra(q, needs_grad=True)
    with torch.no_grad():
        out = ops.attention_extra(q.cuda(), k.cuda(), v.cuda(), mc, scale, causal)
    _fwd_close(out, ref, f"masked forward {kind} {dtype} {B}x{H}/{Hkv}x{L}x{D} causal={causal}")


def test_override_serves_wide_head_masks_with_and_without_a_gradient(monkeypatch):
    """F.scaled_dot_product_attention with attn_mask at head_dim 256: this repo's mask kernels under no_grad and with a gradient wanted (the
    backward's dK/dV as two passes) - forward against stock's result on fp32 copies, gradients against stock's."""
    import flash_attention_amd.auto  # noqa: F401
    from flash_attention_amd._impl import ops

    calls = []
    real = ops.attention_extra
    monkeypatch.setattr(ops, "attention_extra", lambda *a, **kw: (calls.append(a[0].shape), real(*a, **kw))[1])
    torch.manual_seed(9)
    B, H, L, D = 2, 4, 384, 256
    q, k, v = (torch.randn(B, H, L, D, device="cuda", dtype=torch.bfloat16) for _ in range(3))
    keep = torch.ones(B, 1, 1, L, device="cuda", dtype=torch.bool)
    keep[0, ..., L - 77:] = False
    stock = F.scaled_dot_product_attention.__wrapped__
    with torch.no_grad():
        ours = F.scaled_dot_product_attention(q, k, v, attn_mask=keep)
        ref = stock(q.float(), k.float(), v.float(), attn_mask=keep)
    assert (ours.float() - ref).abs().max().item() <= 3e-2
    assert len(calls) == 1                          # this repo's mask kernel ran
    g = torch.randn_like(q)
    grads = []
    for fn, cast in ((F.scaled_dot_product_attention, lambda t: t), (stock, lambda t: t.float())):
        qg, kg, vg = (cast(t).clone().requires_grad_(True) for t in (q, k, v))
        fn(qg, kg, vg, attn_mask=keep).backward(cast(g))
        grads.append((qg.grad.float(), kg.grad.float(), vg.grad.float()))
    assert len(calls) == 2                          # ... and again for the call that wants a gradient
    for a, b in zip(*grads):
        assert (a - b).abs().max().item() <= 2e-2 * max(1.0, b.abs().max().item())
    # a bias that differs from row to row, head_dim 256, gradient wanted - stock SDPA's until round 6 (two 16 KB tile images do not fit beside the
    # backward's 128 KB of operand images): each wave of dK/dV now keeps the bias of its own 32 keys in a private ring of half-item pieces
    bias = torch.randn(1, H, L, L, device="cuda", dtype=torch.bfloat16)
    grads = []
    for fn, cast in ((F.scaled_dot_product_attention, lambda t: t), (stock, lambda t: t.float())):
        qg, kg, vg = (cast(t).clone().requires_grad_(True) for t in (q, k, v))
        fn(qg, kg, vg, attn_mask=cast(bias)).backward(cast(g))
        grads.append((qg.grad.float(), kg.grad.float(), vg.grad.float()))
    assert len(calls) == 3
    for a, b in zip(*grads):
        assert (a - b).abs().max().item() <= 2e-2 * max(1.0, b.abs().max().item())
    with torch.no_grad():
        F.scaled_dot_product_attention(q, k, v, attn_mask=bias)
    assert len(calls) == 4


@pytest.mark.parametrize("causal", [True, False])
@pytest.mark.parametrize("p", [0.1, 0.5])
@pytest.mark.parametrize("dtype,D", [(torch.bfloat16, 64), (torch.float16, 64), (torch.bfloat16, 128), (torch.float16, 192), (torch.bfloat16, 256)])
@pytest.mark.parametrize("B,H,Hkv,L", [(1, 2, 2, 130), (2, 4, 2, 384), (1, 3, 3, 1000)])
def test_dropout_forward_and_backward_vs_oracle(causal, p, dtype, D, B, H, Hkv, L):
    """dropout_p through fa_fwd_ex / fa_bwd_ex: the kernels' keep decisions are the numpy restatement's (same seed / offset /
    group / row / key), so out, logsumexp and all three gradients match the dense formula; with and without the causal mask."""
    from flash_attention_amd._impl import ops
    from oracle import dense_attention_extra, frozen_randn

    seed, offset = 0x1234_5678_9ABC_DEF0 + L, 4096 * L + 12
    q = (frozen_randn(SEED + L, (B, H, L, D)) * 0.6).to(dtype)
    k = (frozen_randn(SEED + L + 1, (B, Hkv, L, D)) * 0.6).to(dtype)
    v = (frozen_randn(SEED + L + 2, (B, Hkv, L, D)) * 0.6).to(dtype)
    g = frozen_randn(SEED + L + 3, (B, H, L, D)).to(dtype)
    scale = D ** -0.5
    qr, kr, vr = (t.double().requires_grad_(True) for t in (q, k, v))
    ref, _ = dense_attention_extra(qr, kr, vr, scale, causal, None, p, seed, offset)
    ref.backward(g.double())
    qc, kc, vc = (t.cuda().requires_grad_(True) for t in (q, k, v))
    out = ops.attention_extra(qc, kc, vc, None, scale, causal, p, seed, offset)
    _fwd_close(out, ref.detach(), f"dropout forward p={p} causal={causal} {dtype} {B}x{H}/{Hkv}x{L}x{D}")
    out.backward(g.cuda())
    for name, got, want in (("dq", qc.grad, qr.grad), ("dk", kc.grad, kr.grad), ("dv", vc.grad, vr.grad)):
        _grad_close(got, want, f"{name} dropout p={p} causal={causal} {dtype} {B}x{H}/{Hkv}x{L}x{D}")
    out2 = ops.attention_extra(qc.detach(), kc.detach(), vc.detach(), None, scale, causal, p, seed, offset + 1)
    assert not torch.equal(out2, out.detach()), "another offset must give another mask"
    out3 = ops.attention_extra(qc.detach(), kc.detach(), vc.detach(), None, scale, causal, p, seed, offset)
    assert torch.equal(out3, out.detach()), "the same (seed, offset) must reproduce the result bit for bit"


def test_mask_and_dropout_together_with_lse():
    """Both at once, plus the causal mask underneath, and the logsumexp stash (which ignores dropout: it is the softmax's)."""
    from flash_attention_amd._impl import _native
    from oracle import dense_attention_extra, frozen_randn

    B, H, L, D, dtype, p = 2, 2, 320, 64, torch.bfloat16, 0.2
    gen = torch.Generator().manual_seed(5)
    q, k, v = ((frozen_randn(SEED + i, (B, H, L, D)) * 0.6).to(dtype) for i in range(3))
    mask = _masks("bias_BHLL", B, H, L, dtype, gen)
    ref, ref_lse = dense_attention_extra(q, k, v, D ** -0.5, True, mask, p, 11, 22)
    out = torch.empty(B, H, L, D, dtype=dtype, device="cuda")
    lse = torch.empty(B * H, L, dtype=torch.float32, device="cuda")
    _native.fwd_ex(q.cuda(), k.cuda(), v.cuda(), out, lse, L, D, D ** -0.5, True, mask.cuda(), p, 11, 22)
    _fwd_close(out, ref, "mask + dropout + causal")
    torch.testing.assert_close(lse.view(B, H, L).cpu().double(), ref_lse, atol=LSE_ATOL, rtol=1e-4)


def test_sdpa_override_serves_attn_mask_and_dropout():
    """`import flash_attention_amd.auto`: F.scaled_dot_product_attention(attn_mask=..., dropout_p=...) no longer raises on the GPU
    (the reference raises NotImplementedError, sdpa.py:21-28): masks agree with stock SDPA and the oracle, dropout is repeatable
    under torch.manual_seed, keeps ~(1 - p) of the weights, and combinations the kernels do not take go to stock SDPA."""
    import flash_attention_amd.auto  # noqa: F401
    from flash_attention_amd._impl import ops
    from oracle import dense_attention_extra, frozen_randn

    patched = F.scaled_dot_product_attention
    stock = patched.__wrapped__
    B, H, L, D, dtype = 2, 4, 256, 64, torch.bfloat16
    q, k, v = ((frozen_randn(SEED + 50 + i, (B, H, L, D)) * 0.6).to(dtype).cuda() for i in range(3))
    gen = torch.Generator().manual_seed(9)
    calls = []
    real = ops.attention_extra
    ops.attention_extra = lambda *a, **kw: (calls.append(1), real(*a, **kw))[1]
    try:
        for kind in ("bool_LL", "bool_padding", "bias_BHLL", "bias_row"):
            mask = _masks(kind, B, H, L, dtype, gen)
            ref, _ = dense_attention_extra(q.cpu(), k.cpu(), v.cpu(), D ** -0.5, False, mask)
            n = len(calls)
            _fwd_close(patched(q, k, v, attn_mask=mask.cuda()), ref, f"override attn_mask {kind}")
            assert len(calls) == n + 1, "the mask kernels must have served this call"
            _fwd_close(stock(q, k, v, attn_mask=mask.cuda()), ref, f"stock attn_mask {kind}")      # same yardstick
        # a float32 mask on bf16 operands is converted; a 3-d [B*H, L, D] call is served too
        m32 = _masks("bias_1HLL", B, H, L, torch.float32, gen)
        ref, _ = dense_attention_extra(q.cpu(), k.cpu(), v.cpu(), D ** -0.5, False, m32.to(dtype))
        _fwd_close(patched(q, k, v, attn_mask=m32.cuda()), ref, "override fp32 bias")
        q3, k3, v3 = (t.reshape(B * H, L, D) for t in (q, k, v))
        ref3, _ = dense_attention_extra(q3.cpu()[:, None], k3.cpu()[:, None], v3.cpu()[:, None], D ** -0.5, False, _masks("bool_LL", 1, 1, L, dtype, torch.Generator().manual_seed(3)))
        _fwd_close(patched(q3, k3, v3, attn_mask=_masks("bool_LL", 1, 1, L, dtype, torch.Generator().manual_seed(3)).cuda())[:, None], ref3, "override 3-d")
        # dropout: repeatable under manual_seed, different on the next call, ~(1 - p) kept (v = ones: out = sum of kept weights / (1 - p))
        n = len(calls)
        torch.manual_seed(123)
        a = patched(q, k, v, dropout_p=0.3, is_causal=True)
        b = patched(q, k, v, dropout_p=0.3, is_causal=True)
        torch.manual_seed(123)
        c = patched(q, k, v, dropout_p=0.3, is_causal=True)
        assert len(calls) == n + 3 and torch.equal(a, c) and not torch.equal(a, b)
        ones = torch.ones_like(v)
        kept = patched(q, k, ones, dropout_p=0.3)[..., 0].float().mean().item()      # E = 1
        assert abs(kept - 1.0) < 0.02, kept
        # training through the override with a mask and dropout
        qg = q.clone().requires_grad_(True)
        patched(qg, k, v, attn_mask=_masks("bool_padding", B, H, L, dtype, gen).cuda(), dropout_p=0.1).float().square().sum().backward()
        assert torch.isfinite(qg.grad).all() and qg.grad.abs().max() > 0
        # head_dim 80 IS the kernels' business since round 4 (a 16-bit multiple of 8 below 128 runs zero-extended inside the MFMA kernels)
        n = len(calls)
        q80 = torch.randn(1, 2, 64, 80, device="cuda", dtype=dtype)
        assert torch.isfinite(patched(q80, q80, q80, dropout_p=0.1)).all() and len(calls) == n + 1
        # not the kernels' business: fp32 operands, head_dim 84, a mask that wants a gradient -> stock SDPA (no raise)
        n = len(calls)
        patched(q.float(), k.float(), v.float(), attn_mask=_masks("bool_LL", B, H, L, dtype, gen).cuda())
        q84 = torch.randn(1, 2, 64, 84, device="cuda", dtype=dtype)
        patched(q84, q84, q84, dropout_p=0.1)
        bias = torch.zeros(L, L, device="cuda", dtype=dtype, requires_grad=True)
        assert patched(q, k, v, attn_mask=bias).grad_fn is not None and len(calls) == n      # (stock SDPA's graph: it owns the bias gradient)
        # attn_mask together with is_causal is stock SDPA's business (its documentation promises an error; this build computes
        # something instead, not even repeatably): the override hands the call over unchanged and the mask kernels stay out of it
        try:
            patched(q, k, v, attn_mask=torch.ones(L, L, dtype=torch.bool, device="cuda"), is_causal=True)
        except RuntimeError:
            pass
        assert len(calls) == n
    finally:
        ops.attention_extra = real
    # off the GPU the reference's guards still stand
    with pytest.raises(NotImplementedError, match="attn_mask"):
        patched(q.cpu(), k.cpu(), v.cpu(), attn_mask=torch.ones(L, L, dtype=torch.bool))
    with pytest.raises(NotImplementedError, match="dropout_p"):
        patched(q.cpu(), k.cpu(), v.cpu(), dropout_p=0.1, is_causal=True)


@pytest.mark.parametrize("dtype,D,L", [(torch.float32, 64, 200), (torch.float32, 40, 130), (torch.bfloat16, 80, 190), (torch.float16, 64, 97)])
def test_generic_kernels_serve_mask_and_dropout_for_everything_else(dtype, D, L):
    """fa_fwd_ex / fa_bwd_ex on operands the MFMA mask kernels do not take - fp32, head dims other than 64 / 128, (f16, 64) with
    unaligned rows - run the generic fp32-arithmetic kernels with the same mask terms and the same dropout decisions: out, logsumexp
    and gradients against the oracle.  (The SDPA override hands such calls to stock SDPA for speed; the C ABI serves them.)"""
    from flash_attention_amd._impl import _native
    from oracle import dense_attention_extra, frozen_randn

    B, H, Hkv, p = 2, 4, 2, 0.2
    gen = torch.Generator().manual_seed(L)
    pad = 1 if (dtype == torch.float16 and D == 64) else 0            # odd row pitch -> unaligned rows -> generic kernels
    mk = lambda seed, h: (frozen_randn(SEED + seed, (B, h, L, D + pad)) * 0.6).to(dtype)[..., :D]  # noqa: E731
    q, k, v, g = mk(1, H), mk(2, Hkv), mk(3, Hkv), mk(4, H)
    mask = _masks("bias_BHLL", B, H, L, dtype, gen)
    scale = D ** -0.5
    qr, kr, vr = (t.double().requires_grad_(True) for t in (q, k, v))
    ref, ref_lse = dense_attention_extra(qr, kr, vr, scale, True, mask, p, 5, 6)
    ref.backward(g.double())
    qc, kc, vc, gc, mc = (t.cuda() for t in (q, k, v, g, mask))
    out = torch.empty(B, H, L, D, dtype=dtype, device="cuda")
    lse = torch.empty(B * H, L, dtype=torch.float32, device="cuda")
    _native.fwd_ex(qc, kc, vc, out, lse, L, D, scale, True, mc, p, 5, 6)
    o = out.cpu().double()
    tol = 1e-4 if dtype == torch.float32 else 1e-2
    assert bool(((o - ref.detach()).abs() <= tol + tol * ref.detach().abs()).all()), (o - ref.detach()).abs().max().item()
    torch.testing.assert_close(lse.view(B, H, L).cpu().double(), ref_lse.detach(), atol=1e-4 if dtype == torch.float32 else LSE_ATOL, rtol=1e-4)
    dq, dk, dv = torch.empty_like(qc.contiguous()), torch.empty_like(kc.contiguous()), torch.empty_like(vc.contiguous())
    _native.bwd_ex(qc, kc, vc, out, gc, lse, dq, dk, dv, L, D, scale, True, mc, p, 5, 6)
    for name, got, want in (("dq", dq, qr.grad), ("dk", dk, kr.grad), ("dv", dv, vr.grad)):
        if dtype == torch.float32:
            err = (got.cpu().double() - want).abs().max().item()
            assert err <= 1e-4 * max(1.0, want.abs().max().item()), (name, err)
        else:
            _grad_close(got, want, f"{name} generic {dtype} D={D}")


@pytest.mark.parametrize("dtype,D", [(torch.bfloat16, 64), (torch.float16, 64), (torch.bfloat16, 128)])
@pytest.mark.parametrize("mask_shape", ["B11L", "BHLL"])
def test_finfo_min_padding_mask_with_fully_masked_rows(dtype, D, mask_shape):
    """The common padding convention: an additive mask filled with torch.finfo(dtype).min (NOT -inf).  bias / softmax_scale overflows
    fp32 for it (-3.39e38 * 8), which used to turn a row whose keys are all padded into l = 0, O = NaN where the op's definition
    (softmax of equal logits) gives the uniform average of V (round-3 advisor finding; fa_extras.h bias_term holds the converted
    term at -2^99).  Left-padded batches
    have such rows (a padded query position sees only padded keys under a padding + causal mask).  Forward against stock SDPA on
    every row and gradients with dO = 0 on the fully masked rows (a padded position carries no loss), against the op's definition
    evaluated in fp64 - finite everywhere."""
    import flash_attention_amd.auto  # noqa: F401

    B, H, L = 2, 3, 256
    gen = torch.Generator().manual_seed(SEED + D)
    q, k, v = ((torch.randn(B, H, L, D, generator=gen) * 0.7).to(dtype).cuda().requires_grad_(True) for _ in range(3))
    neg = torch.finfo(dtype).min
    pad = torch.tensor([37, 0])                                  # batch entry 0 is left-padded by 37 positions
    key_is_pad = torch.arange(L)[None, :] < pad[:, None]         # [B, L]
    if mask_shape == "B11L":      # key padding only; entry 1 of a second mask pads EVERY key (all rows fully masked)
        mask = torch.zeros(B, 1, 1, L)
        mask[key_is_pad[:, None, None, :].expand(B, 1, 1, L)] = neg
        mask[1] = neg
        dead_rows = torch.zeros(B, L, dtype=torch.bool)
        dead_rows[1] = True
    else:                         # padding + causal as one additive [B, H, L, L] mask: rows 0..36 of entry 0 see padded keys only
        causal = torch.triu(torch.ones(L, L, dtype=torch.bool), 1)
        m2 = causal[None] | key_is_pad[:, None, :]
        mask = torch.zeros(B, 1, L, L).masked_fill(m2[:, None], neg).expand(B, H, L, L).contiguous()
        dead_rows = key_is_pad.clone()
    mask = mask.to(dtype).cuda()
    dout = torch.randn(B, H, L, D, generator=gen).to(dtype)
    dout[dead_rows[:, None, :].expand(B, H, L)] = 0
    dout = dout.cuda()

    out = F.scaled_dot_product_attention(q, k, v, attn_mask=mask)
    assert torch.isfinite(out).all(), "a finfo.min mask must not produce NaN / inf rows"
    gq, gk, gv = torch.autograd.grad(out, (q, k, v), dout)
    # the yardstick: torch's definition of the op in fp64 on the same rounded operands and the same (finite) mask values.  (Stock
    # SDPA's fused backend is no yardstick here: on the test box it returned other values for fully masked rows and NaN gradients
    # in f16; its backends disagree with each other about such rows.)
    from oracle import dense_attention_extra

    qr, kr, vr = (t.detach().cpu().double().requires_grad_(True) for t in (q, k, v))
    ref, _ = dense_attention_extra(qr, kr, vr, D ** -0.5, False, mask.cpu().double())
    rq, rk, rv = torch.autograd.grad(ref, (qr, kr, vr), dout.cpu().double())
    live = ~dead_rows[:, None, :, None].expand(B, H, L, D)
    _fwd_close(out.detach().cpu()[live], ref.detach()[live], f"finfo.min mask {mask_shape} {dtype} D={D} forward, live rows")
    # a fully masked row.  bf16: finfo.min / scale is held at -2^99, which absorbs the raw score exactly like the fp64 sum of the
    # definition does (-3.39e38 + s): the uniform average of V.  f16: -65504 / scale stays an ordinary finite bias, equal on every
    # key of the row, so by shift invariance the row is the plain softmax of its scores - again what the definition gives.
    dead = ~live
    if dead.any():
        _fwd_close(out.detach().cpu()[dead], ref.detach()[dead], f"finfo.min mask {mask_shape} {dtype} D={D} forward, fully masked rows")
        if dtype == torch.bfloat16:
            uniform = v.detach().cpu().double().mean(dim=2, keepdim=True).expand(B, H, L, D)
            torch.testing.assert_close(out.detach().cpu().double()[dead], uniform[dead], atol=2e-2, rtol=2e-2)
    for got, want, name in ((gq, rq, "dq"), (gk, rk, "dk"), (gv, rv, "dv")):
        assert torch.isfinite(got).all(), f"{name}: non-finite gradient under a finfo.min mask"
        sel = live if name == "dq" else torch.ones_like(live)
        _grad_close(got.detach().cpu()[sel], want[sel], f"finfo.min mask {mask_shape} {dtype} D={D} {name}")


def test_dropout_under_graph_capture_draws_a_fresh_mask_on_every_replay():
    """Round-3 verdict, item 6: dropout's (seed, offset) used to be read on the host, so a launch captured in a HIP graph replayed ONE
    mask forever.  Under capture the kernels now read the two words from device memory (ABI 5 `dropout_state`): the captured call
    snapshots the device-resident generator state and advances it - captured device operations - so every replay draws fresh
    decisions, and the backward captured in the same graph regenerates exactly its forward's.  Forward + backward captured in
    torch.cuda.graph, replayed twice: two different masks, each equal to oracle/dropout.py's for the state that replay read."""
    import flash_attention_amd
    import flash_attention_amd.auto  # noqa: F401
    from flash_attention_amd._impl import ops
    from oracle import dense_attention_extra, frozen_randn

    B, H, L, D, dtype, p = 2, 3, 256, 64, torch.bfloat16, 0.25
    dev = torch.device("cuda", 0)
    q, k, v = ((frozen_randn(SEED + 31 + i, (B, H, L, D)) * 0.6).to(dtype).to(dev).requires_grad_(True) for i in range(3))
    dout = frozen_randn(SEED + 35, (B, H, L, D)).to(dtype).to(dev)
    torch.manual_seed(1234)
    state = flash_attention_amd.init_graph_dropout_state(dev)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                       # the usual warm-up before a capture
        o = F.scaled_dot_product_attention(q, k, v, dropout_p=p, is_causal=True)
        torch.autograd.grad(o, (q, k, v), dout)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = F.scaled_dot_product_attention(q, k, v, dropout_p=p, is_causal=True)
        gq, gk, gv = torch.autograd.grad(out, (q, k, v), dout)
    results = []
    for _ in range(2):
        seed, offset = (int(x) for x in state.tolist())          # what this replay's kernels will read
        graph.replay()
        torch.cuda.synchronize()
        results.append((seed, offset, out.clone(), gq.clone(), gk.clone(), gv.clone()))
    assert results[0][1] != results[1][1], "the captured increment must advance the device-resident offset on every replay"
    assert not torch.equal(results[0][2], results[1][2]), "two replays of a captured dropout launch must draw two masks"
    for seed, offset, o_, gq_, gk_, gv_ in results:
        qr, kr, vr = (t.detach().cpu().double().requires_grad_(True) for t in (q, k, v))
        ref, _ = dense_attention_extra(qr, kr, vr, D ** -0.5, True, None, p, seed, offset)
        ref.backward(dout.cpu().double())
        _fwd_close(o_, ref.detach(), f"captured dropout forward (seed {seed}, offset {offset})")
        for name, got, want in (("dq", gq_, qr.grad), ("dk", gk_, kr.grad), ("dv", gv_, vr.grad)):
            _grad_close(got, want, f"captured dropout {name} (offset {offset})")
    # an eager call afterwards still works and keeps taking its randomness from torch's host generator
    eager = F.scaled_dot_product_attention(q.detach(), k.detach(), v.detach(), dropout_p=p, is_causal=True)
    assert torch.isfinite(eager).all()
    assert ops._GRAPH_DROPOUT_STATE[0] is state
    # re-seeding AFTER the capture (round-4 advisor): the state is overwritten in place - the same tensor, the same address the graph holds -
    # so the next replay reads the new seed, and a re-seed to a known value reproduces that replay's mask
    ptr = state.data_ptr()
    again = flash_attention_amd.init_graph_dropout_state(dev, seed=777)
    assert again is state and state.data_ptr() == ptr
    seed_a, off_a = (int(x) for x in state.tolist())
    assert seed_a == (777 ^ ops._GRAPH_STREAM_SALT) & (2 ** 63 - 1) and seed_a != results[0][0]
    graph.replay()
    torch.cuda.synchronize()
    first = out.clone()
    qr, kr, vr = (t.detach().cpu().double() for t in (q, k, v))
    ref, _ = dense_attention_extra(qr, kr, vr, D ** -0.5, True, None, p, seed_a, off_a)
    _fwd_close(first, ref, "replay after an in-place re-seed")
    flash_attention_amd.init_graph_dropout_state(dev, seed=777)
    state[1] = off_a                                    # (same seed, same offset: the same decisions)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, first)
    # the device-resident stream is not the eager stream: same torch seed, different decisions
    torch.manual_seed(1234)
    assert int(flash_attention_amd.init_graph_dropout_state(dev)[0]) != 1234


def test_attn_mask_stays_on_the_hip_path_under_torch_compile():
    """torch.compile(fullgraph=True) of a block that passes attn_mask: the override traces through the
    `flash_attention_amd::fwd_ex` custom op (fake implementation for shape inference, registered autograd) instead of handing the
    call to stock SDPA - forward and gradients equal the eager HIP path bit for bit."""
    import flash_attention_amd.auto  # noqa: F401
    from flash_attention_amd._impl import _native

    B, H, L, D, dtype = 2, 4, 256, 64, torch.bfloat16
    gen = torch.Generator(device="cuda").manual_seed(5)
    q, k, v = (torch.randn(B, H, L, D, device="cuda", generator=gen).to(dtype).requires_grad_(True) for _ in range(3))
    lens = torch.tensor([200, 256], device="cuda")
    mask = (torch.arange(L, device="cuda")[None, :] < lens[:, None]).view(B, 1, 1, L)
    calls = []
    real, real_eager = _native.fwd_ex, _native.fwd_ex_bhld          # (the custom op's entry point, and the eager path's lean form of it)
    _native.fwd_ex = lambda *a, **kw: (calls.append(1), real(*a, **kw))[1]
    _native.fwd_ex_bhld = lambda *a, **kw: (calls.append(1), real_eager(*a, **kw))[1]
    try:
        def block(q, k, v, mask):
            return F.scaled_dot_product_attention(q * 1.0, k, v, attn_mask=mask) + 1.0

        eager = block(q, k, v, mask)
        ge = torch.autograd.grad(eager, (q, k, v), torch.ones_like(eager))
        n_eager = len(calls)
        compiled = torch.compile(block, fullgraph=True)
        out = compiled(q, k, v, mask)
        gc = torch.autograd.grad(out, (q, k, v), torch.ones_like(out))
    finally:
        _native.fwd_ex, _native.fwd_ex_bhld = real, real_eager
    assert n_eager == 1 and len(calls) == 2, "the compiled block must call fa_fwd_ex (not stock SDPA)"
    assert torch.equal(out, eager)
    for a, b in zip(gc, ge):
        assert torch.equal(a, b)


def test_dropout_stays_on_the_hip_path_under_torch_compile():
    """Round-5 verdict, item 6: the last mode that silently left the HIP path in a compiled training step.  torch.compile(fullgraph=True) of a
    block with dropout_p > 0: the override reads (seed, offset) from the device-resident generator state - a graph input whose snapshot and
    advance are traced device operations - and calls `flash_attention_amd::fwd_ex` / `bwd_ex`; two calls of the compiled block draw two masks,
    and each call's forward AND gradients equal oracle/dropout.py's decisions for the state that call read."""
    import flash_attention_amd
    import flash_attention_amd.auto  # noqa: F401
    from flash_attention_amd._impl import _native
    from flash_attention_amd._impl import scaled_dot_product_attention as sdpa_mod
    from oracle import dense_attention_extra, frozen_randn

    B, H, L, D, dtype, p = 2, 3, 256, 64, torch.bfloat16, 0.2
    dev = torch.device("cuda", 0)
    q, k, v = ((frozen_randn(SEED + 51 + i, (B, H, L, D)) * 0.6).to(dtype).to(dev).requires_grad_(True) for i in range(3))
    dout = frozen_randn(SEED + 57, (B, H, L, D)).to(dtype).to(dev)
    torch.manual_seed(4321)
    state = flash_attention_amd.init_graph_dropout_state(dev)
    calls = {"fwd": 0, "bwd": 0}
    real_f, real_b = _native.fwd_ex, _native.bwd_ex
    _native.fwd_ex = lambda *a, **kw: (calls.__setitem__("fwd", calls["fwd"] + 1), real_f(*a, **kw))[1]
    _native.bwd_ex = lambda *a, **kw: (calls.__setitem__("bwd", calls["bwd"] + 1), real_b(*a, **kw))[1]
    handoffs = set(sdpa_mod._HANDOFF_REASONS)
    try:
        def block(q, k, v):
            return F.scaled_dot_product_attention(q * 1.0, k, v, dropout_p=p, is_causal=True)

        compiled = torch.compile(block, fullgraph=True)
        results = []
        for _ in range(2):
            seed, offset = (int(x) for x in state.tolist())          # what this call's kernels will read
            out = compiled(q, k, v)
            gq, gk, gv = torch.autograd.grad(out, (q, k, v), dout)
            torch.cuda.synchronize()
            results.append((seed, offset, out.detach().clone(), gq.clone(), gk.clone(), gv.clone()))
    finally:
        _native.fwd_ex, _native.bwd_ex = real_f, real_b
    assert calls == {"fwd": 2, "bwd": 2}, f"the compiled block must call fa_fwd_ex / fa_bwd_ex, not stock SDPA: {calls}"
    assert set(sdpa_mod._HANDOFF_REASONS) == handoffs, "no hand-off to stock SDPA"
    assert results[0][1] != results[1][1], "the traced in-place add must advance the device-resident offset on every call"
    assert not torch.equal(results[0][2], results[1][2]), "two calls of a compiled dropout block must draw two masks"
    for seed, offset, o_, gq_, gk_, gv_ in results:
        qr, kr, vr = (t.detach().cpu().double().requires_grad_(True) for t in (q, k, v))
        ref, _ = dense_attention_extra(qr, kr, vr, D ** -0.5, True, None, p, seed, offset)
        ref.backward(dout.cpu().double())
        _fwd_close(o_, ref.detach(), f"compiled dropout forward (seed {seed}, offset {offset})")
        for name, got, want in (("dq", gq_, qr.grad), ("dk", gk_, kr.grad), ("dv", gv_, vr.grad)):
            _grad_close(got, want, f"compiled dropout {name} (offset {offset})")
