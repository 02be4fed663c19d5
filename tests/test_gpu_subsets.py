"""Group counts that are not a multiple of 8 (GPU, -m gpu): the forward kernels map groups to the 8 XCDs by index mod 8, and where that
leaves XCDs idle or unevenly loaded the launch is laid out over (group, q-tile subset) pairs (fa_plan_t.xs, csrc/fa_capi.cpp subsets_for).
Every forward kernel that takes the map is checked against the fp64 oracle and - bit for bit where the kernel is the same - against the
whole-group map (`subsets` = 0).  The reference has no counterpart: Poplar places its tiles itself (FA.cpp:229-328); the contract checked is
the op's, fa.py:48-53.
"""

import pytest
import torch

pytestmark = pytest.mark.gpu

SEED = 1123581321


def _mods():
    from flash_attention_amd._impl import _native

    assert _native.library_path() is not None
    return _native


def _run(native, qkv, causal, scale=1.0):
    _, G, L, D = qkv.shape
    out = torch.empty(G, L, D, device="cuda", dtype=qkv.dtype)
    lse = torch.empty(G, L, device="cuda", dtype=torch.float32)
    st = (L * D, 0, D)
    native.fwd_strided(qkv[0], qkv[1], qkv[2], out, lse, G, L, D, 1, st, st, st, st, 1, 1, scale, causal)
    torch.cuda.synchronize()
    return out, lse


def _oracle_rows(qkv, rows, causal, scale=1.0):
    """fp64 attention of the chosen rows of every group (the oracle's arithmetic, oracle/dense_attention.py, on a row subset)"""
    q, k, v = (t.double() for t in qkv)
    L = q.shape[1]
    s = torch.einsum("grd,gkd->grk", q[:, rows], k) * scale
    if causal:
        s = s.masked_fill((torch.arange(L, device=q.device)[None, :] > torch.tensor(rows, device=q.device)[:, None])[None], float("-inf"))
    return torch.softmax(s, -1) @ v, torch.logsumexp(s, -1)


CASES = [
    # G, L, D, kernel, expected xs
    (2, 8192, 64, None, 4),        # six XCDs idle without subsets
    (1, 4096, 64, None, 8),
    (12, 1024, 64, None, 2),       # GPT-2 small at batch 1
    (12, 2048, 64, "pipe", 2),
    (3, 4096, 128, "pipe16", 8),
    (5, 2048, 128, "kh", 8),
    (6, 2048, 64, "dma", 4),
    (25, 2048, 64, None, 4),       # GPT-2 XL at batch 1: the smallest subset count that evens the XCDs out
    (2, 500, 64, "dma", 1),        # fewer 256-row q tiles than subsets: whole groups
    (7, 4096, 128, None, 1),       # 8 subsets would put 14 MB of K/V on every XCD: whole groups
    (100, 256, 64, None, 1),       # 4 % waste: whole groups
]


@pytest.mark.parametrize("causal", [True, False])
@pytest.mark.parametrize("G,L,D,kernel,xs", CASES)
def test_subsets_vs_oracle_and_whole_group_map(G, L, D, kernel, xs, causal):
    native = _mods()
    from oracle import frozen_randn

    dtype = torch.bfloat16
    qkv = (frozen_randn(SEED + G + L, (3, G, L, D)) * D ** -0.25).to(dtype).cuda()
    with native.debug_options(kernel=kernel):
        plan = native.plan(G, L, D, dtype, 1, 1, causal=causal)
        assert plan.xs == xs, (plan.xs, xs)
        out, lse = _run(native, qkv, causal)
        kern = plan.kernel
    with native.debug_options(kernel=kernel, subsets=0):
        whole = native.plan(G, L, D, dtype, 1, 1, causal=causal)
        assert whole.xs == 1
        out0, lse0 = _run(native, qkv, causal)
    if (whole.kernel, whole.waves, whole.block_q) == (kern, plan.waves, plan.block_q):
        assert torch.equal(out, out0) and torch.equal(lse, lse0)      # the same program on every row, in another order
    rows = sorted({0, 1, 63, 64, L // 2, L - 129 if L > 129 else 0, L - 2, L - 1} | set(range(127, L, max(128, L // 8 // 128 * 128))))
    ref, ref_lse = _oracle_rows(qkv, rows, causal)
    for o, l in ((out, lse), (out0, lse0)):
        err = (o[:, rows].double() - ref).abs()
        assert torch.isfinite(o.float()).all()
        assert not bool((err > 1e-2 + 1e-2 * ref.abs()).any()), err.max().item()
        torch.testing.assert_close(l[:, rows].double(), ref_lse, atol=6e-3, rtol=1e-4)


@pytest.mark.parametrize("dtype", [torch.float16, torch.float32])
def test_subsets_other_dtypes(dtype):
    native = _mods()
    from oracle import frozen_randn

    G, L, D = 3, 2048, 64
    qkv = (frozen_randn(SEED, (3, G, L, D)) * D ** -0.25).to(dtype).cuda()
    plan = native.plan(G, L, D, dtype, 1, 1)
    assert plan.xs == 8
    out, lse = _run(native, qkv, True)
    rows = [0, 1, 127, 128, 1000, L - 1]
    ref, ref_lse = _oracle_rows(qkv, rows, True)
    tol = 1e-4 if dtype == torch.float32 else 1e-2
    assert (out[:, rows].double() - ref).abs().max().item() <= tol
    torch.testing.assert_close(lse[:, rows].double(), ref_lse, atol=6e-3 if dtype != torch.float32 else 1e-4, rtol=1e-4)


def test_subsets_through_the_public_op_with_a_mask_and_the_backward():
    """G = 12 (batch 1 x 12 heads): the override's forward runs on subsets, the mask kernel takes the same map, and the backward (whole
    groups at 8 groups and more: fa_capi.cpp bwd_impl) consumes the forward's logsumexp - gradients against autograd through the dense fp64 expression."""
    native = _mods()
    import flash_attention_amd.auto  # noqa: F401  (patches F.scaled_dot_product_attention)
    import torch.nn.functional as F

    torch.manual_seed(3)
    B, H, L, D = 1, 12, 1024, 64
    q, k, v = (torch.randn(B, H, L, D, device="cuda", dtype=torch.bfloat16, requires_grad=True) for _ in range(3))
    assert native.plan(B * H, L, D, torch.bfloat16, 4, 4, softmax_scale=D ** -0.5).xs == 2
    out = F.scaled_dot_product_attention(q, k, v, is_causal=True)
    g = torch.randn_like(out)
    out.backward(g)
    qd, kd, vd = (t.detach().double().requires_grad_(True) for t in (q, k, v))
    s = (qd @ kd.transpose(-1, -2)) * D ** -0.5
    s = s.masked_fill(torch.triu(torch.ones(L, L, device="cuda", dtype=torch.bool), 1), float("-inf"))
    ref = torch.softmax(s, -1) @ vd
    ref.backward(g.double())
    assert (out.double() - ref).abs().max().item() <= 3e-2
    for a, b in ((q.grad, qd.grad), (k.grad, kd.grad), (v.grad, vd.grad)):
        assert (a.double() - b).abs().max().item() <= 6e-2 * max(1.0, b.abs().max().item())
    # key-padding mask through the same map
    keep = torch.ones(B, 1, 1, L, device="cuda", dtype=torch.bool)
    keep[..., L - 100:] = False
    om = F.scaled_dot_product_attention(q.detach(), k.detach(), v.detach(), attn_mask=keep)
    sm = ((qd.detach() @ kd.detach().transpose(-1, -2)) * D ** -0.5).masked_fill(~keep, float("-inf"))
    assert (om.double() - torch.softmax(sm, -1) @ vd.detach()).abs().max().item() <= 3e-2


@pytest.mark.parametrize("B,H,Hkv,L,D", [(1, 2, 2, 2048, 64), (1, 6, 2, 1024, 128), (1, 3, 3, 2048, 64)])
def test_backward_block_subsets_vs_autograd_fp64_and_whole_group_map(B, H, Hkv, L, D):
    """The dQ grid (q-tile subsets) and the dK/dV grid (key-block subsets, over the K/V groups: grouped-query heads included) of a group count
    that leaves XCDs idle: gradients against autograd through the dense fp64 expression, and bit for bit against the whole-group map."""
    native = _mods()
    import flash_attention_amd.auto  # noqa: F401
    import torch.nn.functional as F

    torch.manual_seed(11)
    q = torch.randn(B, H, L, D, device="cuda", dtype=torch.bfloat16, requires_grad=True)
    k, v = (torch.randn(B, Hkv, L, D, device="cuda", dtype=torch.bfloat16, requires_grad=True) for _ in range(2))
    g = torch.randn(B, H, L, D, device="cuda", dtype=torch.bfloat16)
    grads = {}
    for sub in (-1, 0):
        with native.debug_options(subsets=sub):
            out = F.scaled_dot_product_attention(q, k, v, is_causal=True, enable_gqa=H != Hkv)
            grads[sub] = (out.detach().clone(),) + torch.autograd.grad(out, (q, k, v), g)
    for a, b in zip(grads[-1][1:], grads[0][1:]):
        assert torch.equal(a, b)                   # the same program on every block, in another order
    qd, kd, vd = (t.detach().double().requires_grad_(True) for t in (q, k, v))
    ke, ve = (t.repeat_interleave(H // Hkv, dim=1) for t in (kd, vd))
    s = (qd @ ke.transpose(-1, -2)) * D ** -0.5
    s = s.masked_fill(torch.triu(torch.ones(L, L, device="cuda", dtype=torch.bool), 1), float("-inf"))
    ref = torch.softmax(s, -1) @ ve
    ref.backward(g.double())
    assert (grads[-1][0].double() - ref).abs().max().item() <= 3e-2
    for a, b in zip(grads[-1][1:], (qd.grad, kd.grad, vd.grad)):
        assert (a.double() - b).abs().max().item() <= 6e-2 * max(1.0, b.abs().max().item())


@pytest.mark.parametrize("B,H,Hkv,L,D,dtype,bsplit,causal", [
    (1, 1, 1, 2048, 64, torch.bfloat16, 3, True),       # the rule's own case: one group, 16 blocks per grid
    (1, 2, 2, 1500, 64, torch.float16, 3, True),        # ragged L: the groups' item counts differ by one
    (1, 4, 2, 1024, 64, torch.bfloat16, 2, True),       # grouped-query heads: the dK/dV stream is (head, query tile) items
    (1, 2, 2, 2048, 128, torch.bfloat16, 2, True),      # head_dim 128: dQ with two kv split groups, dK/dV with two row-half groups on shared tiles
    (1, 6, 2, 1100, 128, torch.float16, 2, True),       # the same with grouped-query heads and a ragged length
    (2, 1, 1, 4096, 64, torch.bfloat16, -1, True),      # whatever the library picks by itself
    (1, 2, 2, 1024, 64, torch.bfloat16, 3, False),      # no causal mask: every block sweeps everything
    (1, 12, 12, 4096, 64, torch.bfloat16, -1, True),    # the rule's upper range: 384 workgroups per grid at L >= 2048 -> two groups
    (1, 40, 40, 1024, 64, torch.float16, -1, True),     # 320 workgroups at L = 1024 -> two groups
    (1, 12, 4, 4096, 128, torch.bfloat16, -1, True),    # head_dim 128, grouped-query: dQ grid 384 (two kv split groups), dK/dV grid 128 (row-half groups)
])
def test_backward_split_groups_vs_autograd_fp64_and_bit_reproducible(B, H, Hkv, L, D, dtype, bsplit, causal):
    """Round-5 verdict, item 4: grids of at most one workgroup per CU run the dQ and dK/dV workgroups as 2 | 3 split groups whose partial sums meet
    in LDS in group order (fa_bwd_kernels.hip QS).  Gradients against autograd through the dense fp64 expression on the same 16-bit inputs,
    bit-reproducible from call to call, and within rounding of the unsplit kernels (another summation order, not another result)."""
    native = _mods()
    torch.manual_seed(B * 131 + H * 17 + L)
    q = (torch.randn(B, H, L, D, device="cuda") * 0.7).to(dtype)
    k, v = ((torch.randn(B, Hkv, L, D, device="cuda") * 0.7).to(dtype) for _ in range(2))
    g = torch.randn(B, H, L, D, device="cuda").to(dtype)
    scale = D ** -0.5
    out = torch.empty_like(q)
    lse = torch.empty(B * H, L, device="cuda", dtype=torch.float32)
    st = lambda t: (t.stride(0), t.stride(1), t.stride(2))  # noqa: E731
    native.fwd_strided(q, k, v, out, lse, B * H, L, D, H, st(q), st(k), st(v), st(out), 1, 1, scale, causal, heads_kv=Hkv)

    def grads(mode):
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        with native.debug_options(bsplit=mode):
            native.bwd(q, k, v, out, g, lse, dq, dk, dv, B * H, L, D, H, scale, causal, heads_kv=Hkv)
        torch.cuda.synchronize()
        return dq, dk, dv

    a, b, plain = grads(bsplit), grads(bsplit), grads(1)
    for x, y in zip(a, b):
        assert torch.equal(x, y), "split groups fold in group order: the same bits on every call"
    qd, kd, vd = (t.double().requires_grad_(True) for t in (q, k, v))
    ke, ve = (t.repeat_interleave(H // Hkv, dim=1) for t in (kd, vd))
    s = (qd @ ke.transpose(-1, -2)) * scale
    if causal:
        s = s.masked_fill(torch.triu(torch.ones(L, L, device="cuda", dtype=torch.bool), 1), float("-inf"))
    ref = torch.softmax(s, -1) @ ve
    ref.backward(g.double())
    for got, want, unsplit, name in zip(a, (qd.grad, kd.grad, vd.grad), plain, ("dq", "dk", "dv")):
        lim = 6e-2 * max(1.0, want.abs().max().item())
        assert (got.double() - want).abs().max().item() <= lim, name
        assert (got.float() - unsplit.float()).abs().max().item() <= 0.5 * lim, f"{name}: split and unsplit kernels disagree beyond rounding"


@pytest.mark.parametrize("B,H,L,D,dtype,bsplit,head_inner", [
    (1, 2, 2048, 64, torch.bfloat16, -1, False),     # 32 row blocks -> 48 units; the library's own split groups on top
    (1, 1, 4096, 64, torch.float16, 3, False),       # one group: subsets spread its units over the XCDs
    (2, 2, 1024, 64, torch.bfloat16, 1, True),       # the smallest count the form takes (8 row blocks), unsplit workgroups, head-inner operands
    (1, 2, 1900, 64, torch.bfloat16, 2, False),      # ragged L (15 row blocks: not a multiple of 4 -> the form is refused, the plain grid runs)
    (1, 2, 1980, 64, torch.float16, 2, False),       # ragged L inside the last of 16 row blocks
    (1, 4, 2048, 128, torch.bfloat16, -1, True),     # head_dim 128
    (1, 2, 4096, 128, torch.float16, 2, False),
])
def test_backward_balanced_dq_units_vs_autograd_fp64_and_bit_reproducible(B, H, L, D, dtype, bsplit, head_inner):
    """Balanced dQ units (fa_bwd_kernels.hip BAL; debug option bbal): the heavy half of a causal grid's row blocks run as two units whose fp32 partial
    sums are parked in the dK / dV buffers until a reduce pass adds them - so: dQ against autograd through the dense fp64 expression and bit-reproducible,
    and dK / dV (written AFTER their buffers served as scratch) bit-equal to the run without the form."""
    native = _mods()
    torch.manual_seed(B * 131 + H * 17 + L + D)
    def mk():
        if head_inner:
            return (torch.randn(B, L, H, D, device="cuda") * 0.7).to(dtype).transpose(1, 2)
        return (torch.randn(B, H, L, D, device="cuda") * 0.7).to(dtype)
    q, k, v, g = mk(), mk(), mk(), mk()
    scale = D ** -0.5
    out = torch.empty_like(q)
    lse = torch.empty(B * H, L, device="cuda", dtype=torch.float32)
    st = lambda t: (t.stride(0), t.stride(1), t.stride(2))  # noqa: E731
    native.fwd_strided(q, k, v, out, lse, B * H, L, D, H, st(q), st(k), st(v), st(out), 1, 1, scale, True)

    def grads(bbal):
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        with native.debug_options(bbal=bbal, bsplit=bsplit):
            native.bwd(q, k, v, out, g, lse, dq, dk, dv, B * H, L, D, H, scale, True)
        torch.cuda.synchronize()
        return dq, dk, dv

    a, b, plain = grads(1), grads(1), grads(0)
    for x, y in zip(a, b):
        assert torch.equal(x, y), "two partial sums per element, added in one order: the same bits on every call"
    assert torch.equal(a[1], plain[1]) and torch.equal(a[2], plain[2]), "dK / dV must not notice that their buffers were scratch"
    qd, kd, vd = (t.double().requires_grad_(True) for t in (q, k, v))
    s = (qd @ kd.transpose(-1, -2)) * scale
    s = s.masked_fill(torch.triu(torch.ones(L, L, device="cuda", dtype=torch.bool), 1), float("-inf"))
    (torch.softmax(s, -1) @ vd).backward(g.double())
    lim = 6e-2 * max(1.0, qd.grad.abs().max().item())
    assert (a[0].double() - qd.grad).abs().max().item() <= lim
    assert (a[0].float() - plain[0].float()).abs().max().item() <= 0.5 * lim
