"""GPU parity tests (run with -m gpu on an MI355X).  Every case calls the HIP kernels through the
C ABI (ctypes -> libflash_attention.so) and checks them against the CPU oracle evaluated in fp64 on
the SAME 16-bit-rounded inputs.

Tolerance (from the reference's own fp16 criterion t_fa.py:78-79 / t_sdpa.py:86-87):
    |gpu - oracle| <= 1e-2 + 1e-2 * |oracle|      for fp16 AND bf16
plus the C++-test-style bound max|delta| <= 1e-2 (fp16, t_cpp:264-267) / 3e-2 (bf16), and
    atol 1e-3 / rtol 1e-5-style (t_fa.py:78-79) tightened to max|delta| <= 1e-4 for fp32.
"""

import json

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SEED = 1123581321
MAXABS = {torch.float16: 1e-2, torch.bfloat16: 3e-2, torch.float32: 1e-4}
# logsumexp stash: the softmax denominator is the sum of the 16-bit-rounded P the PV product uses (formed on the
# matrix pipe), so a row dominated by one key carries that key's rounding: <= 2^-8 relative for bf16 (log -> 3.9e-3)
LSE_ATOL = 6e-3


def _mods():
    import flash_attention_amd as fa
    from flash_attention_amd._impl import _native
    from flash_attention_amd._impl.flash_attention_qkv_packed import flash_attention_qkv_packed_with_lse

    assert _native.library_path() is not None, "native library not loaded"
    return fa, _native, flash_attention_qkv_packed_with_lse


def _check(out, ref, dtype, what=""):
    out = out.detach().cpu().double()
    err = (out - ref).abs()
    assert torch.isfinite(out).all(), f"{what}: non-finite output"
    if dtype == torch.float32:
        assert err.max().item() <= MAXABS[dtype], f"{what}: max|delta|={err.max().item():.3e}"
        return err.max().item()
    bad = err > (1e-2 + 1e-2 * ref.abs())
    assert not bool(bad.any()), f"{what}: {int(bad.sum())} elements out of tolerance, max|delta|={err.max().item():.3e}"
    assert err.max().item() <= MAXABS[dtype], f"{what}: max|delta|={err.max().item():.3e}"
    return err.max().item()


def _oracle(qkv16, scale=1.0):
    from oracle import dense_causal_attention_lse

    return dense_causal_attention_lse(qkv16.cpu(), scale=scale, compute_dtype=torch.float64)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("G,L,D,nq,nk", [
    (1, 128, 64, 4, 4),      # BASELINE config 1 shape
    (3, 64, 64, 1, 1),       # single kv tile
    (2, 32, 64, 1, 1),       # single wave tile
    (2, 96, 64, 3, 1),       # ragged vs the 64-key tile
    (5, 200, 64, 1, 1),      # L not a multiple of anything, G not a multiple of 8
    (9, 257, 64, 1, 1),      # one row past a tile edge
    (2, 1024, 64, 4, 4),
    (2, 1000, 128, 1, 1),
    (4, 1024, 128, 16, 16),  # t_fa.py:60-66 shape family (N=4, D=128, chunks 16x16)
    (8, 2048, 128, 2, 4),    # t_cpp:264-267 shape
    (1, 4096, 64, 8, 8),     # headline L/D, auto-heuristic chunks
    (1, 3000, 128, 1, 1),
])
def test_packed_vs_oracle(dtype, G, L, D, nq, nk):
    fa, _, with_lse = _mods()
    from oracle import frozen_randn

    qkv = frozen_randn(SEED + L + D, (3, G, L, D)).to(dtype)
    ref, ref_lse = _oracle(qkv)
    out = fa.flash_attention_qkv_packed(qkv.cuda(), nq, nk)
    assert out.shape == (G, L, D) and out.dtype == dtype and out.is_cuda
    _check(out, ref, dtype, f"packed {dtype} {G}x{L}x{D}")
    out2, lse = with_lse(qkv.cuda(), nq, nk)
    assert torch.equal(out2, out)
    torch.testing.assert_close(lse.cpu().double(), ref_lse, atol=LSE_ATOL, rtol=1e-4)


MFMA_KERNELS = [("dma", 1), ("dma", 2), ("pipe", 1), ("pipe", 2)]


@pytest.mark.parametrize("kernel,qb", MFMA_KERNELS)
@pytest.mark.parametrize("G,L,D", [(2, 32, 64), (9, 257, 64), (3, 600, 64), (2, 2304, 64), (1, 4608, 64), (2, 1000, 128)])
def test_every_mfma_kernel(knobs, kernel, qb, G, L, D):
    """The MFMA kernels (64-key tiles with LDS-DMA, software-pipelined 32-key steps), with one and two 32-row
    query blocks per wave, against the oracle - whatever the default
    selection for a shape is.  (Two blocks exist for head_dim 64 only; the request falls back to one otherwise.)"""
    _, _, with_lse = _mods()
    from oracle import frozen_randn

    knobs.set("FA_AMD_KERNEL", kernel)
    knobs.set("FA_AMD_QB", str(qb))
    qkv = frozen_randn(SEED + 7 * L + D, (3, G, L, D)).to(torch.bfloat16)
    ref, ref_lse = _oracle(qkv)
    out, lse = with_lse(qkv.cuda(), 1, 1)
    _check(out, ref, torch.bfloat16, f"{kernel} qb={qb} {G}x{L}x{D}")
    torch.testing.assert_close(lse.cpu().double(), ref_lse, atol=LSE_ATOL, rtol=1e-4)


@pytest.mark.parametrize("G,L", [(2, 1000), (1, 4608), (3, 257)])
def test_pipelined_kernel_8_wave_workgroups_head_dim_128(knobs, G, L):
    """The pipelined kernel with 8 waves x 32 rows (256-row workgroups, head_dim 128 only): out and logsumexp."""
    _, native, with_lse = _mods()
    from oracle import frozen_randn

    knobs.set("kernel", "pipe")
    knobs.set("waves", 8)
    p = native.plan(G, L, 128, torch.bfloat16, 1, 1)
    assert (p.kernel, p.waves, p.block_q, p.block_kv) == (2, 8, 256, 32)
    qkv = frozen_randn(SEED + 11 * L, (3, G, L, 128)).to(torch.bfloat16)
    ref, ref_lse = _oracle(qkv)
    out, lse = with_lse(qkv.cuda(), 1, 1)
    _check(out, ref, torch.bfloat16, f"pipe 8 waves {G}x{L}x128")
    torch.testing.assert_close(lse.cpu().double(), ref_lse, atol=LSE_ATOL, rtol=1e-4)


@pytest.mark.parametrize("waves", [4, 8])
@pytest.mark.parametrize("dtype,scale", [(torch.bfloat16, 1.0), (torch.bfloat16, 128 ** -0.5), (torch.float16, 128 ** -0.5)])
@pytest.mark.parametrize("G,L", [(2, 32), (3, 100), (2, 257), (2, 1000), (1, 4608), (9, 600)])
def test_pipelined_kernel_on_16x16x32_mfma_head_dim_128(knobs, dtype, scale, G, L, waves):
    """fa_fwd_pipe16_kernel (the pipelined kernel rebuilt on v_mfma_f32_16x16x32, head_dim 128): out and logsumexp
    against the oracle for ragged / short / long sequences - every path of its step machinery (generic first step
    only, left-over steps, groups of four, diagonal step, drain) - in bf16 and 1/sqrt(D)-scaled f16."""
    _, native, _ = _mods()
    from oracle import frozen_randn

    knobs.set("kernel", "pipe16")
    knobs.set("waves", waves)
    p = native.plan(G, L, 128, dtype, 1, 1, softmax_scale=scale)
    assert (p.kernel, p.waves, p.block_q, p.block_kv) == (4, waves, 32 * waves, 32)
    x = frozen_randn(SEED + 13 * L + G, (3, G, L, 128)).to(dtype)
    ref, ref_lse = _oracle(x, scale=scale)
    q, k, v = x.cuda()
    out = torch.empty_like(q)
    lse = torch.empty(G, L, device="cuda", dtype=torch.float32)
    st = (L * 128, 0, 128)
    native.fwd_strided(q, k, v, out, lse, G, L, 128, 1, st, st, st, st, 1, 1, scale)
    _check(out, ref, dtype, f"pipe16 {dtype} scale={scale:.3g} {G}x{L}x128")
    torch.testing.assert_close(lse.cpu().double(), ref_lse, atol=LSE_ATOL, rtol=1e-4)


@pytest.mark.parametrize("D,split", [(64, 2), (64, 4), (128, 2)])
@pytest.mark.parametrize("dtype,scale", [(torch.bfloat16, 1.0), (torch.float16, 1.0), (torch.float16, 0.125)])
@pytest.mark.parametrize("G,L", [(2, 32), (3, 100), (2, 257), (2, 1000), (1, 4608), (9, 600), (2, 2049)])
def test_split_kv_inside_the_workgroup(knobs, D, split, dtype, scale, G, L):
    """fa_fwd_dma_kernel with 2 | 4 kv split groups per workgroup (each group streams every split-th kv tile for the same
    128 query rows; (m, l, O) merged through LDS): out and logsumexp against the oracle - groups with no tile at all
    (short rows), with an odd tile count, ragged last tiles; skip-max (bf16, scaled f16) and exact-max (unscaled f16)."""
    _, native, _ = _mods()
    from oracle import frozen_randn

    knobs.set("split", split)
    p = native.plan(G, L, D, dtype, 1, 1, softmax_scale=scale)
    assert (p.kernel, p.waves, p.qb, p.block_q, p.block_kv) == (0, 4 * split, 1, 128, 64)
    x = (frozen_randn(SEED + 19 * L + G + D, (3, G, L, D)) * (0.5 if dtype == torch.float16 and scale == 1.0 else 1.0)).to(dtype)
    ref, ref_lse = _oracle(x, scale=scale)
    q, k, v = x.cuda()
    out = torch.empty_like(q)
    lse = torch.empty(G, L, device="cuda", dtype=torch.float32)
    st = (L * D, 0, D)
    native.fwd_strided(q, k, v, out, lse, G, L, D, 1, st, st, st, st, 1, 1, scale)
    _check(out, ref, dtype, f"split-kv {split} {dtype} scale={scale:.3g} {G}x{L}x{D}")
    torch.testing.assert_close(lse.cpu().double(), ref_lse, atol=LSE_ATOL, rtol=1e-4)


@pytest.mark.parametrize("D", [64, 128])
@pytest.mark.parametrize("dtype,scale", [(torch.bfloat16, 1.0), (torch.bfloat16, 0.125), (torch.float16, 1.0), (torch.float16, 0.125)])
@pytest.mark.parametrize("G,L", [(3, 129), (2, 256), (9, 384), (2, 1000), (10, 1024), (1, 2049), (17, 640)])
def test_q_tile_pairs_per_workgroup(knobs, D, dtype, scale, G, L):
    """fa_fwd_dma_kernel with q-tile pairs (a workgroup runs q tile n-1-i, then i, the second tile's Q rows and first K/V tile
    requested under the first tile's epilogue): odd tile counts (a middle tile alone), ragged last tiles, two-tile sequences,
    skip-max and exact-max arithmetic - out and logsumexp against the oracle, and bit for bit against the unpaired kernel (the
    arithmetic of a row is the same program either way)."""
    _, native, _ = _mods()
    from oracle import frozen_randn

    x = (frozen_randn(SEED + 23 * L + G + D, (3, G, L, D)) * (0.5 if dtype == torch.float16 and scale == 1.0 else 1.0)).to(dtype)
    ref, ref_lse = _oracle(x, scale=scale)
    q, k, v = x.cuda()
    st = (L * D, 0, D)
    res = {}
    for pair in (1, 0):
        knobs.set("kernel", "dma")
        knobs.set("waves", 4)
        knobs.set("pair", pair)
        knobs.set("balance", 0)
        p = native.plan(G, L, D, dtype, 1, 1, softmax_scale=scale)
        nq = (L + 127) // 128
        assert (p.kernel, p.waves, p.qb, p.block_q, p.pair) == (0, 4, 1, 128, pair)
        # (pair = 0: a group count that leaves XCDs idle is laid out over (group, q-tile subset) pairs, fa_plan_t.xs - tests/test_gpu_subsets.py)
        assert p.grid == ((G + 7) // 8 * 8 * ((nq + 1) // 2) if pair else (G * p.xs + 7) // 8 * 8 * ((nq + p.xs - 1) // p.xs))
        out = torch.empty_like(q)
        lse = torch.empty(G, L, device="cuda", dtype=torch.float32)
        native.fwd_strided(q, k, v, out, lse, G, L, D, 1, st, st, st, st, 1, 1, scale)
        res[pair] = (out, lse)
    _check(res[1][0], ref, dtype, f"pairs {dtype} scale={scale:.3g} {G}x{L}x{D}")
    torch.testing.assert_close(res[1][1].cpu().double(), ref_lse, atol=LSE_ATOL, rtol=1e-4)
    assert torch.equal(res[1][0], res[0][0]) and torch.equal(res[1][1], res[0][1]), "paired and unpaired kernels must agree bit for bit"


@pytest.mark.parametrize("G,L,D", [(96, 1024, 64), (40, 1024, 64), (24, 2048, 64), (48, 1024, 128), (96, 1000, 64)])
def test_placement_balanced_block_order(knobs, G, L, D):
    """Grids below one round of workgroup slots run their blocks in a placement-balanced order (a permutation of the XCD's
    slots in the kernel argument block): every (group, q tile) must still be computed exactly once - bit-identical output and
    logsumexp with the order on and off, and the oracle on sampled groups."""
    _, native, _ = _mods()
    from oracle import frozen_randn

    dtype = torch.bfloat16
    x = frozen_randn(SEED + 29 * L + G, (3, G, L, D)).to(dtype)
    q, k, v = x.cuda()
    st = (L * D, 0, D)
    res = {}
    for bal in (1, 0):
        knobs.set("kernel", "dma")
        knobs.set("waves", 4)
        knobs.set("pair", 0)
        knobs.set("balance", bal)
        p = native.plan(G, L, D, dtype, 1, 1, softmax_scale=D ** -0.5)
        assert p.balanced == bal, (p.balanced, p.grid)
        out = torch.full_like(q, float("nan"))
        lse = torch.full((G, L), float("nan"), device="cuda", dtype=torch.float32)
        native.fwd_strided(q, k, v, out, lse, G, L, D, 1, st, st, st, st, 1, 1, D ** -0.5)
        res[bal] = (out, lse)
    assert torch.equal(res[1][0], res[0][0]) and torch.equal(res[1][1], res[0][1])
    for g in (0, G // 2, G - 1):
        ref, ref_lse = _oracle(x[:, g:g + 1], scale=D ** -0.5)
        _check(res[1][0][g:g + 1], ref, dtype, f"balanced order group {g}")
        torch.testing.assert_close(res[1][1][g:g + 1].cpu().double(), ref_lse, atol=LSE_ATOL, rtol=1e-4)


@pytest.mark.parametrize("qb", [1, 2])
@pytest.mark.parametrize("dtype,scale", [(torch.bfloat16, 1.0), (torch.bfloat16, 64 ** -0.5), (torch.float16, 64 ** -0.5)])
@pytest.mark.parametrize("G,L", [(2, 32), (3, 100), (2, 257), (2, 1000), (1, 4608), (9, 600), (2, 2049)])
def test_pipelined_kernel_on_16x16x32_mfma_head_dim_64(knobs, dtype, scale, G, L, qb):
    """fa_fwd_pipe16_kernel at head_dim 64 (32 or 64 query rows per wave: one or two diagonal steps per wave): out and
    logsumexp against the oracle over every path of the step machinery."""
    _, native, _ = _mods()
    from oracle import frozen_randn

    knobs.set("kernel", "pipe16")
    knobs.set("qb", qb)
    p = native.plan(G, L, 64, dtype, 1, 1, softmax_scale=scale)
    assert (p.kernel, p.waves, p.qb, p.block_q, p.block_kv) == (4, 4, qb, 128 * qb, 32)
    x = frozen_randn(SEED + 17 * L + G, (3, G, L, 64)).to(dtype)
    ref, ref_lse = _oracle(x, scale=scale)
    q, k, v = x.cuda()
    out = torch.empty_like(q)
    lse = torch.empty(G, L, device="cuda", dtype=torch.float32)
    st = (L * 64, 0, 64)
    native.fwd_strided(q, k, v, out, lse, G, L, 64, 1, st, st, st, st, 1, 1, scale)
    _check(out, ref, dtype, f"pipe16 {dtype} scale={scale:.3g} qb={qb} {G}x{L}x64")
    torch.testing.assert_close(lse.cpu().double(), ref_lse, atol=LSE_ATOL, rtol=1e-4)


@pytest.mark.parametrize("causal", [1, 0])
@pytest.mark.parametrize("D,dtype,scale", [(64, torch.bfloat16, 1.0), (64, torch.bfloat16, 64 ** -0.5), (64, torch.float16, 1.0), (64, torch.float16, 64 ** -0.5),
                                           (128, torch.bfloat16, 128 ** -0.5), (128, torch.float16, 1.0)])
@pytest.mark.parametrize("G,L", [(1, 32), (2, 64), (3, 100), (2, 257), (17, 300), (2, 1000), (1, 4608), (9, 600), (5, 1024)])
def test_key_halves_kernel(knobs, D, dtype, scale, G, L, causal):
    _key_halves_case(knobs, D, dtype, scale, G, L, causal, 1)


@pytest.mark.parametrize("split", [2, 4])
@pytest.mark.parametrize("causal", [1, 0])
@pytest.mark.parametrize("D,dtype,scale", [(64, torch.bfloat16, 1.0), (64, torch.float16, 1.0), (64, torch.float16, 64 ** -0.5), (128, torch.bfloat16, 128 ** -0.5)])
@pytest.mark.parametrize("G,L", [(1, 32), (3, 100), (2, 257), (17, 300), (1, 4608), (5, 1024), (2, 1000)])
def test_key_halves_kernel_with_kv_split_groups(knobs, D, dtype, scale, G, L, causal, split):
    """The same kernel with 2 | 4 kv split groups per workgroup (8 | 16 waves: every group streams every SPLIT-th tile through buffers of
    its own; 2 SPLIT partial states per row block folded through LDS; 4 groups exist at head_dim 64 only - a request at 128 runs 1)."""
    _key_halves_case(knobs, D, dtype, scale, G, L, causal, split if (split == 2 or D == 64) else 1, force=split)


def _key_halves_case(knobs, D, dtype, scale, G, L, causal, split, force=None):
    """fa_fwd_kh_kernel (fa_fwd_kh.hip: 64-row workgroups, every wave = (row block, key half) of the streamed 64-key tiles, the two
    halves' softmax statistics merged through LDS): out and logsumexp against the oracle - a single half tile, key halves that hold
    nothing for a wave, ragged last tiles, waves without rows, both arithmetic modes (skip-max and per-tile max), causal and unmasked."""
    _, native, _ = _mods()
    from oracle import dense_attention_lse, frozen_randn

    knobs.set("kernel", "kh")
    if force:
        knobs.set("split", force)
    p = native.plan(G, L, D, dtype, 1, 1, softmax_scale=scale, causal=causal)
    assert (p.kernel, p.waves, p.qb, p.block_q, p.block_kv) == (6, 4 * split, 1, 64, 64)
    assert p.grid == ((G * p.xs + 7) // 8 * 8) * (((L + 63) // 64 + p.xs - 1) // p.xs)     # (xs: q-tile subsets, tests/test_gpu_subsets.py)
    assert p.lds_bytes == split * (2 * (64 * D * 2 + (D // 32) * (64 * 64 + 64)) + 16)
    x = frozen_randn(SEED + 23 * L + G + D, (3, G, L, D)).to(dtype)
    if causal:
        ref, ref_lse = _oracle(x, scale=scale)
    else:
        ref, ref_lse = dense_attention_lse(x[0][None], x[1][None], x[2][None], scale=scale, causal=False)
        ref, ref_lse = ref[0], ref_lse[0]
    q, k, v = x.cuda()
    out = torch.full_like(q, float("nan"))
    lse = torch.full((G, L), float("nan"), device="cuda", dtype=torch.float32)
    st = (L * D, 0, D)
    native.fwd_strided(q, k, v, out, lse, G, L, D, 1, st, st, st, st, 1, 1, scale, causal=causal)
    _check(out, ref, dtype, f"kh {dtype} D={D} scale={scale:.3g} causal={causal} {G}x{L}")
    torch.testing.assert_close(lse.cpu().double(), ref_lse, atol=LSE_ATOL, rtol=1e-4)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("step", [3.0, 25.0, 100.0])
def test_key_halves_growing_logits_force_the_exact_pass(knobs, dtype, step):
    """The optimistic pass of fa_fwd_kh_kernel must detect a P that left the safe range and redo the workgroup through its exact pass."""
    _, native, with_lse = _mods()
    from oracle import frozen_randn

    knobs.set("kernel", "kh")
    knobs.set("skipmax", 1)
    G, L, D = 2, 1024, 64
    x = frozen_randn(SEED + 58, (3, G, L, D)) * 0.05
    u = torch.zeros(D)
    u[:4] = 1.0
    tile = torch.arange(L) // 64
    x[0] += u
    x[1] += u[None, None, :] * (step / 4.0) * tile[None, :, None]
    x = x.to(dtype)
    assert native.plan(G, L, D, dtype, 1, 1).kernel == 6
    ref, ref_lse = _oracle(x)
    out, lse = with_lse(x.cuda(), 1, 1)
    _check(out, ref, dtype, f"kh growing logits step={step}")
    torch.testing.assert_close(lse.cpu().double(), ref_lse, atol=5e-2, rtol=2e-3)


@pytest.mark.parametrize("G,L,D,split", [(8, 4096, 64, 2), (64, 1024, 64, 1), (29, 2048, 64, 1), (16, 2048, 128, 1), (13, 1300, 64, 1)])
def test_key_halves_placement_balanced_block_order(knobs, G, L, D, split):
    """The key-halves kernel in the placement-balanced block order (grids below one round of its slots): every (group, q tile) still
    computed exactly once - bit-identical output and logsumexp with the order on and off, and the oracle on sampled groups."""
    _, native, _ = _mods()
    from oracle import frozen_randn

    dtype = torch.bfloat16
    x = frozen_randn(SEED + 31 * L + G, (3, G, L, D)).to(dtype)
    q, k, v = x.cuda()
    st = (L * D, 0, D)
    res = {}
    for bal in (1, 0):
        knobs.set("kernel", "kh")
        knobs.set("split", split)
        knobs.set("balance", bal)
        p = native.plan(G, L, D, dtype, 1, 1, softmax_scale=D ** -0.5)
        assert (p.kernel, p.waves, p.balanced) == (6, 4 * split, bal), (p.kernel, p.waves, p.balanced, p.grid)
        out = torch.full_like(q, float("nan"))
        lse = torch.full((G, L), float("nan"), device="cuda", dtype=torch.float32)
        native.fwd_strided(q, k, v, out, lse, G, L, D, 1, st, st, st, st, 1, 1, D ** -0.5)
        res[bal] = (out, lse)
    assert torch.equal(res[1][0], res[0][0]) and torch.equal(res[1][1], res[0][1])
    for g in (0, G // 2, G - 1):
        ref, ref_lse = _oracle(x[:, g:g + 1], scale=D ** -0.5)
        _check(res[1][0][g:g + 1], ref, dtype, f"kh balanced order group {g}")
        torch.testing.assert_close(res[1][1][g:g + 1].cpu().double(), ref_lse, atol=LSE_ATOL, rtol=1e-4)


def _run_strided(native, x, scale, causal=True):
    G, L, D = x.shape[1:]
    q, k, v = x.cuda()
    out = torch.full_like(q, float("nan"))
    lse = torch.full((G, L), float("nan"), device="cuda", dtype=torch.float32)
    st = (L * D, 0, D)
    native.fwd_strided(q, k, v, out, lse, G, L, D, 1, st, st, st, st, 1, 1, scale, causal=causal)
    return out, lse


@pytest.mark.parametrize("step", [3.0, 25.0, 100.0])
@pytest.mark.parametrize("D,qb", [(128, 1), (64, 1), (64, 2)])
def test_pipe16_growing_logits_force_the_exact_pass(knobs, step, D, qb):
    """The optimistic pass of fa_fwd_pipe16_kernel must detect a P that left the safe range (inf / NaN row sum) and
    redo the workgroup through its exact pass (cdna guide rule 26: force the rare branch)."""
    _, native, with_lse = _mods()
    from oracle import frozen_randn

    knobs.set("kernel", "pipe16")
    knobs.set("qb", qb)
    G, L = 2, 1024
    x = frozen_randn(SEED + 56, (3, G, L, D)) * 0.05
    u = torch.zeros(D)
    u[:4] = 1.0
    tile = torch.arange(L) // 64
    x[0] += u
    x[1] += u[None, None, :] * (step / 4.0) * tile[None, :, None]
    x = x.to(torch.bfloat16)
    ref, ref_lse = _oracle(x)
    out, lse = with_lse(x.cuda(), 1, 1)
    _check(out, ref, torch.bfloat16, f"pipe16 growing logits step={step} D={D} qb={qb}")
    torch.testing.assert_close(lse.cpu().double(), ref_lse, atol=5e-2, rtol=2e-3)


@pytest.mark.parametrize("nq,nk", [(1, 1), (1, 2), (2, 1), (2, 2), (2, 3), (3, 2), (3, 3)])
def test_tiny_fp32_all_chunk_pairs(golden_dir, nq, nk):
    """t_cpp:245-251: fp32 G=4 L=6 D=2, every chunk pair; golden G3 from the reference CPU path."""
    fa, _, _ = _mods()
    g = np.load(golden_dir / "g3_tiny_chunks.npz")
    qkv = torch.from_numpy(g["qkv"])
    out = fa.flash_attention_qkv_packed(qkv.cuda(), nq, nk)
    idx = [tuple(p) for p in g["pairs"].tolist()].index((nq, nk))
    assert np.abs(out.cpu().numpy() - g["outs"][idx]).max() <= 1e-5


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("G,L,D", [(2, 6, 2), (2, 33, 7), (3, 70, 24), (2, 300, 80), (8, 256, 128), (1, 1500, 32), (2, 1, 64), (1, 1, 1)])
def test_generic_kernel(knobs, dtype, G, L, D):
    """Shapes the generic kernel serves (odd D, L=1, and fp32 at any head dim when the fp32 MFMA kernel is switched
    off) - incl. t_cpp:256-259's fp32 8x256x128."""
    fa, native, _ = _mods()
    from oracle import frozen_randn

    if dtype != torch.float32 and D in (64, 128):
        pytest.skip("served by the MFMA kernel")
    if dtype == torch.float32 and D in (64, 128):
        knobs.set("FA_AMD_F32", "generic")      # (the fp32 MFMA kernel has its own test below)
    qkv = (frozen_randn(SEED + D, (3, G, L, D)) * (1.0 if D > 16 else 2.0)).to(dtype)
    # (round 4: a 16-bit head dim that is a multiple of 8 below 128 - 24, 32, 80 here - runs zero-extended inside the MFMA kernel instead;
    #  the same shapes, the same oracle: tests/test_gpu_head_dims.py has the dedicated cases)
    zero_extended = dtype != torch.float32 and D % 8 == 0 and D < 128
    assert native.plan(G, L, D, dtype, 1, 1).kernel in ((0, 2, 4, 6) if zero_extended else (1,))
    ref, ref_lse = _oracle(qkv)
    out = fa.flash_attention_qkv_packed(qkv.cuda(), 1, 1)
    _check(out, ref, dtype, f"generic {dtype} {G}x{L}x{D}")


@pytest.mark.parametrize("G,L,D", [(8, 256, 128), (2, 1, 64), (3, 600, 64), (2, 1000, 128), (9, 257, 64), (1, 4096, 64), (2, 33, 128)])
def test_fp32_mfma_kernel(G, L, D):
    """fp32 inputs at head_dim 64 / 128 run v_mfma_f32_32x32x2_f32 (fp32 products, fp32 accumulation): the reference's
    FLOAT path with its own bound, max|delta| <= 1e-4 (t_cpp:249-262; 8x256x128 is its shape), plus the log-sum-exp."""
    _, native, with_lse = _mods()
    from oracle import frozen_randn

    qkv = frozen_randn(SEED + 3 * L + D, (3, G, L, D))
    assert native.plan(G, L, D, torch.float32, 1, 1).kernel == 3
    ref, ref_lse = _oracle(qkv)
    out, lse = with_lse(qkv.cuda(), 1, 1)
    assert out.dtype == torch.float32
    _check(out, ref, torch.float32, f"fp32 mfma {G}x{L}x{D}")
    torch.testing.assert_close(lse.cpu().double(), ref_lse, atol=1e-4, rtol=1e-5)


def test_golden_config1_and_g5(golden_dir):
    """Golden vectors produced by the reference's own Python (tests/golden/generate_golden.py)."""
    fa, _, _ = _mods()
    from oracle import frozen_randn

    g = np.load(golden_dir / "g1_g2_config1.npz")
    qkv = torch.from_numpy(g["qkv"])
    # G1 (fp32 reference output): our fp32 GPU path to 1e-5; our 16-bit paths within 16-bit tolerance of it
    out32 = fa.flash_attention_qkv_packed(qkv.cuda(), 4, 4).cpu()
    assert (out32 - torch.from_numpy(g["out_f32"])).abs().max().item() <= 1e-5
    # G2: the reference's 16-bit CPU path rounds scores to 16 bits (noise up to 1e-2 fp16 / 7e-2 bf16,
    # see tests/test_oracle.py); bound = that noise + our tolerance
    for name, dt, bound in (("out_f16", torch.float16, 3e-2), ("out_bf16", torch.bfloat16, 1.5e-1)):
        o = fa.flash_attention_qkv_packed(qkv.to(dt).cuda(), 4, 4).cpu().float()
        assert (o - torch.from_numpy(g[name])).abs().max().item() <= bound
    g5 = np.load(golden_dir / "g5_large_sampled.npz")
    for tag in ("N4_L1024_D128", "G2_L4096_D64"):
        G, L, D = (int(x) for x in g5[f"{tag}_shape"])
        rows = g5[f"{tag}_rowidx"]
        x = frozen_randn(SEED, (3, G, L, D))
        o32 = fa.flash_attention_qkv_packed(x.cuda(), 1, 1).cpu().double()
        assert (o32[:, rows, :] - torch.from_numpy(g5[f"{tag}_f32_rows"])).abs().max().item() <= 1e-4
        assert abs(o32.sum().item() - float(g5[f"{tag}_f32_sum"])) <= 1e-1
        for dn, dt, bound in (("f16", torch.float16, 4e-2), ("bf16", torch.bfloat16, 2e-1)):
            o = fa.flash_attention_qkv_packed(x.to(dt).cuda(), 1, 1).cpu().double()
            assert (o[:, rows, :] - torch.from_numpy(g5[f"{tag}_{dn}_rows"])).abs().max().item() <= bound


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("batch_shape", [(2,), (2, 3), (2, 3, 5)])
@pytest.mark.parametrize("L", [256, 1024])
def test_sdpa_override_vs_oracle(dtype, batch_shape, L):
    """t_sdpa.py:59-94: batch shapes (2,), (2,3), (2,3,5), L in {256,1024}, D=64, vs stock SDPA math."""
    import torch.nn.functional as F

    import flash_attention_amd.auto  # noqa: F401
    from oracle import frozen_randn

    assert hasattr(F.scaled_dot_product_attention, "__wrapped__")
    D = 64
    if len(batch_shape) == 3 and L == 1024 and dtype == torch.float32:
        pytest.skip("generic fp32 kernel: keep the GPU suite short")
    q, k, v = frozen_randn(SEED + L + len(batch_shape), (3, *batch_shape, L, D)).to(dtype)
    out = F.scaled_dot_product_attention(q.cuda(), k.cuda(), v.cuda(), is_causal=True)
    assert out.shape == q.shape and out.dtype == dtype
    G = int(np.prod(batch_shape))
    ref, _ = _oracle(torch.stack([q, k, v]).reshape(3, G, L, D), scale=D ** -0.5)
    _check(out.reshape(G, L, D), ref, dtype, f"sdpa {dtype} {batch_shape} L={L}")


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("kernel", ["", "dma", "pipe"])
def test_sdpa_long_sequence_scaled_logits(knobs, dtype, kernel):
    """The SDPA override at L = 4096 with a [B, T, 3C]-style head-inner projection (strided q/k/v views): the
    1/sqrt(D) scale selects the skip-max arithmetic for f16 too, so the default here is the pipelined kernel for
    both dtypes (G = 64 groups x 16 row tiles = 1024 workgroups); forced kernels for comparison."""
    import torch.nn.functional as F

    import flash_attention_amd.auto  # noqa: F401
    _, native, _ = _mods()
    from oracle import frozen_randn

    if kernel:
        knobs.set("FA_AMD_KERNEL", kernel)
    B, H, L, D = 4, 16, 4096, 64
    assert native.plan(B * H, L, D, torch.bfloat16, 8, 8).kernel in ((0,) if kernel == "dma" else (2, 4))   # (pipe | pipe16)
    x = frozen_randn(SEED + 4096, (B, L, 3 * H * D)).to(dtype).cuda()
    q, k, v = (t.view(B, L, H, D).transpose(1, 2) for t in x.split(H * D, dim=2))
    out = F.scaled_dot_product_attention(q, k, v, is_causal=True)
    assert out.shape == (B, H, L, D) and out.dtype == dtype
    for b, hh in ((0, 0), (3, 15), (1, 7)):
        ref, _ = _oracle(torch.stack([q[b, hh], k[b, hh], v[b, hh]]).unsqueeze(1).cpu(), scale=D ** -0.5)
        _check(out[b, hh].unsqueeze(0), ref, dtype, f"sdpa L=4096 {dtype} kernel={kernel or 'default'} ({b},{hh})")


def test_sdpa_golden_g4_and_strided_views(golden_dir):
    import torch.nn.functional as F

    import flash_attention_amd.auto  # noqa: F401
    from oracle import frozen_randn

    g = np.load(golden_dir / "g4_sdpa.npz")
    rows = g["rowidx"]
    q, k, v = frozen_randn(SEED + 4, (3, 2, 3, 256, 64))
    out = F.scaled_dot_product_attention(q.cuda(), k.cuda(), v.cuda(), is_causal=True).cpu()
    assert (out[..., rows, :] - torch.from_numpy(g["out_ipu_branch_rows"])).abs().max().item() <= 1e-5
    assert (out[..., rows, :] - torch.from_numpy(g["out_passthrough_rows"])).abs().max().item() <= 1e-5
    # nanoGPT-style views: q,k,v = split of a [B,T,3C] projection, viewed [B,H,T,hs] (no copy on our side)
    B, T, H, hs = 2, 512, 12, 64
    x = frozen_randn(SEED + 9, (B, T, 3 * H * hs)).to(torch.bfloat16).cuda()
    qv, kv, vv = (t.view(B, T, H, hs).transpose(1, 2) for t in x.split(H * hs, dim=2))
    out = F.scaled_dot_product_attention(qv, kv, vv, is_causal=True)
    assert out.shape == (B, H, T, hs)
    assert out.transpose(1, 2).is_contiguous(), "output should come back head-inner like the input"
    ref, _ = _oracle(torch.stack([qv, kv, vv]).reshape(3, B * H, T, hs).cpu(), scale=hs ** -0.5)
    _check(out.reshape(B * H, T, hs), ref, torch.bfloat16, "nanoGPT-style strided views")
    # off-device / unsupported: passthrough keeps working
    qc = q[0, 0]
    ref_cpu = F.scaled_dot_product_attention.__wrapped__(qc, qc, qc, is_causal=True)
    assert torch.equal(F.scaled_dot_product_attention(qc, qc, qc, is_causal=True), ref_cpu)


def test_forced_rescale_spike():
    """Online-softmax rescale path forced at a chosen tile (a K row spiked against a Q row late in
    the sweep), cdna guide rule 26: random data alone never proves the rare branch."""
    fa, _, _ = _mods()
    from oracle import frozen_randn

    G, L, D = 2, 1024, 64
    qkv = frozen_randn(SEED + 77, (3, G, L, D)) * 0.25
    qkv[1, 0, 700] = qkv[0, 0, 900] * 40.0      # key 700 dominates query 900 (and nearby rows less so)
    qkv[1, 1, 10] = qkv[0, 1, 1000] * 40.0       # early spike: later tiles must not disturb it
    for dtype in (torch.bfloat16, torch.float16):
        x = qkv.to(dtype)
        ref, _ = _oracle(x)
        _check(fa.flash_attention_qkv_packed(x.cuda(), 1, 1), ref, dtype, "spiked rescale")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_full_size_properties_config2(knobs, dtype):
    """BASELINE config 2 (G=128, L=4096, D=64): size-independent properties + oracle on sampled groups."""
    fa, native, _ = _mods()
    G, L, D = 128, 4096, 64
    gen = torch.Generator(device="cuda").manual_seed(0)
    qkv = torch.randn(3, G, L, D, device="cuda", generator=gen).to(dtype)
    out = fa.flash_attention_qkv_packed(qkv, 8, 8)
    assert torch.isfinite(out).all()
    # (1) row 0 attends to key 0 only: out[:, 0] == v[:, 0] bit-exactly
    assert torch.equal(out[:, 0], qkv[2, :, 0])
    # (2) convexity: every output lies inside the running [min, max] of the values it may attend to
    v32 = qkv[2].float()
    lo, hi = torch.cummin(v32, dim=1).values, torch.cummax(v32, dim=1).values
    o32 = out.float()
    eps = 2e-2
    assert bool(((o32 >= lo - eps) & (o32 <= hi + eps)).all())
    # (3) linearity in V for power-of-two factors is bit-exact
    qkv2 = qkv.clone()
    qkv2[2] *= 4
    # (bit-exact for every normal value; fp16 outputs below 2^-14 sit on the denormal grid, where
    # round(4x) != 4*round(x), hence the 2^-22 slack that only tiny values can use)
    o4 = fa.flash_attention_qkv_packed(qkv2, 8, 8)
    assert torch.equal(o4, out * 4) or (o4.float() - out.float() * 4).abs().max().item() <= 2.0 ** -22
    # (4) groups are independent: a permutation of the groups permutes the output bit-exactly
    perm = torch.randperm(G, device="cuda", generator=gen)
    assert torch.equal(fa.flash_attention_qkv_packed(qkv[:, perm].contiguous(), 8, 8), out[perm])
    # (5) causality: truncating the sequence does not change the surviving rows - bit-exactly when the same
    # kernel serves both lengths (the default selection depends on L), within rounding otherwise
    Lt = 1536
    trunc = fa.flash_attention_qkv_packed(qkv[:, :, :Lt].contiguous(), 8, 8)
    # (two 16-bit results within an ulp of the exact value can sit two ulps apart: 2^-6 relative for bf16)
    d_, r_ = (trunc.float() - out[:, :Lt].float()).abs(), out[:, :Lt].float().abs()
    assert bool((d_ <= 1e-2 + 2.0 ** -6 * r_).all()), f"truncated, default kernel: max|delta|={d_.max().item():.3e}"
    plan = native.plan(G, L, D, dtype, 8, 8)
    knobs.set("FA_AMD_KERNEL", {0: "dma", 2: "pipe", 4: "pipe16"}[plan.kernel])
    knobs.set("FA_AMD_QB", str(plan.block_q // (32 * plan.waves)))
    assert torch.equal(fa.flash_attention_qkv_packed(qkv[:, :, :Lt].contiguous(), 8, 8), out[:, :Lt])
    knobs.reset("FA_AMD_KERNEL")
    knobs.reset("FA_AMD_QB")
    # (6) the chunk counts never change the result beyond rounding; same hardware tile => identical
    assert torch.equal(fa.flash_attention_qkv_packed(qkv, 16, 4), out)
    # (7) oracle on sampled groups
    for gidx in (0, 77, 127):
        ref, _ = _oracle(qkv[:, gidx:gidx + 1])
        _check(out[gidx:gidx + 1], ref, dtype, f"config2 group {gidx}")


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_notebook_longest_context_65536(dtype):
    """The reference's longest demonstrated context (demo/flash-attention-ipu-demo.ipynb:939, :952-963): qkv [3, 8, 65536, 64]
    fp16 through flash_attention_qkv_packed with num_chunks_q = num_chunks_kv = 64, and its bf16 twin: the whole launch - 1024
    kv tiles per last row; unscaled fp16 runs the per-tile-max arithmetic, bf16 the pipelined skip-max kernel - row 0, convexity,
    and the CPU oracle (dense_causal_attention_rows: fa.py:48-53 restricted to sampled rows) with logsumexp on every group."""
    fa, native, with_lse = _mods()
    from oracle import dense_causal_attention_rows

    G, L, D = 8, 65536, 64
    plan = native.plan(G, L, D, dtype, 64, 64)
    assert plan.kernel in (0, 2, 4) and plan.chunk_q == 1024 and plan.chunk_kv == 1024
    gen = torch.Generator(device="cuda").manual_seed(65536)
    qkv = torch.randn(3, G, L, D, device="cuda", generator=gen).to(dtype)
    out, lse = with_lse(qkv, 64, 64)
    assert torch.isfinite(out).all() and torch.isfinite(lse).all()
    assert torch.equal(out[:, 0], qkv[2, :, 0])                             # row 0 sees key 0 only
    v32 = qkv[2].float()
    lo, hi = torch.cummin(v32, dim=1).values, torch.cummax(v32, dim=1).values
    assert bool(((out.float() >= lo - 2e-2) & (out.float() <= hi + 2e-2)).all())   # convexity
    del v32, lo, hi
    rows = [1, 63, 64, 1023, 1024, 4097, 16383, 32768, 50001, 65534, 65535]
    ref, ref_lse = dense_causal_attention_rows(qkv.cpu(), rows)
    _check(out[:, rows], ref, dtype, f"notebook shape [3,8,65536,64] {dtype}")
    torch.testing.assert_close(lse[:, rows].cpu().double(), ref_lse, atol=LSE_ATOL if dtype == torch.bfloat16 else 2e-3, rtol=1e-4)


def test_long_context_config3_full_groups_vs_oracle_rows():
    """BASELINE config 3 as written (G = 16, L = 32768, D = 128, bf16): the whole launch, checked against the CPU
    oracle (oracle.dense_causal_attention_rows: fa.py:48-53 restricted to sampled query rows) for every group."""
    fa, native, with_lse = _mods()
    from oracle import dense_causal_attention_rows

    G, L, D = 16, 32768, 128
    plan = native.plan(G, L, D, torch.bfloat16, 16, 16)
    assert plan.kernel in (0, 2, 4) and plan.block_q in (128, 256)           # an MFMA kernel (rules: 8-wave 64-key tiles)
    gen = torch.Generator(device="cuda").manual_seed(3)
    qkv = torch.randn(3, G, L, D, device="cuda", generator=gen).to(torch.bfloat16)
    out, lse = with_lse(qkv, 16, 16)
    assert torch.isfinite(out).all() and torch.equal(out[:, 0], qkv[2, :, 0])
    rows = [1, 31, 32, 255, 256, 4095, 4096, 16383, 20000, 32767]
    ref, ref_lse = dense_causal_attention_rows(qkv.cpu(), rows)
    _check(out[:, rows], ref, torch.bfloat16, "config 3, G=16, sampled rows")
    torch.testing.assert_close(lse[:, rows].cpu().double(), ref_lse, atol=LSE_ATOL, rtol=1e-4)


def test_config5_per_gpu_shard_shape(knobs):
    """BASELINE config 5's per-GPU shard at 8 GPUs: [3, 256, 8192, 64] bf16 (G = 2048 / 8): size-independent
    properties on the whole launch + the dense oracle on sampled groups + oracle rows on every group."""
    fa, native, with_lse = _mods()
    from oracle import dense_causal_attention_rows

    G, L, D = 256, 8192, 64
    plan = native.plan(G, L, D, torch.bfloat16, 16, 8)
    assert (plan.kernel, plan.block_q) == (2, 256)                          # pipelined kernel, 128 steps per last wave
    gen = torch.Generator(device="cuda").manual_seed(5)
    qkv = torch.randn(3, G, L, D, device="cuda", generator=gen).to(torch.bfloat16)
    out, lse = with_lse(qkv, 16, 8)
    assert torch.isfinite(out).all()
    assert torch.equal(out[:, 0], qkv[2, :, 0])                             # row 0 sees key 0 only
    v32 = qkv[2].float()
    lo, hi = torch.cummin(v32, dim=1).values, torch.cummax(v32, dim=1).values
    assert bool(((out.float() >= lo - 2e-2) & (out.float() <= hi + 2e-2)).all())   # convexity
    del v32, lo, hi
    perm = torch.randperm(G, device="cuda", generator=gen)                  # groups are independent
    assert torch.equal(fa.flash_attention_qkv_packed(qkv[:, perm].contiguous(), 16, 8), out[perm])
    Lt = 4352                                                               # causal truncation, same kernel forced
    knobs.set("kernel", "pipe")
    knobs.set("qb", 2)
    assert torch.equal(fa.flash_attention_qkv_packed(qkv[:, :, :Lt].contiguous(), 16, 8), out[:, :Lt])
    knobs.reset()
    assert torch.equal(fa.flash_attention_qkv_packed(qkv, 4, 4), out)       # chunk counts change nothing (same tile)
    rows = [1, 63, 64, 255, 256, 4095, 4096, 8000, 8191]
    ref, ref_lse = dense_causal_attention_rows(qkv.cpu(), rows)
    _check(out[:, rows], ref, torch.bfloat16, "config 5 shard, sampled rows of every group")
    torch.testing.assert_close(lse[:, rows].cpu().double(), ref_lse, atol=LSE_ATOL, rtol=1e-4)
    for gidx in (0, 255):
        ref, _ = _oracle(qkv[:, gidx:gidx + 1])
        _check(out[gidx:gidx + 1], ref, torch.bfloat16, f"config 5 shard group {gidx}")


def test_error_paths_through_c_abi():
    fa, native, _ = _mods()
    x = torch.zeros(3, 1, 6, 4, device="cuda")
    with pytest.raises(ValueError, match="divisible by num_chunks_q"):
        fa.flash_attention_qkv_packed(x, 4, 1)
    with pytest.raises(TypeError):
        fa.flash_attention_qkv_packed(x.double(), 1, 1)
    # empty batch is a no-op
    assert fa.flash_attention_qkv_packed(torch.zeros(3, 0, 8, 64, device="cuda", dtype=torch.bfloat16), 1, 1).shape == (0, 8, 64)
    # the C ABI rejects what the Python layer cannot see
    lib = native.load_native_library()
    rc = lib.fa_fwd_qkv_packed(None, None, None, 1, 8, 64, 1, 1, 1, None)
    assert rc == -1 and b"non-null" in lib.fa_last_error()


def test_stream_and_graph_capture():
    """The C ABI enqueues on the caller's stream and neither allocates nor synchronises: the forward can be
    issued on a side stream and captured into a HIP graph (replayed with new input contents)."""
    _, native, _ = _mods()
    from oracle import frozen_randn

    G, L, D = 4, 512, 64
    x1 = frozen_randn(SEED + 21, (3, G, L, D)).to(torch.bfloat16)
    x2 = frozen_randn(SEED + 22, (3, G, L, D)).to(torch.bfloat16)
    qkv = x1.cuda()
    out = torch.empty(G, L, D, device="cuda", dtype=torch.bfloat16)
    lse = torch.empty(G, L, device="cuda", dtype=torch.float32)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        native.fwd_packed(qkv, out, lse, 1, 1)          # warm-up on the side stream
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            native.fwd_packed(qkv, out, lse, 1, 1)
    torch.cuda.current_stream().wait_stream(side)
    for x in (x1, x2):
        qkv.copy_(x.cuda())
        out.zero_()
        graph.replay()
        torch.cuda.synchronize()
        ref, ref_lse = _oracle(x)
        _check(out, ref, torch.bfloat16, "graph replay")
        torch.testing.assert_close(lse.cpu().double(), ref_lse, atol=LSE_ATOL, rtol=1e-4)


def test_noncontiguous_and_offset_inputs():
    """Packed op on a non-contiguous qkv view (made contiguous by the wrapper) and fa_fwd on views that
    start at a non-zero, 16-byte aligned offset inside a larger buffer."""
    fa, native, _ = _mods()
    from oracle import frozen_randn

    G, L, D = 3, 200, 64
    big = frozen_randn(SEED + 31, (3, G, L, 2 * D)).to(torch.float16)
    view = big[..., D:]                                   # stride 2D along rows: not contiguous
    ref, _ = _oracle(view.contiguous())
    _check(fa.flash_attention_qkv_packed(view.cuda(), 1, 1), ref, torch.float16, "non-contiguous packed")
    bg = big.cuda()
    q, k, v = (bg[i][..., D:] for i in range(3))          # row stride 2D, base offset D elements (128 B)
    out = torch.empty(G, L, D, device="cuda", dtype=torch.float16)
    st = (L * 2 * D, 0, 2 * D)
    native.fwd_strided(q, k, v, out, None, G, L, D, 1, st, st, st, (L * D, 0, D), 1, 1, 1.0)
    _check(out, ref, torch.float16, "strided offset views")


def test_cpp_caller_of_the_c_abi(tmp_path):
    """tests/cpp/test_capi.cpp: plain C++ (no torch) links libflash_attention.so and checks it against a dense
    fp64 host computation - the counterpart of the reference's C++ test binary (t_cpp:34-102)."""
    import pathlib
    import subprocess

    root = pathlib.Path(__file__).resolve().parent.parent
    lib_dir = root / "flash-attention-ipu_amd" / "build"
    exe = tmp_path / "test_capi"
    subprocess.check_call(["hipcc", "-O2", f"-I{root / 'include'}", str(root / "tests" / "cpp" / "test_capi.cpp"),
                           f"-L{lib_dir}", "-lflash_attention", f"-Wl,-rpath,{lib_dir}", "-o", str(exe)])
    res = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0 and "PASSED" in res.stdout, res.stdout + res.stderr


@pytest.mark.parametrize("kernel,qb", [("", 0), ("dma", 2), ("pipe", 1), ("pipe", 2), ("split2", 1), ("split4", 1)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("step", [3.0, 25.0, 100.0])
def test_growing_logits_force_the_redo_branch(knobs, dtype, step, kernel, qb):
    """The bf16 kernel keeps its reference max from the first tile and only redoes a tile (exact max +
    rescale) when a row sum shows that some P left the safe range (cdna guide rule 26: a rare,
    data-dependent branch needs inputs that FORCE it and a full-tensor independent reference).
    Logits grow by `step` per 64-key tile: 3 never triggers, 25 triggers once the lag passes 2^60
    (every other tile), 100 overflows exp2 to +inf inside a single tile.  The pipelined kernel handles the
    same event by flagging it and re-running the workgroup through its safe pass."""
    fa, _, _ = _mods()
    from oracle import frozen_randn

    if kernel.startswith("split"):      # kv split groups: the groups' reference maxima differ by up to 3 tiles of growth
        knobs.set("split", int(kernel[5:]))
    elif kernel:
        knobs.set("FA_AMD_KERNEL", kernel)
        knobs.set("FA_AMD_QB", str(qb))

    G, L, D = 2, 1024, 64
    x = frozen_randn(SEED + 55, (3, G, L, D)) * 0.05
    u = torch.zeros(D)
    u[:4] = 1.0                                           # |u|^2 = 4
    tile = torch.arange(L) // 64
    x[0] += u                                             # q = u + noise
    x[1] += u[None, None, :] * (step / 4.0) * tile[None, :, None]   # k_c = u * step/4 * tile(c): s grows by `step` per tile
    x = x.to(dtype)
    ref, ref_lse = _oracle(x)
    out2, lse = _mods()[2](x.cuda(), 1, 1)
    _check(out2, ref, dtype, f"growing logits step={step}")
    torch.testing.assert_close(lse.cpu().double(), ref_lse, atol=5e-2, rtol=2e-3)


def test_differential_fuzz_of_the_kernel_variants():
    """tests/tools/fuzz_kernels.py, bounded: random shapes / dtypes / scales / layouts; every MFMA kernel variant
    against the register-staged baseline kernel and (short sequences) the fp64 oracle."""
    import pathlib
    import sys

    sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent / "tools"))
    from fuzz_kernels import fuzz

    msgs = []
    bad, _ = fuzz(40, 20260, log=msgs.append)
    assert bad == 0, "\n".join(m for m in msgs if m.startswith("MISMATCH"))


def test_graft_entry_smoke_runs():
    """`__graft_entry__.smoke()` - what the driver runs on the GPU box before the bench - must pass with the shipped tuning
    table (its plan check follows the table, not a fixed kernel)."""
    import __graft_entry__

    __graft_entry__.smoke()
