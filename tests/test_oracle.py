"""CPU tests of the oracle itself: the restatements against the golden vectors captured from the
reference (tests/golden/generate_golden.py) and against each other.  No GPU, no product code."""

import json

import numpy as np
import pytest
import torch

from oracle import dense_causal_attention, dense_causal_attention_lse, frozen_randn, sdpa_chunk_heuristic
from oracle import chunked

SEED = 1123581321


def test_dense_matches_golden_g1_g2(golden_dir):
    g = np.load(golden_dir / "g1_g2_config1.npz")
    qkv = torch.from_numpy(g["qkv"])
    # G1: fp32, bit-for-bit the same expression as the reference's CPU branch
    out = dense_causal_attention(qkv)
    assert torch.equal(out, torch.from_numpy(g["out_f32"]))
    # G2: the oracle run in the same 16-bit dtype reproduces the reference's 16-bit CPU path exactly.
    # That path rounds the *scores* to 16 bits (torch CPU matmul in the tensor dtype), so it sits
    # up to 1e-2 (fp16) / 7e-2 (bf16) away from exact arithmetic on the same rounded inputs: the
    # bounds below document that noise, they are not the GPU parity tolerance.
    for name, dt, noise in (("out_f16", torch.float16, 2e-2), ("out_bf16", torch.bfloat16, 1e-1)):
        assert torch.equal(dense_causal_attention(qkv.to(dt)).float(), torch.from_numpy(g[name]))
        ref16 = torch.from_numpy(g[name]).double()
        orc = dense_causal_attention(qkv.to(dt), compute_dtype=torch.float64)
        assert (ref16 - orc).abs().max().item() < noise


def test_chunked_matches_golden_g3_all_chunk_pairs(golden_dir):
    g = np.load(golden_dir / "g3_tiny_chunks.npz")
    qkv = g["qkv"]
    for (nq, nk), ref in zip(g["pairs"], g["outs"]):
        if 6 // nk == 1:
            continue
        out, _ = chunked.chunked_attention(qkv.astype(np.float64), int(nq), int(nk))
        np.testing.assert_allclose(out, ref, atol=1e-5, rtol=0)  # t_cpp:245-251 bound
        out32, _ = chunked.chunked_attention(qkv, int(nq), int(nk))
        np.testing.assert_allclose(out32, ref, atol=1e-5, rtol=0)
    # chunk independence of the reference itself
    assert np.abs(g["outs"] - g["outs"][0]).max() < 1e-6


def test_triu_offset_sequences():
    # examples quoted in SURVEY.md 8(a7) from FA.cpp:47-126
    assert chunked.triu_offsets(2, 2) == [1]
    assert chunked.triu_offsets(2, 3) == [1, 0, 2]
    assert chunked.triu_offsets(512, 1024) == [1, 513]
    assert chunked.triu_offsets(1024, 512) == [1, -511]
    with pytest.raises(RuntimeError):
        chunked.triu_offsets(4, 1)  # the reference loops forever here


@pytest.mark.parametrize("L", [6, 8, 12, 16, 24, 32, 48, 64])
def test_chunked_equals_dense_every_divisor_pair(L):
    """The cyclic-mask chunked program == the dense -10000-masked result for every (nq, nk) with
    Bk > 1 (semantic contract)."""
    qkv = frozen_randn(SEED + L, (3, 2, L, 4)).double()
    dense, lse_d = dense_causal_attention_lse(qkv)
    divs = [d for d in range(1, L + 1) if L % d == 0]
    for nq in divs:
        for nk in divs:
            if L // nk == 1:
                continue
            out, lse = chunked.chunked_attention(qkv.numpy(), nq, nk)
            assert np.abs(out - dense.numpy()).max() < 1e-9, (nq, nk)
            assert np.abs(lse - lse_d.numpy()).max() < 1e-9, (nq, nk)


def test_chunked_medium_shapes_fp32():
    # t_cpp:256-259: fp32 G=8 L=256 D=128, (nq,nk) in {2,4}^2, max|delta| <= 1e-5 (reduced G for CPU time)
    qkv = frozen_randn(SEED, (3, 2, 256, 128))
    dense = dense_causal_attention(qkv, compute_dtype=torch.float64).numpy()
    for nq in (2, 4):
        for nk in (2, 4):
            # the reference bound (1e-5) is flash-vs-vanilla with BOTH in fp32; against an fp64 dense
            # result the scalar fp32 accumulation alone costs ~1.5e-5, so check fp32 at 1e-4 and the
            # fp64 run of the same program at 1e-12
            out, _ = chunked.chunked_attention(qkv.numpy(), nq, nk)
            assert np.abs(out - dense).max() <= 1e-4
            out64, _ = chunked.chunked_attention(qkv.numpy().astype(np.float64), nq, nk)
            assert np.abs(out64 - dense).max() <= 1e-12


def test_dense_matches_golden_g5_sampled_rows(golden_dir):
    g = np.load(golden_dir / "g5_large_sampled.npz")
    for tag in ("N4_L1024_D128", "G2_L4096_D64"):
        G, L, D = (int(x) for x in g[f"{tag}_shape"])
        rows = g[f"{tag}_rowidx"]
        x = frozen_randn(SEED, (3, G, L, D))
        out = dense_causal_attention(x, compute_dtype=torch.float32)
        np.testing.assert_allclose(out[:, rows, :].double().numpy(), g[f"{tag}_f32_rows"], atol=1e-5, rtol=1e-5)
        assert abs(out.double().sum().item() - float(g[f"{tag}_f32_sum"])) < 1e-2
        for dt_name, dt, noise in (("bf16", torch.bfloat16, 0.15), ("f16", torch.float16, 3e-2)):
            # same-dtype oracle == reference 16-bit path (bitwise); fp64 oracle within that path's noise
            o16 = dense_causal_attention(x.to(dt))
            assert torch.equal(o16[:, rows, :].double(), torch.from_numpy(g[f"{tag}_{dt_name}_rows"]))
            o64 = dense_causal_attention(x.to(dt), compute_dtype=torch.float64)
            assert (o64[:, rows, :] - torch.from_numpy(g[f"{tag}_{dt_name}_rows"])).abs().max().item() < noise


def test_dense_matches_golden_g7_head_dims(golden_dir):
    """G7: the reference's own Python at head dims other than 64 / 128 (32, 96, 192, 256: its op has no head-dim restriction, FA.cpp:175-195) -
    pins the oracle where the round-4 kernels run zero-extended (32, 96) or at the new native dims (192, 256)."""
    g = np.load(golden_dir / "g7_head_dims.npz")
    G, L = (int(x) for x in g["shape"])
    rows = g["rowidx"]
    for D in (int(d) for d in g["dims"]):
        x = frozen_randn(SEED + 7 + D, (3, G, L, D)) * 0.5
        out = dense_causal_attention(x, compute_dtype=torch.float32)
        np.testing.assert_allclose(out[:, rows, :].double().numpy(), g[f"D{D}_f32_rows"], atol=1e-5, rtol=1e-5)
        assert abs(out.double().sum().item() - float(g[f"D{D}_f32_sum"])) < 1e-2
        for dt_name, dt in (("bf16", torch.bfloat16), ("f16", torch.float16)):
            o16 = dense_causal_attention(x.to(dt))      # same-dtype oracle == the reference's 16-bit CPU path, bit for bit
            assert torch.equal(o16[:, rows, :].double(), torch.from_numpy(g[f"D{D}_{dt_name}_rows"]))


def test_heuristic_table(golden_dir):
    meta = json.loads((golden_dir / "g6_meta.json").read_text())
    for key, (nq, nk) in meta["heuristic"].items():
        L, D = (int(x) for x in key.split(","))
        assert sdpa_chunk_heuristic(L, D) == (nq, nk)


def test_lse_definition():
    qkv = frozen_randn(3, (3, 2, 33, 8)).double()
    out, lse = dense_causal_attention_lse(qkv, scale=0.5)
    q, k, _ = qkv
    s = (q @ k.transpose(1, 2)) * 0.5
    mask = torch.ones(33, 33, dtype=torch.bool).tril()
    ref = torch.logsumexp(s.masked_fill(~mask, float("-inf")), dim=-1)
    torch.testing.assert_close(lse, ref, atol=1e-10, rtol=0)


def test_dropout_generator_restatement_statistics():
    """oracle/dropout.py restates the kernels' counter-based dropout generator (csrc/fa_extras.h): the keep rate is 1 - p, the
    decisions of neighbouring rows / keys / groups / offsets are uncorrelated, the function is deterministic and the 65536-window
    key makes positions beyond 2^16 differ from their low-16-bit aliases."""
    import numpy as np

    from oracle import dropout_keep, dropout_key0, dropout_threshold

    assert dropout_threshold(0.0) == 0 and dropout_threshold(0.5) == 2 ** 31 and dropout_threshold(0.999999999) == 2 ** 32 - 1
    assert dropout_key0(1, 2) != dropout_key0(2, 1) != dropout_key0(1, 2 + 2 ** 32)
    for p in (0.1, 0.5, 0.9):
        k = dropout_keep(7, 11, range(8), range(512), range(512), p).numpy()
        n = k.size
        assert abs(k.mean() - (1 - p)) < 4 * np.sqrt(p * (1 - p) / n) + 1e-4, (p, k.mean())
        x = k.astype(np.float64) - k.mean()
        var = (x * x).mean()
        for a, b in ((x[:, :, :-1], x[:, :, 1:]), (x[:, :-1], x[:, 1:]), (x[:-1], x[1:])):       # neighbours along keys, rows, groups
            assert abs((a * b).mean() / var) < 5 / np.sqrt(a.size), (p, (a * b).mean() / var)
        k2 = dropout_keep(7, 12, range(8), range(512), range(512), p).numpy()
        x2 = k2.astype(np.float64) - k2.mean()
        assert abs((x * x2).mean() / var) < 5 / np.sqrt(n)                                           # next offset
        assert np.array_equal(k, dropout_keep(7, 11, range(8), range(512), range(512), p).numpy())
    lo = dropout_keep(3, 4, [0], range(64), range(64), 0.5).numpy()
    hi = dropout_keep(3, 4, [0], range(65536, 65536 + 64), range(64), 0.5).numpy()
    hi2 = dropout_keep(3, 4, [0], range(64), range(65536, 65536 + 64), 0.5).numpy()
    assert not np.array_equal(lo, hi) and not np.array_equal(lo, hi2) and not np.array_equal(hi, hi2)


@pytest.mark.parametrize("L,D,slab,scale", [(96, 16, 32, 1.0), (257, 64, 100, 0.125), (512, 128, 2048, 1.0)])
def test_blocked_gradient_oracle_equals_autograd_through_the_dense_oracle(L, D, slab, scale):
    """dense_causal_attention_grads_blocked (the fp64 gradient checker for the reference's large test sizes, t_fa.py:56-93) is the
    same function as autograd through the pinned dense restatement of fa.py:48-53 - ragged slabs, one slab, scaled and unscaled."""
    from oracle import dense_causal_attention_grads_blocked, frozen_randn
    from oracle.dense_attention import MASK_VALUE

    x = (frozen_randn(99 + L, (3, L, D)) * 0.5).double().requires_grad_(True)
    dout = frozen_randn(98 + L, (L, D)).double()
    q, k, v = x
    mask = torch.triu(torch.full((L, L), MASK_VALUE, dtype=torch.float64), 1)
    out = torch.softmax((q @ k.T) * scale + mask, dim=-1) @ v
    out.backward(dout)
    o2, dq, dk, dv = dense_causal_attention_grads_blocked(q.detach(), k.detach(), v.detach(), dout, scale=scale, slab=slab)
    torch.testing.assert_close(o2, out.detach(), atol=1e-12, rtol=1e-10)
    for got, want in zip((dq, dk, dv), x.grad):
        torch.testing.assert_close(got, want, atol=1e-11, rtol=1e-9)
