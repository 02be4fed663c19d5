"""CPU tests of the host-side mirror of the reference interface and of the C-ABI library
(loads, exports every declared symbol, validates arguments).  No GPU compute calls."""

import ctypes
import os
import json
import math
import pathlib
import re
import subprocess

import numpy as np
import pytest
import torch
import torch.nn.functional as F

ROOT = pathlib.Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def fa():
    import flash_attention_amd

    return flash_attention_amd


@pytest.fixture(scope="module")
def native():
    from flash_attention_amd._impl import _native

    return _native


@pytest.fixture(scope="module")
def meta(golden_dir):
    return json.loads((golden_dir / "g6_meta.json").read_text())


def test_library_exports_every_declared_symbol(native):
    header = (ROOT / "include" / "flash_attention_amd.h").read_text()
    body = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(fa_[a-z_0-9]+)\s*\(", body))
    assert {"fa_fwd", "fa_fwd_qkv_packed", "fa_plan", "fa_abi_version", "fa_last_error",
            "fa_error_string", "fa_infer_output_shape"} <= declared
    lib = native.load_native_library()
    for sym in declared:
        assert hasattr(lib, sym), f"{sym} declared in include/flash_attention_amd.h but not exported"
    assert lib.fa_abi_version() == 7
    assert native.library_path().name == "libflash_attention.so"


def test_header_is_plain_c(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include "flash_attention_amd.h"\nint main(void){fa_plan_t p; (void)p; return FA_ABI_VERSION - 7;}\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", str(ROOT / "include"), "-c", str(src), "-o", str(tmp_path / "t.o")])


def test_c_abi_under_address_and_ub_sanitizers(tmp_path):
    """SURVEY 5 (sanitizers; the reference only has -Werror, Makefile:10): the C ABI's validation and plan-selection
    code built host-only with -fsanitize=address,undefined against stub launchers, driven through every error path,
    extreme sizes and concurrent callers (tests/cpp/test_capi_host.cpp).  CPU only - never on the GPU."""
    exe = tmp_path / "test_capi_host_asan"
    res = subprocess.run(["make", "-C", str(ROOT / "flash-attention-ipu_amd"), "sanitize", f"SAN_OUT={exe}"],
                         capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and "PASSED" in res.stdout, res.stdout[-3000:] + res.stderr[-3000:]
    assert "runtime error" not in res.stderr and "AddressSanitizer" not in res.stderr


def test_c_abi_argument_validation(native):
    lib = native.load_native_library()
    shape = (ctypes.c_int64 * 4)(3, 8, 128, 64)
    out = (ctypes.c_int64 * 3)()
    assert lib.fa_infer_output_shape(shape, 4, out) == 0 and list(out) == [8, 128, 64]  # FA.cpp:571-577
    assert lib.fa_infer_output_shape(shape, 3, out) == -2
    bad = (ctypes.c_int64 * 4)(2, 8, 128, 64)
    assert lib.fa_infer_output_shape(bad, 4, out) == -2 and b"size 3" in lib.fa_last_error()
    plan = native.FaPlan()
    assert lib.fa_plan(1, 6, 2, 2, 4, 1, 1.0, 1, ctypes.byref(plan)) == -4   # L % nq  (FA.cpp:190)
    assert lib.fa_plan(1, 6, 2, 2, 1, 4, 1.0, 1, ctypes.byref(plan)) == -5   # L % nk  (FA.cpp:191)
    assert lib.fa_plan(0, 6, 2, 2, 1, 1, 1.0, 1, ctypes.byref(plan)) == -2
    assert lib.fa_plan(1, 6, 2, 7, 1, 1, 1.0, 1, ctypes.byref(plan)) == -3
    # a softmax scale the kernels cannot serve (exp2 of scale*log2e*s against a fixed reference max) is refused
    for bad_scale in (0.0, -0.125, float("inf"), float("nan")):
        assert lib.fa_plan(1, 6, 2, 2, 1, 1, bad_scale, 1, ctypes.byref(plan)) == -11
    assert lib.fa_plan(1, 6, 2, 2, 1, 1, 1.0, 2, ctypes.byref(plan)) == -7   # causal must be 0 or 1
    assert lib.fa_fwd_qkv_packed(None, None, None, 1, 8, 64, 1, 1, 1, None) == -1
    assert lib.fa_error_string(-4).decode().endswith("num_chunks_q")


def test_debug_options_are_explicit_and_not_read_from_the_environment_per_call(native, monkeypatch):
    """The tile-plan knobs are process options set through fa_debug_set_option (tests, tuning); the FA_AMD_*
    environment variables are parsed once at first use and never per launch."""
    lib = native.load_native_library()
    auto = native.plan(128, 4096, 64, torch.bfloat16, 8, 8)
    assert auto.source in (0, 1)
    monkeypatch.setenv("FA_AMD_KERNEL", "dma")           # too late: the environment was read at first use
    monkeypatch.setenv("FA_AMD_QB", "1")
    p = native.plan(128, 4096, 64, torch.bfloat16, 8, 8)
    assert (p.kernel, p.qb) == (auto.kernel, auto.qb)
    with native.debug_options(kernel="dma", qb=2):
        p = native.plan(128, 4096, 64, torch.bfloat16, 8, 8)
        assert (p.kernel, p.block_q, p.qb, p.source) == (0, 256, 2, 2)
    with native.debug_options(waves=8):
        assert native.plan(8, 1024, 128, torch.bfloat16, 1, 1).waves == 8
    with native.debug_options(sched=3):
        assert native.plan(128, 4096, 64, torch.bfloat16, 8, 8).sched == 3
    p = native.plan(128, 4096, 64, torch.bfloat16, 8, 8)  # restored
    assert (p.kernel, p.qb, p.sched) == (auto.kernel, auto.qb, auto.sched)
    val = ctypes.c_int32(123)
    assert lib.fa_debug_get_option(b"kernel", ctypes.byref(val)) == 0 and val.value == 0
    assert lib.fa_debug_set_option(b"no_such_knob", 1) == -12
    assert lib.fa_debug_set_option(b"waves", 3) == -12
    assert lib.fa_debug_set_option(b"kernel", ord("z")) == -12


def test_measured_tuning_table_is_consulted_and_bounded_by_the_chunks(native):
    """fa_tuning_table.h (generated by scripts/tune.py --emit on an MI355X) decides where it has a row
    (plan.source == 1); num_chunks_q still bounds the tile; table=0 falls back to the built-in rules."""
    import re

    text = (ROOT / "flash-attention-ipu_amd" / "csrc" / "fa_tuning_table.h").read_text()
    rows = re.findall(r"\{\s*(\d+), (\d), (\d),\s*(\d+),\s*(\d+),\s*(\d+)ll,\s*(\d+)ll, '(\w)', (\d), (\d),\s*(\d+), (\d), (\d)\}", text)
    rows = [r for r in rows if int(r[0]) != 0]
    if not rows:
        pytest.skip("no measured rows in this build")
    assert "generated on:" in text and "CUs" in text and "kernel sources (sha256/16" in text and len(rows) >= 20    # device and source stamps (round 3)
    checked = 0
    for D, skip, causal, L_lo, L_hi, wg_lo, wg_hi, kern, waves, qb, sched, split, pair in rows[::5]:
        split = max(1, int(split or 1))      # (kv split groups: plan.waves counts all of the workgroup's waves)
        D, L_lo, L_hi, wg_lo, wg_hi, waves, qb = int(D), int(L_lo), int(L_hi), int(wg_lo), int(wg_hi), int(waves), int(qb)
        L = 128 * ((max(L_lo, 256) + 127) // 128)
        if L > min(L_hi, 1 << 20):
            continue
        G = 8 * max(1, -(-max(wg_lo, 1) // (8 * (L // 128))))
        if not (wg_lo <= G * (L // 128) <= wg_hi):
            continue
        dt = torch.bfloat16 if skip == "1" else torch.float16
        p = native.plan(G, L, D, dt, 1, 1, causal=causal == "1")
        assert p.source == 1 and (p.kernel, p.waves, p.qb, p.pair) == ({"p": 2, "x": 4, "d": 0, "h": 6}[kern], waves * split, qb, int(pair)), (D, skip, causal, L, G)
        with native.debug_options(table=0):
            assert native.plan(G, L, D, dt, 1, 1, causal=causal == "1").source == 0
        checked += 1
    assert checked >= 5
    # the q chunk still bounds the workgroup's rows whatever the table says
    assert native.plan(128, 4096, 64, torch.bfloat16, 64, 1).block_q <= 64
    assert native.plan(128, 4096, 64, torch.bfloat16, 128, 1).block_q == 32


def test_plan_maps_chunks_to_hardware_tiles(native):
    """The built-in rules (what serves a request that no row of the measured table covers)."""
    with native.debug_options(table=0):
        _check_builtin_rules(native)


def _check_builtin_rules(native):
    # chunk length is an upper bound on the tile; >= 128 rows -> the 4-wave 128-row tile
    p = native.plan(128, 2048, 64, torch.bfloat16, 8, 8)       # L < 4096: 64-key-tile kernel, 128-row workgroups
    assert (p.kernel, p.block_q, p.block_kv, p.waves, p.chunk_q, p.chunk_kv) == (0, 128, 64, 4, 256, 256)
    assert p.grid == 128 * 16 and p.lds_bytes == 2 * 8192 + 4 * (4096 + 64)
    # config 2: heuristic (8,8) -> 512x512 chunks; bf16, D=64, L >= 4096 -> pipelined 32-key steps, 2 query blocks
    # per wave = 256-row workgroups, K/V rings of four 32-key slots
    p = native.plan(128, 4096, 64, torch.bfloat16, 8, 8)
    assert (p.kernel, p.block_q, p.block_kv, p.waves, p.chunk_q, p.chunk_kv) == (2, 256, 32, 4, 512, 512)
    assert p.grid == 128 * 16 and p.lds_bytes == 4 * 4096 + 8 * (2048 + 64) + 16
    p = native.plan(128, 4096, 64, torch.bfloat16, 32, 32)     # 128-row chunks bound the q tile: no 256-row workgroups
    assert (p.kernel, p.block_q, p.block_kv) == (0, 128, 64)
    p = native.plan(128, 4096, 64, torch.float16, 8, 8)        # f16 packed op (scale 1) keeps the per-tile row max
    assert (p.kernel, p.block_q, p.block_kv) == (0, 128, 64)
    p = native.plan(16, 32768, 128, torch.bfloat16, 16, 16)    # config 3: head_dim 128, long, bf16 -> 8 waves = 256 rows
    assert (p.kernel, p.block_q, p.waves, p.lds_bytes) == (0, 256, 8, 2 * 16384 + 8 * (4096 + 64))
    p = native.plan(64, 2048, 128, torch.bfloat16, 8, 8)
    assert (p.kernel, p.block_q, p.waves) == (0, 128, 4)
    # a grid of <= 512 128-row workgroups cannot fill the chip: the kv loop of every q tile is split over 2 | 4 wave groups
    # of the same workgroup (plan.waves counts them all; block_q stays 128)
    p = native.plan(16, 2048, 128, torch.bfloat16, 8, 8)
    assert (p.kernel, p.block_q, p.waves, p.lds_bytes) == (0, 128, 8, 2 * (2 * 16384 + 8 * (4096 + 64)))
    p = native.plan(8, 8192, 64, torch.bfloat16, 8, 8)
    assert (p.kernel, p.block_q, p.waves, p.grid) == (0, 128, 16, 8 * 64)
    # ... at head_dim 64 and L = 2048 .. 4096 by 64-row key-half workgroups with two kv split groups (8 waves)
    p = native.plan(8, 4096, 64, torch.bfloat16, 8, 8)
    assert (p.kernel, p.block_q, p.waves, p.grid) == (6, 64, 8, 8 * 64)
    p = native.plan(8, 1024, 128, torch.bfloat16, 8, 8)
    assert (p.kernel, p.block_q, p.waves) == (0, 128, 8)
    # ... and at head_dim 64 up to L = 2048 by 64-row workgroups whose four waves are (row block, key half) pairs (fa_fwd_kh.hip)
    p = native.plan(8, 1024, 64, torch.bfloat16, 8, 8)
    assert (p.kernel, p.block_q, p.block_kv, p.waves, p.grid, p.lds_bytes) == (6, 64, 64, 4, 8 * 16, 2 * 8192 + 4 * (4096 + 64) + 16)
    assert native.plan(32, 2048, 64, torch.bfloat16, 8, 8).kernel == 6
    assert native.plan(64, 2048, 64, torch.bfloat16, 8, 8).kernel == 0          # 1024 128-row workgroups: the ordinary tiles
    assert native.plan(4, 4096, 64, torch.float16, 64, 64).block_q == 64     # 64-row chunks honoured
    # num_chunks_kv bounds the kv step: a chunk below 64 keys is served by 32-key steps where the pipelined kernel
    # applies (skip-max arithmetic, causal); 64-key tiles otherwise (the result never depends on the chunking)
    p = native.plan(128, 4096, 64, torch.bfloat16, 8, 128)
    assert (p.chunk_kv, p.kernel, p.block_kv) == (32, 2, 32)
    p = native.plan(8, 2048, 64, torch.bfloat16, 8, 64)
    assert (p.chunk_kv, p.kernel, p.block_kv, p.block_q) == (32, 2, 32, 128)
    p = native.plan(8, 2048, 64, torch.bfloat16, 8, 32)
    assert (p.chunk_kv, p.block_kv) == (64, 64)
    p = native.plan(8, 2048, 64, torch.float16, 8, 64)         # unscaled f16 keeps the per-tile max: 64-key tiles
    assert (p.chunk_kv, p.block_kv, p.skipmax) == (32, 64, 0)
    assert native.plan(8, 2048, 64, torch.float16, 8, 64, softmax_scale=0.125).block_kv == 32
    # the non-causal forward (a widening) goes through the same kernels (mask mode is a template parameter)
    p = native.plan(128, 4096, 64, torch.bfloat16, 8, 8, causal=False)
    assert (p.kernel, p.block_q, p.block_kv) == (2, 256, 32)
    assert native.plan(128, 2048, 64, torch.bfloat16, 8, 8, causal=False).kernel == 0
    assert native.plan(4, 4096, 64, torch.float16, 128, 64).block_q == 32    # 32-row chunks honoured
    assert native.plan(5, 200, 64, torch.bfloat16, 1, 1).grid == 8 * 2       # G padded to the 8 XCDs
    # the A/B kernels of earlier rounds (register-staged 'tile', complementary-phase 'pp') left the library with ABI 6: their names are refused
    for gone in ("pp", "tile"):
        with pytest.raises(KeyError):
            native.set_debug_option("kernel", gone)
    assert not hasattr(native.load_native_library(), "fa_experimental_kernels")      # (ABI 7: the link-compatibility stub is gone)
    assert native.plan(4, 6, 2, torch.float32, 3, 3).kernel == 1             # generic kernel
    assert native.plan(4, 256, 64, torch.float32, 1, 1).kernel == 3             # fp32 MFMA kernel (D 64 / 128)
    assert native.plan(4, 256, 80, torch.float32, 1, 1).kernel == 1
    p96 = native.plan(4, 256, 96, torch.bfloat16, 1, 1)      # 16-bit, a multiple of 8 below 128: zero-extended at 128 by the 64-key-tile MFMA kernel
    assert (p96.kernel, p96.block_kv, p96.qb) == (0, 64, 1)
    assert native.plan(4, 256, 100, torch.bfloat16, 1, 1).kernel == 1      # not a multiple of 8: generic
    p256 = native.plan(4, 256, 256, torch.bfloat16, 1, 1)    # head_dim 256 (16-bit), short: the 64-key-tile kernel's 4-wave form, 132 KB of LDS
    p256p = native.plan(64, 2048, 256, torch.bfloat16, 1, 1)  # ... from L = 1024 on (skip-max arithmetic): the pipelined kernel's 4-wave form
    assert (p256p.kernel, p256p.block_q, p256p.block_kv, p256p.waves, p256p.qb, p256p.lds_bytes) == (2, 128, 32, 4, 1, 133136)
    assert native.plan(64, 2048, 256, torch.float16, 1, 1).kernel == 0     # (unscaled f16 keeps the row max: the 64-key-tile kernel)
    assert (p256.kernel, p256.block_q, p256.block_kv, p256.waves, p256.qb, p256.pair) == (0, 128, 64, 4, 1, 0) and p256.lds_bytes == 132096
    assert native.plan(4, 256, 200, torch.float16, 1, 1).kernel == 0       # ... and the multiples of 8 below it, zero-extended
    p192 = native.plan(4, 256, 160, torch.float16, 1, 1)                   # ... 136 .. 192 at head_dim 192: 99 KB of LDS
    assert (p192.kernel, p192.block_q, p192.waves, p192.lds_bytes) == (0, 128, 4, 2 * 64 * 192 * 2 + 2 * 6 * (64 * 64 + 64))
    assert native.plan(4, 256, 256, torch.float32, 1, 1).kernel == 1       # fp32: 64 / 128 only
    assert native.plan(4, 256, 264, torch.bfloat16, 1, 1).kernel == 1


def test_value_errors_match_reference_text(fa, meta):
    msgs = meta["errors"]["ValueError"]
    cases = [(torch.zeros(3, 4, 4), 1, 1), (torch.zeros(2, 1, 4, 4), 1, 1),
             (torch.zeros(3, 1, 6, 4), 4, 1), (torch.zeros(3, 1, 6, 4), 1, 4)]
    for (x, a, b), msg in zip(cases, msgs):
        with pytest.raises(ValueError) as e:
            fa.flash_attention_qkv_packed(x, a, b)
        assert str(e.value) == msg


def test_cpu_branch_matches_golden_config1(fa, golden_dir):
    """BASELINE config 1 (CPU plumbing): 1 head, seq 128, D 64, chunks 4x4."""
    g = np.load(golden_dir / "g1_g2_config1.npz")
    qkv = torch.from_numpy(g["qkv"])
    assert torch.equal(fa.flash_attention_qkv_packed(qkv, 4, 4), torch.from_numpy(g["out_f32"]))
    assert torch.equal(fa.flash_attention_qkv_packed(qkv.half(), 4, 4).float(), torch.from_numpy(g["out_f16"]))
    assert torch.equal(fa.flash_attention_qkv_packed(qkv.bfloat16(), 4, 4).float(), torch.from_numpy(g["out_bf16"]))
    assert fa.__all__ == ["flash_attention_qkv_packed"]


def test_patch_function_contract():
    from flash_attention_amd._impl.utils import patch_function
    import types

    mod = types.SimpleNamespace()

    def orig(x, y=1):
        """doc"""
        return x + y

    mod.orig = orig

    @patch_function(orig, [mod])
    def repl(orig_fn, x, y=1):
        return 10 * orig_fn(x, y)

    assert mod.orig is repl and mod.orig(1, y=2) == 30
    assert mod.orig.__wrapped__ is orig and mod.orig.__name__ == "orig" and mod.orig.__doc__ == "doc"


def test_sdpa_override_guards_and_passthrough(meta):
    orig = F.scaled_dot_product_attention
    orig = getattr(orig, "__wrapped__", orig)
    import flash_attention_amd.auto  # noqa: F401
    from flash_attention_amd._impl.scaled_dot_product_attention import chunk_heuristic

    patched = F.scaled_dot_product_attention
    assert patched.__wrapped__ is orig and patched.__name__ == "scaled_dot_product_attention"
    x = torch.zeros(1, 8, 4)
    msgs = meta["errors"]["NotImplementedError"]
    calls = [dict(attn_mask=torch.ones(1, dtype=torch.bool), is_causal=True), dict(dropout_p=0.1, is_causal=True),
             dict(is_causal=False)]   # (off the GPU is_causal=False raises like the reference; on it, it is served)
    for kw, msg in zip(calls, msgs[:3]):
        with pytest.raises(NotImplementedError) as e:
            patched(x, x, x, **kw)
        assert str(e.value) == msg
    with pytest.raises(NotImplementedError) as e:   # multi-element mask: NotImplementedError, not RuntimeError
        patched(x, x, x, attn_mask=torch.ones(8, 8, dtype=torch.bool), is_causal=True)
    assert str(e.value) == msgs[0]
    with pytest.raises(NotImplementedError) as e:
        patched(x, torch.zeros(1, 4, 4), x, is_causal=True)
    assert str(e.value) == msgs[3]
    with pytest.raises(NotImplementedError) as e:
        patched(x, x, torch.zeros(1, 8, 2), is_causal=True)
    assert str(e.value) == msgs[4]
    # off-device: passthrough to the original, scale kwarg forwarded
    q = torch.randn(2, 3, 16, 8)
    assert torch.equal(patched(q, q, q, is_causal=True), orig(q, q, q, is_causal=True))
    assert torch.equal(patched(q, q, q, is_causal=True, scale=0.3), orig(q, q, q, is_causal=True, scale=0.3))
    # un-patch / re-patch through __wrapped__ as the reference's OOM test does (t_sdpa.py:123-137)
    F.scaled_dot_product_attention = patched.__wrapped__
    assert F.scaled_dot_product_attention is orig
    F.scaled_dot_product_attention = patched
    for key, (nq, nk) in meta["heuristic"].items():
        L, D = (int(v) for v in key.split(","))
        assert chunk_heuristic(L, D) == (nq, nk)


def test_custom_ops_shape_inference_on_meta_tensors():
    """The fake (meta) implementations = the reference's Op::setup() shape inference (FA.cpp:571-577):
    [3,G,L,D] -> [G,L,D]; the [B,H,L,D] inference op keeps a head-inner input layout on its output."""
    from flash_attention_amd._impl import ops

    qkv = torch.empty(3, 5, 200, 64, device="meta", dtype=torch.bfloat16)
    out, lse = torch.ops.flash_attention_amd.fwd_qkv_packed(qkv, 1, 1)
    assert out.shape == (5, 200, 64) and out.dtype == torch.bfloat16 and lse.shape == (5, 200) and lse.dtype == torch.float32
    x = torch.empty(2, 16, 3 * 4 * 8, device="meta", dtype=torch.float16)
    q = x.split(32, dim=2)[1].view(2, 16, 4, 8).transpose(1, 2)          # nanoGPT-style [B,H,T,hs] view
    o = ops.fwd_bhld(q, q, q, 0.35, 1, 1)
    assert o.shape == (2, 4, 16, 8) and o.transpose(1, 2).is_contiguous()
    c = torch.empty(2, 4, 16, 8, device="meta", dtype=torch.float16)
    assert ops.fwd_bhld(c, c, c, 0.35, 1, 1).is_contiguous()
    g2 = torch.empty(2, 2, 16, 8, device="meta", dtype=torch.float16)            # grouped-query K/V: 2 of 4 heads
    assert ops.fwd_bhld(c, g2, g2, 0.35, 1, 1, False).shape == (2, 4, 16, 8)
    dq, dk, dv = torch.ops.flash_attention_amd.bwd(c[0], c[0], c[0], c[0], c[0], torch.empty(4, 16, device="meta"), 1.0)
    assert dq.shape == dk.shape == dv.shape == (4, 16, 8)


def test_missing_library_is_an_import_error(monkeypatch, tmp_path):
    """No silent fallback: without the .so the package refuses to import (init.py:20 behaviour)."""
    from flash_attention_amd._impl import _native

    monkeypatch.setattr(_native, "_LIB", None)
    monkeypatch.setattr(_native, "candidate_paths", lambda: [tmp_path / "nope.so"])
    with pytest.raises(ImportError, match="Cannot find libflash_attention.so"):
        _native.load_native_library()


def test_gpu_tensor_without_gpu_fails_loudly(fa):
    if torch.cuda.is_available():
        pytest.skip("needs a CPU-only box")
    with pytest.raises((RuntimeError, AssertionError)):
        fa.flash_attention_qkv_packed(torch.zeros(3, 1, 8, 64).cuda(), 1, 1)


@pytest.mark.parametrize("G,L,D", [(1, 4096, 64), (2, 32768, 64), (3, 2048, 64), (4, 16384, 128), (5, 1024, 64), (6, 8192, 64), (7, 4096, 128),
                                   (9, 2048, 64), (10, 1024, 128), (12, 1024, 64), (20, 2048, 64), (25, 4096, 64), (12, 128, 64), (100, 1024, 64), (96, 1024, 64)])
def test_q_tile_subsets_cover_every_tile_once(native, G, L, D):
    """Groups go to the 8 XCDs by index mod 8; a group count that wastes an eighth of them or more is laid out as G * xs (group, q-tile subset) pairs
    (fa_plan_t.xs).  The kernels' block map (csrc/fa_fwd_*.hip: XCD-major slots, bunches of `sched` groups, heaviest tile first, then
    FwdParams::xs) is restated here from the plan's fields: every (group, q tile) must be reached by exactly one workgroup."""
    p = native.plan(G, L, D, torch.bfloat16, 1, 1)
    g8 = (G + 7) // 8 * 8
    want_xs = 1      # the smallest of 2, 4, 8 that wastes < 9 % of the XCDs, where the pairs of an XCD still share its L2 (fa_capi.cpp subsets_for)
    if (g8 - G) * 8 >= g8:
        kv = 2 * L * D * 2
        for xs_ in (2, 4, 8):
            gv8 = (G * xs_ + 7) // 8 * 8
            if (gv8 - G * xs_) * 100 < 9 * gv8:
                want_xs = xs_ if (L + 255) // 256 >= xs_ and (kv >= 4 << 20 or gv8 // 8 * kv <= 8 << 20) else 1
                break
    assert p.xs == want_xs
    nq_real = (L + p.block_q - 1) // p.block_q
    xs = p.xs
    Gv = G * xs if xs > 1 else G
    G8 = (Gv + 7) // 8 * 8
    nq = (nq_real + xs - 1) // xs if xs > 1 else nq_real
    if p.pair:
        pytest.skip("pairs have a map of their own")
    assert p.grid == G8 * nq
    seen = {}
    for b in range(p.grid):
        xcd, slot = b & 7, b >> 3
        ng = G8 >> 3
        per_bunch = p.sched * nq
        bunch, rem = divmod(slot, per_bunch)
        g0 = bunch * p.sched
        bc = min(ng - g0, p.sched)
        qd = rem // bc
        g = (g0 + (rem - qd * bc)) * 8 + xcd
        if g >= Gv:
            continue
        if xs > 1:
            s, g = g % xs, g // xs
            qd = qd * xs + s
            if qd >= nq_real:
                continue
        key = (g, nq_real - 1 - qd)
        assert key not in seen, key
        seen[key] = b
    assert len(seen) == G * nq_real and all(0 <= g < G and 0 <= q < nq_real for g, q in seen)
    if xs > 1:      # every XCD gets the same number of (group, subset) pairs when G * xs is a multiple of 8
        per_xcd = [sum(1 for (g, q), b in seen.items() if b & 7 == x) for x in range(8)]
        assert max(per_xcd) - min(per_xcd) <= (nq_real + xs - 1) // xs, per_xcd


def test_no_shipped_kernel_uses_scratch_memory():
    """The Makefile compiles every kernel object with -Rpass-analysis=kernel-resource-usage and fails the build on a kernel that needs
    scratch (csrc register budgets are set so that none does: a spilled register in an attention loop is a reload behind a vmcnt(0)).
    The remarks it leaves beside the objects are read back here: every instantiation, zero bytes."""
    import pathlib
    import re

    obj = pathlib.Path(__file__).resolve().parent.parent / "flash-attention-ipu_amd" / "build" / "obj"
    remarks = sorted(obj.glob("*.remarks"))
    if not remarks:
        pytest.skip("no build/obj/*.remarks (the library was not built by this Makefile here)")
    kernels = {}
    for f in remarks:
        name = None
        for line in f.read_text().splitlines():
            m = re.search(r"Function Name: (\S+)", line)
            if m:
                name = m.group(1)
            m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
            if m and name:
                kernels[name] = int(m.group(1))
    assert len(kernels) >= 150, len(kernels)
    assert {k: v for k, v in kernels.items() if v} == {}


def test_bench_reads_the_engine_clock_from_the_supported_sclk_levels():
    """`rocm-smi -s` as an idle MI355X prints it: the engine clock bench.py reports is the highest SUPPORTED sclk level (2 400), not a
    current speed (an idle device runs at 96 MHz) and not the memory clock (2 000)."""
    import importlib.util
    import pathlib

    spec = importlib.util.spec_from_file_location("bench_for_test", pathlib.Path(__file__).resolve().parent.parent / "bench.py")
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    text = ("GPU[0]\t\t: Supported fclk frequencies on GPU0\nGPU[0]\t\t: 0: 1250Mhz *\nGPU[0]\t\t: Supported mclk frequencies on GPU0\n"
            "GPU[0]\t\t: 0: 2000Mhz *\nGPU[0]\t\t: Supported sclk frequencies on GPU0\nGPU[0]\t\t: S: 96Mhz *\nGPU[0]\t\t: 0: 500Mhz\n"
            "GPU[0]\t\t: 1: 2400Mhz\nGPU[0]\t\t: Supported socclk frequencies on GPU0\nGPU[0]\t\t: S: 38Mhz *\nGPU[0]\t\t: 0: 1200Mhz\n")
    assert bench.parse_max_sclk(text) == 2400
    assert bench.parse_max_sclk("no such block") is None


def _load_bench():
    import importlib.util
    import pathlib

    spec = importlib.util.spec_from_file_location("bench_for_test", pathlib.Path(__file__).resolve().parent.parent / "bench.py")
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    return bench


@pytest.mark.parametrize("var,value", [("LD_PRELOAD", "/opt/rocm/lib/librocprofiler-sdk-tool.so"), ("LD_PRELOAD", "/some/other/tool.so"),
                                       ("HSA_TOOLS_LIB", "libsomething.so"), ("ROCP_TOOL_LIBRARIES", "x"), ("ROCPROFILER_REGISTER_FORCE_LOAD", "1"),
                                       ("ROCPROF_COUNTERS", "pmc: SQ_WAVES"), ("ROCPROFILER_SDK_ANYTHING_NEW", "1"), ("ROCTRACER_DOMAIN", "hip")])
def test_bench_never_spawns_a_child_when_a_tool_is_preloaded(monkeypatch, tmp_path, var, value):
    """Round-5 verdict, item 5: under rocprofv3 (whose preloaded library initialises the GPU before main()) bench.py must not fork + exec
    `rocm-smi` - the GPU box refuses that exec.  Whatever the preload or tool variable is called, no subprocess is started; the engine clock
    then comes from sysfs (a file read) or is None."""
    import subprocess

    bench = _load_bench()
    for k in list(os.environ):
        if k in ("LD_PRELOAD", "HSA_TOOLS_LIB") or k.startswith(("ROCP_", "ROCPROFILER_", "ROCPROF_", "ROCTRACER_")):
            monkeypatch.delenv(k)
    assert not bench.tool_preloaded()
    monkeypatch.setenv(var, value)
    assert bench.tool_preloaded()

    def refuse(*a, **kw):
        raise AssertionError(f"bench.py started a child process with {var} set: {a}")

    monkeypatch.setattr(subprocess, "run", refuse)
    monkeypatch.setattr(subprocess, "Popen", refuse)
    monkeypatch.setattr(bench, "sysfs_max_sclk_mhz", lambda index: None)      # (no sysfs clock file: the rocm-smi branch would be next)
    assert bench.smi_clock_mhz(0) is None


def test_bench_engine_clock_from_sysfs_needs_no_child_process(monkeypatch, tmp_path):
    import subprocess

    bench = _load_bench()
    for i, text in ((1, "0: 500Mhz\n1: 2400Mhz *\n"), (0, "0: 500Mhz *\n1: 2100Mhz\n")):
        d = tmp_path / f"card{i}" / "device"
        d.mkdir(parents=True)
        (d / "pp_dpm_sclk").write_text(text)
    (tmp_path / "card2" / "device").mkdir(parents=True)       # (a display adapter without the file is skipped)
    assert bench.sysfs_max_sclk_mhz(0, root=str(tmp_path)) == 2100 and bench.sysfs_max_sclk_mhz(1, root=str(tmp_path)) == 2400
    assert bench.sysfs_max_sclk_mhz(2, root=str(tmp_path)) is None and bench.sysfs_max_sclk_mhz(0, root=str(tmp_path / "absent")) is None
    monkeypatch.setattr(bench, "sysfs_max_sclk_mhz", lambda index: 2400)
    monkeypatch.setattr(subprocess, "run", lambda *a, **kw: (_ for _ in ()).throw(AssertionError("no child process when sysfs answers")))
    assert bench.smi_clock_mhz(0) == 2400


def test_design_md_quotes_the_tree(native):
    """Round-5 verdict, item 7: DESIGN.md drifted from the tree (a pass count two rounds old, rocprofv3's register field quoted as the kernel's
    register count).  Every `GPUTEST_rNN.json`: **N passed** it quotes must be that driver record's count, the count it gives for the final
    sources must not be below the newest driver record (tests are only added), and the headline kernel's VGPR figure must be the build's."""
    design = (ROOT / "DESIGN.md").read_text()
    quoted = re.findall(r"`GPUTEST_r(\d+)\.json`: \*\*([\d ]+) passed", design)
    assert quoted, "DESIGN.md section 6 names the driver's GPU test record it quotes"
    newest = 0
    for rnd, count in quoted:
        rec = ROOT / f"GPUTEST_r{rnd}.json"
        if rec.exists():
            passed = json.loads(rec.read_text())["gpu_tests_passed"]
            assert int(count.replace(" ", "")) == passed, f"DESIGN.md quotes {count} for GPUTEST_r{rnd}.json, the record says {passed}"
            newest = max(newest, passed)
    for rec in ROOT.glob("GPUTEST_r*.json"):
        newest = max(newest, json.loads(rec.read_text()).get("gpu_tests_passed") or 0)
    final = re.search(r"final sources:\s+\*\*([\d ]+) passed", design)
    assert final and int(final.group(1).replace(" ", "")) >= newest
    m = re.search(r"(\d+) VGPRs \(the build's resource remark", design)
    assert m, "DESIGN.md section 3.1 quotes the headline kernel's register count from the build"
    remarks = ROOT / "flash-attention-ipu_amd" / "build" / "obj" / "fa_fwd_pipe.remarks"
    if remarks.exists():
        text = remarks.read_text()
        at = text.index("fa_fwd_pipe_kernelIDF16bLi64ELi4ELi2ELb1EE")
        vgprs = int(re.search(r"VGPRs: (\d+)", text[at:]).group(1))
        assert int(m.group(1)) == vgprs, f"DESIGN.md says {m.group(1)} VGPRs for the headline kernel, the build says {vgprs}"


def test_eager_training_path_helpers_keep_the_operand_layout_and_cache_stride_blocks(native):
    """Round 6 (host cost of an eager forward + backward): `ops._dense_like` allocates results and gradients in the LAYOUT of the operand
    (a head-inner [B, L, H, D] view stays head-inner, three gradients come as one allocation), and `_native._stride_block` builds the ctypes
    stride block of a layout once."""
    from flash_attention_amd._impl import ops

    B, H, L, D = 2, 3, 5, 8
    contiguous = torch.empty(B, H, L, D)
    head_inner = torch.empty(B, L, 3 * H * D).split(H * D, dim=2)[1].view(B, L, H, D).transpose(1, 2)      # nanoGPT's k
    assert head_inner.stride() == (L * 3 * H * D, D, 3 * H * D, 1)
    a, b = ops._dense_like(contiguous), ops._dense_like(head_inner)
    assert a.shape == b.shape == (B, H, L, D) and a.is_contiguous()
    assert b.stride() == (L * H * D, D, H * D, 1) and b.transpose(1, 2).is_contiguous()      # head-inner in, head-inner out (dense)
    g3 = ops._dense_like(head_inner, 3)
    assert g3.shape == (3, B, H, L, D) and g3.stride() == (B * H * L * D, L * H * D, D, H * D, 1)
    dq, dk, dv = g3.unbind(0)
    assert dq.stride() == b.stride() and dk.data_ptr() - dq.data_ptr() == B * H * L * D * 4
    # one ctypes block per distinct layout, the same object on the next call; forward blocks (3 strides) and backward blocks (8 x 3) do not collide
    blk = native._stride_block(head_inner.stride(), 1)
    assert blk is native._stride_block(head_inner.stride(), 1) and list(blk) == [L * 3 * H * D, D, 3 * H * D]
    key = (contiguous.stride(),) * 8
    arr, ptr = native._stride_block(key, 8)
    assert native._stride_block(key, 8)[0] is arr and list(arr) == [H * L * D, L * D, D] * 8
    assert native._stride_block(contiguous.stride(), 1) is not arr
