"""Bounded runs of the differential fuzzers under tests/tools (GPU, -m gpu): every case compares the HIP path with the fp64 oracle
(oracle/dropout.py dense_attention_extra: fa.py:48-53 with torch's mask / dropout semantics) on random shapes the parametrised tests do
not enumerate.  The full-length runs of the round are recorded under profiles/r06_fuzz; the mask fuzzer's 37th case found the per-dword
range-check bug that tests/test_gpu_extra.py::test_mask_view_with_padded_rows_... now pins.
"""

import pathlib
import sys

import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent / "tools"))


def test_fuzz_of_the_mask_bias_and_dropout_kernels():
    """tests/tools/fuzz_extra.py: every mask kind incl. strided views with and without aligned strides, dropout, head dims 64 .. 256, with and
    without the causal mask - out and all three gradients against the oracle, bit-reproducible."""
    from fuzz_extra import fuzz

    msgs = []
    bad = fuzz(36, 6001, log=msgs.append)
    assert bad == 0, "\n".join(m for m in msgs if m.startswith("BAD"))


def test_fuzz_of_the_plain_kernels_on_padded_views():
    """tests/tools/fuzz_strided.py: fa_fwd / fa_bwd on operands, results and gradients that are padded / offset / head-inner views (f16, bf16,
    fp32; MFMA and generic kernels) - out, logsumexp and gradients against the oracle, nothing written outside the views, bit-reproducible."""
    from fuzz_strided import fuzz

    msgs = []
    bad = fuzz(60, 6002, log=msgs.append)
    assert bad == 0, "\n".join(m for m in msgs if m.startswith("BAD"))


def test_fuzz_of_the_sdpa_override_against_stock_sdpa():
    """tests/tools/fuzz_override.py: every routing decision of the `F.scaled_dot_product_attention` override (HIP kernels, zero-extended and padded
    head dims, mask kernels, generic kernel, hand-off to stock) gives stock SDPA's fp64 result and gradients; where it raises, it is the
    reference's chunk guard (fa.py:28-35 via sdpa.py:52-53) and nothing else."""
    from fuzz_override import fuzz

    msgs = []
    bad = fuzz(60, 6003, log=msgs.append)
    assert bad == 0, "\n".join(m for m in msgs if m.startswith("BAD"))
