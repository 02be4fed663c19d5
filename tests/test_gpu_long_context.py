"""A quarter of a million rows in one group (GPU, -m gpu): four times the reference notebook's longest demonstrated context (65 536,
tests/test_gpu_parity.py) - forward, logsumexp, dQ rows and dK / dV columns against fp64 expressions computed directly
(scripts/long_context_check.py; the one-off run at L = 2^20 is recorded in profiles/r06_long).  The op's contract is fa.py:48-53; what is
exercised is index arithmetic, buffer extents and grids of thousands of row blocks that no other parity test reaches."""

import pathlib
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent / "scripts"))


@pytest.mark.parametrize("D,dtype", [(64, torch.bfloat16), (128, torch.float16)])
def test_one_group_of_262144_rows_forward_and_backward(D, dtype):
    from long_context_check import check

    msgs = []
    assert check(1 << 18, D, dtype, log=msgs.append), "\n".join(msgs)
