"""CPU oracle for the flash_attention_qkv_packed forward path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package
(`flash-attention-ipu_amd/`) may import, link or execute anything from this
directory.  Allowed users: `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py`.

Parity status: PINNED.  The restatements here were checked in the build
container against the reference's own Python (imported from /root/reference,
see tests/golden/generate_golden.py) and against the committed golden vectors
under tests/golden/ (tests/test_oracle.py re-checks the latter on every run).
"""

from .dense_attention import (  # noqa: F401
    dense_causal_attention,
    dense_causal_attention_lse,
    dense_causal_attention_rows,
    dense_causal_attention_grads_blocked,
    dense_attention_lse,
    sdpa_chunk_heuristic,
    frozen_randn,
)
from .dropout import dense_attention_extra, dropout_keep, dropout_key0, dropout_threshold  # noqa: F401,E402
