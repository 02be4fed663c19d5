"""Timing-only probe (results of the second form are not attention of distinct groups): how much of a many-group short-L launch is the re-fetch of
K / V?  The same launch twice - K / V per group as usual, and every group reading ONE group's K / V (group stride 0: 256 KB that never leave the
L2) - Q reads and O writes unchanged.  If the second is much faster, the launch is bound on K / V bytes; if not, on something else.

    python scripts/kv_refetch_probe.py [G,L,D ...]
"""
import pathlib
import statistics
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "flash-attention-ipu_amd")]
import torch  # noqa: E402
from flash_attention_amd._impl import _native  # noqa: E402

shapes = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]] or [(512, 1024, 64), (2048, 1024, 64), (96, 1024, 64), (1024, 512, 64), (256, 2048, 64), (256, 1024, 128)]


def timed(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


warm = torch.randn(3, 128, 4096, 64, device="cuda").bfloat16()
wo = torch.empty(128, 4096, 64, device="cuda", dtype=torch.bfloat16)
st_w = (4096 * 64, 0, 64)
for _ in range(500):
    _native.fwd_strided(warm[0], warm[1], warm[2], wo, None, 128, 4096, 64, 1, st_w, st_w, st_w, st_w, 1, 1, 1.0, True)
torch.cuda.synchronize()
for G, L, D in shapes:
    q, k, v = (torch.randn(G, L, D, device="cuda").bfloat16() for _ in range(3))
    out = torch.empty_like(q)
    st, st0 = (L * D, 0, D), (0, 0, D)
    own = lambda: _native.fwd_strided(q, k, v, out, None, G, L, D, 1, st, st, st, st, 1, 1, D ** -0.5, True)        # noqa: E731
    one = lambda: _native.fwd_strided(q, k, v, out, None, G, L, D, 1, st, st0, st0, st, 1, 1, D ** -0.5, True)      # noqa: E731
    n = max(5, int(8.0 / max(timed(own, 3), 1e-3)))
    a, b = [], []
    for _ in range(5):
        a.append(timed(own, n))
        b.append(timed(one, n))
    ta, tb = statistics.median(a), statistics.median(b)
    fl = 2.0 * G * L * L * D
    print(f"[{G},{L},{D}] bf16 causal: own K/V {ta * 1e3:7.1f} us ({fl / ta / 1e9:6.1f} TFLOP/s)   one group's K/V for all {tb * 1e3:7.1f} us ({fl / tb / 1e9:6.1f})   x{ta / tb:5.3f}", flush=True)
