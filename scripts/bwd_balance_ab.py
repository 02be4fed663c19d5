"""Developer aid (round 6): balanced dQ units (fa_bwd_kernels.hip BAL, debug option bbal: 0 off, 1 forced, -1 the library's rule) against the plain
grid, whole fa_bwd, interleaved in-process.   usage: python scripts/bwd_balance_ab.py [out.txt]"""
import sys, pathlib, statistics
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "flash-attention-ipu_amd")]
import torch
from flash_attention_amd._impl import _native
dev = torch.device("cuda")
lines = []
def say(s):
    print(s, flush=True); lines.append(s)
def t_us(fn, n=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
def run(G, L, D, dtype=torch.bfloat16):
    torch.manual_seed(G * 7 + L)
    q, k, v, g = (torch.randn(1, G, L, D, device=dev).to(dtype) * 0.5 for _ in range(4))
    out = torch.empty_like(q); lse = torch.empty(G, L, device=dev, dtype=torch.float32)
    st = lambda t: (t.stride(0), t.stride(1), t.stride(2))
    _native.fwd_strided(q, k, v, out, lse, G, L, D, G, st(q), st(k), st(v), st(out), 1, 1, D ** -0.5, True)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    def call(): _native.bwd(q, k, v, out, g, lse, dq, dk, dv, G, L, D, G, D ** -0.5, True)
    res = {}
    for _ in range(5): call()
    for rnd in range(7):
        for b in (0, 1, -1):
            with _native.debug_options(bbal=b):
                call(); call(); res.setdefault(b, []).append(t_us(call))
    m = {b: statistics.median(x) for b, x in res.items()}
    say(f"[{G},{L},{D}] plain grid {m[0]:8.1f} us   balanced dQ units forced {m[1]:8.1f} us (x{m[0] / m[1]:.3f})   the rule {m[-1]:8.1f} us (x{m[0] / m[-1]:.3f})")
for G, L, D in [(4, 8192, 128), (2, 8192, 128), (1, 16384, 128), (8, 8192, 128), (6, 8192, 128), (8, 4096, 128), (1, 8192, 64), (2, 16384, 64), (1, 32768, 64), (4, 4096, 64), (8, 2048, 64),
                (16, 2048, 64), (16, 1024, 64), (32, 1024, 64), (8, 4096, 64), (12, 4096, 64), (16, 4096, 64), (1, 2048, 64), (128, 4096, 64)]:
    run(G, L, D)
if len(sys.argv) > 1:
    pathlib.Path(sys.argv[1]).write_text("\n".join(lines) + "\n")
