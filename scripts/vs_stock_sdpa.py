"""Developer aid: the override against stock torch SDPA (ROCm fused backend) on the same box, forward and forward + backward.

usage: python scripts/vs_stock_sdpa.py [out.md]
`import flash_attention_amd.auto` patches F.scaled_dot_product_attention; `.__wrapped__` is stock SDPA.  Each timing = the best of 3 rounds of
N back-to-back calls between two events (N sized for >= 20 ms per round, at most 400).  FLOPs: 4 B H L^2 D (halved for causal), x 3.5 for forward + backward (the usual 2 + 5 GEMM convention / 2).
"""
import sys, pathlib
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "flash-attention-ipu_amd")]
import torch
import torch.nn.functional as F
import flash_attention_amd.auto  # noqa: F401

ours, stock = F.scaled_dot_product_attention, F.scaled_dot_product_attention.__wrapped__


def best(fn, n):
    fn()
    torch.cuda.synchronize()
    t = float("inf")
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        t = min(t, e0.elapsed_time(e1) / n)
    return t


CASES = [  # name, B, H, Hkv, L, D, dtype, causal
    ("cfg2 shape (BASELINE config 2 through the override)", 8, 16, 16, 4096, 64, torch.bfloat16, True),
    ("cfg2 shape, f16", 8, 16, 16, 4096, 64, torch.float16, True),
    ("cfg3 shape (BASELINE config 3)", 1, 16, 16, 32768, 128, torch.bfloat16, True),
    ("GPT-2 small layer (BASELINE config 4), B = 8", 8, 12, 12, 1024, 64, torch.bfloat16, True),
    ("GPT-2 small layer, B = 16", 16, 12, 12, 1024, 64, torch.bfloat16, True),
    ("cfg5 shard shape", 8, 32, 32, 8192, 64, torch.bfloat16, True),
    ("head_dim 128, L = 2048", 8, 16, 16, 2048, 128, torch.bfloat16, True),
    ("head_dim 128, L = 4096", 4, 32, 32, 4096, 128, torch.bfloat16, True),
    ("head_dim 128, L = 8192", 2, 32, 32, 8192, 128, torch.bfloat16, True),
    ("grouped-query 32 / 8 heads, head_dim 128", 4, 32, 8, 4096, 128, torch.bfloat16, True),
    ("unmasked (is_causal=False), head_dim 64", 8, 16, 16, 4096, 64, torch.bfloat16, False),
    ("unmasked, head_dim 128", 4, 16, 16, 4096, 128, torch.bfloat16, False),
    ("head_dim 96", 8, 16, 16, 2048, 96, torch.bfloat16, True),
    ("head_dim 192", 8, 16, 16, 2048, 192, torch.bfloat16, True),
    ("head_dim 256", 8, 16, 16, 2048, 256, torch.bfloat16, True),
]
rows = ["| call (q [B, H, L, D]) | forward: override / stock SDPA | TFLOP/s | forward + backward: override / stock SDPA | TFLOP/s |", "|---|---|---|---|---|"]
for name, B, H, Hkv, L, D, dt, causal in CASES:
    gen = torch.Generator(device="cuda").manual_seed(0)
    q = torch.randn(B, H, L, D, device="cuda", generator=gen).to(dt).requires_grad_(True)
    k, v = (torch.randn(B, Hkv, L, D, device="cuda", generator=gen).to(dt).requires_grad_(True) for _ in range(2))
    g = torch.randn(B, H, L, D, device="cuda", generator=gen).to(dt)
    kw = dict(is_causal=causal)
    if Hkv != H:
        kw["enable_gqa"] = True
    fl = 4.0 * B * H * L * L * D * (0.5 if causal else 1.0)
    n = max(3, min(400, int(2e-2 / (fl / 6e14))))      # (>= 20 ms per round where the call allows: a 16-call round of a 0.1 ms call reads its own ramp)

    def fwd_bwd(f):
        o = f(q, k, v, **kw)
        o.backward(g)
        q.grad = k.grad = v.grad = None

    with torch.no_grad():
        tf_o, tf_s = best(lambda: ours(q, k, v, **kw), n), best(lambda: stock(q, k, v, **kw), n)
    tb_o, tb_s = best(lambda: fwd_bwd(ours), max(3, n // 2)), best(lambda: fwd_bwd(stock), max(3, n // 2))
    rows.append(f"| {name}: [{B}, {H}{'/' + str(Hkv) if Hkv != H else ''}, {L}, {D}] {str(dt).split('.')[-1]} | **{tf_o * 1e3:.0f} us** / {tf_s * 1e3:.0f} us ({tf_s / tf_o:.2f}x) | "
                f"{fl / tf_o / 1e9:.0f} / {fl / tf_s / 1e9:.0f} | **{tb_o * 1e3:.0f} us** / {tb_s * 1e3:.0f} us ({tb_s / tb_o:.2f}x) | {3.5 * fl / tb_o / 1e9:.0f} / {3.5 * fl / tb_s / 1e9:.0f} |")
    print(rows[-1], flush=True)
    del q, k, v, g
text = "\n".join(rows) + "\n"
if len(sys.argv) > 1:
    pathlib.Path(sys.argv[1]).write_text(text)
