"""Forward and forward + backward time at the wide head dims (192 / 256 and a zero-extended 160) and at 128, this repo against stock torch SDPA:
    python scripts/bench_head_dims.py"""
import sys, pathlib
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "flash-attention-ipu_amd")]
import torch, torch.nn.functional as F
from flash_attention_amd._impl import ops
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best
for (B, H, L, D) in ((8, 16, 2048, 256), (4, 8, 8192, 256), (8, 16, 2048, 192), (8, 16, 2048, 160), (2, 8, 1024, 256), (8, 16, 4096, 128), (8, 16, 4096, 64)):
    for dt in (torch.bfloat16, torch.float16):
        q, k, v = (torch.randn(B, H, L, D, device="cuda").to(dt).requires_grad_(True) for _ in range(3))
        g = torch.randn(B, H, L, D, device="cuda").to(dt)
        sc = D ** -0.5
        ours = lambda: ops.attention(q.view(B * H, L, D), k.view(B * H, L, D), v.view(B * H, L, D), sc, 1, 1, True)
        stock = lambda: F.scaled_dot_product_attention(q, k, v, is_causal=True)
        with torch.no_grad():
            to, ts = t(ours), t(stock)
        fl = 2.0 * B * H * L * L * D
        def fb(f):
            o = f(); o.backward(g.view_as(o)); q.grad = k.grad = v.grad = None
        tob, tsb = t(lambda: fb(ours)), t(lambda: fb(stock))
        print(f"[{B},{H},{L},{D}] {dt}: fwd ours {to*1e3:.1f} us ({fl/to/1e9:.0f} TF) stock {ts*1e3:.1f} us ({fl/ts/1e9:.0f} TF); fwd+bwd ours {tob*1e3:.1f} us stock {tsb*1e3:.1f} us", flush=True)
