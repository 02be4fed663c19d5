"""Developer aid: host-side cost of the mask / dropout training path of the override (char-GPT layer shape, nanoGPT layout)."""
import sys, pathlib, time
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "flash-attention-ipu_amd")]
import torch, torch.nn.functional as F
import flash_attention_amd.auto  # noqa
patched = F.scaled_dot_product_attention
stock = patched.__wrapped__


def bench(fn, n=500):
    for _ in range(50): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for tag, (B, H, L, D) in (("char-GPT layer [64,6,256,64]", (64, 6, 256, 64)), ("GPT-2 layer [8,12,1024,64]", (8, 12, 1024, 64)), ("tiny [2,4,64,64]", (2, 4, 64, 64))):
    proj = torch.randn(B, L, 3 * H * D, device="cuda", dtype=torch.bfloat16, requires_grad=True)
    g = torch.randn(B, H, L, D, device="cuda", dtype=torch.bfloat16)
    pad = torch.ones(B, 1, 1, L, device="cuda", dtype=torch.bool); pad[..., L - 7:] = False
    def fb(f, **kw):
        q, k, v = (t.view(B, L, H, D).transpose(1, 2) for t in proj.split(H * D, dim=2))
        o = f(q, k, v, **kw); o.backward(g); proj.grad = None
    print(f"== {tag} (head-inner slices of a fused projection), forward + backward us per call: override / stock")
    for name, kw in (("causal", dict(is_causal=True)), ("causal + dropout 0.2", dict(is_causal=True, dropout_p=0.2)), ("key-padding mask", dict(attn_mask=pad)),
                     ("key-padding mask + dropout 0.1", dict(attn_mask=pad, dropout_p=0.1))):
        print(f"  {name:32s} {bench(lambda: fb(patched, **kw)):8.1f} / {bench(lambda: fb(stock, **kw)):8.1f}", flush=True)
