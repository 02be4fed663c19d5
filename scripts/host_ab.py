"""Developer aid (round 6): SAME-BOX A/B of the override's eager training call - the lean [B, H, L, D] path (ops._BhldAttentionFn) against the path it
replaced (reshape to [G, L, D] -> ops._AttentionFn -> view back; still in the tree: 3-D callers and torch.compile use it) and against stock SDPA.
Host-bound sizes, so the figures depend on the box's host CPU: only rows of ONE run compare.  Rounds are interleaved.
    python scripts/host_ab.py [out.md]"""
import sys, pathlib, time, statistics
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "flash-attention-ipu_amd")]
import torch, torch.nn.functional as F
import flash_attention_amd.auto  # noqa
from flash_attention_amd._impl import ops
patched = F.scaled_dot_product_attention
stock = patched.__wrapped__
lean = ops.attention_bhld_train


def old_path(q, k, v, scale, nq, nk, causal):
    B, H, L, D = q.shape
    return ops.attention(q.reshape(-1, L, D), k.reshape(-1, L, D), v.reshape(-1, L, D), scale, nq, nk, causal).view(B, H, L, D)


def per_call_us(fn, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


rows = ["| forward + backward per call, us (bf16 causal, one box, interleaved rounds, median of 7) | replaced path | lean path | stock SDPA |", "|---|---|---|---|"]
for tag, (B, H, L, D), layout in (("tiny [2,4,64,64], contiguous leaves", (2, 4, 64, 64), "leaves"),
                                  ("GPT-2 layer [8,12,1024,64], contiguous leaves (the vs_stock table's call)", (8, 12, 1024, 64), "leaves"),
                                  ("GPT-2 layer [16,12,1024,64], contiguous leaves", (16, 12, 1024, 64), "leaves"),
                                  ("GPT-2 layer [8,12,1024,64], head-inner slices of a fused [B,T,3C] projection (nanoGPT)", (8, 12, 1024, 64), "fused"),
                                  ("char-GPT layer [64,6,256,64], head-inner slices of a fused projection", (64, 6, 256, 64), "fused")):
    g = torch.randn(B, H, L, D, device="cuda", dtype=torch.bfloat16)
    if layout == "leaves":
        q, k, v = (torch.randn(B, H, L, D, device="cuda", dtype=torch.bfloat16, requires_grad=True) for _ in range(3))
        def fb(f):
            o = f(q, k, v, is_causal=True); o.backward(g); q.grad = k.grad = v.grad = None
    else:
        proj = torch.randn(B, L, 3 * H * D, device="cuda", dtype=torch.bfloat16, requires_grad=True)
        def fb(f):
            qq, kk, vv = (t.view(B, L, H, D).transpose(1, 2) for t in proj.split(H * D, dim=2))
            o = f(qq, kk, vv, is_causal=True); o.backward(g); proj.grad = None
    res = {"old": [], "lean": [], "stock": []}
    for _ in range(300): fb(patched)          # clock and caches up
    for rnd in range(7):
        for name in ("old", "lean", "stock"):
            ops.attention_bhld_train = old_path if name == "old" else lean
            f = stock if name == "stock" else patched
            for _ in range(20): fb(f)
            res[name].append(per_call_us(lambda: fb(f), 300))
    ops.attention_bhld_train = lean
    med = {k: statistics.median(v) for k, v in res.items()}
    rows.append(f"| {tag} | {med['old']:.1f} | **{med['lean']:.1f}** | {med['stock']:.1f} |")
    print(rows[-1], flush=True)
if len(sys.argv) > 1:
    pathlib.Path(sys.argv[1]).write_text("\n".join(rows) + "\n")
