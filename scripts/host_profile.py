import sys, pathlib, cProfile, pstats, io
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "flash-attention-ipu_amd")]
import torch, torch.nn.functional as F
import flash_attention_amd.auto  # noqa
patched = F.scaled_dot_product_attention
qg = torch.randn(2, 4, 64, 64, device="cuda", dtype=torch.bfloat16).requires_grad_(True)
def fb():
    o = patched(qg, qg, qg, is_causal=True); o.sum().backward(); qg.grad = None
for _ in range(50): fb()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(300): fb()
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45); print(s.getvalue()[:9000])
