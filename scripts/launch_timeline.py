"""Developer aid: per-launch durations of N back-to-back launches (does throughput sag under sustained load?).
usage: python scripts/launch_timeline.py G L D N [env assignments like FA_AMD_KERNEL=dma]"""
import os, sys, pathlib
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "flash-attention-ipu_amd")]
G, L, D, N = map(int, sys.argv[1:5])
for kv in sys.argv[5:]:
    k, v = kv.split("=")
    os.environ[k] = v
import torch
import flash_attention_amd as fa
qkv = torch.randn(3, G, L, D, device="cuda").to(torch.bfloat16)
fa.flash_attention_qkv_packed(qkv, 8, 8)
torch.cuda.synchronize()
import time
time.sleep(0.5)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
ev[0].record()
for i in range(N):
    fa.flash_attention_qkv_packed(qkv, 8, 8)
    ev[i + 1].record()
torch.cuda.synchronize()
ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(N)]
fl = 2.0 * G * L * L * D / 1e12
print(" ".join(sys.argv[5:]) or "default", "TF per launch:", " ".join(f"{fl / (m * 1e-3):.0f}" for m in ms[::max(1, N // 24)]))
