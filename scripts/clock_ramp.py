import sys, time, pathlib
ROOT = pathlib.Path("/root/repo")
sys.path[:0] = [str(ROOT), str(ROOT / "flash-attention-ipu_amd")]
import torch
import flash_attention_amd as fa
qkv = torch.randn(3, 128, 4096, 64, device="cuda").bfloat16()
fa.flash_attention_qkv_packed(qkv, 8, 8); torch.cuda.synchronize()
time.sleep(2.0)     # let the chip go idle
flops = 2.0 * 128 * 4096 * 4096 * 64
t0 = time.perf_counter()
evs = []
while time.perf_counter() - t0 < (1.5 if "--full" in sys.argv else 8.0):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fa.flash_attention_qkv_packed(qkv, 8, 8)
    e1.record()
    e1.synchronize()
    evs.append((time.perf_counter() - t0, e0.elapsed_time(e1) / 20))
full = "--full" in sys.argv
for t, ms in (evs if full else evs[:12] + evs[12:60:6] + evs[60::60]):
    print(f"t={t*1e3:7.1f} ms  {flops/ms/1e9:7.1f} TFLOP/s")
slow = [(t, flops / ms / 1e9) for t, ms in evs[12:] if flops / ms / 1e9 < 980.0]
print("20-launch groups under 980 TFLOP/s after the first 60 ms:", ", ".join(f"t={t*1e3:.0f} ms {tf:.0f}" for t, tf in slow) or "none")
