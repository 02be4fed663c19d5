import sys, pathlib, time
sys.path[:0] = ["/root/repo", "/root/repo/flash-attention-ipu_amd"]
import torch, flash_attention_amd as fa
for (G, L, D) in [(8, 2048, 128), (128, 4096, 64)]:
    qkv = torch.randn(3, G, L, D, device="cuda")
    fa.flash_attention_qkv_packed(qkv, 1, 1); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): fa.flash_attention_qkv_packed(qkv, 1, 1)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    print(f"fp32 G={G} L={L} D={D}: {ms:.2f} ms  {2.0*G*L*L*D/ms/1e9:.2f} TFLOP/s")
