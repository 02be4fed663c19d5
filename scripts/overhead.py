import sys, pathlib, time
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "flash-attention-ipu_amd")]
import torch
import flash_attention_amd as fa
from flash_attention_amd._impl import _native
G, L, D = 128, 4096, 64
qkv = torch.randn(3, G, L, D, device="cuda").to(torch.bfloat16)
out = torch.empty(G, L, D, device="cuda", dtype=torch.bfloat16)
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(n): fn()
    e1.record(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return e0.elapsed_time(e1) / n * 1e3, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6
for rnd in range(3):
    print("python api      (gpu us, host-issue us, wall us):", t(lambda: fa.flash_attention_qkv_packed(qkv, 8, 8)))
    print("native same out (gpu us, host-issue us, wall us):", t(lambda: _native.fwd_packed(qkv, out, None, 8, 8)))
    outs = [torch.empty_like(out) for _ in range(4)]
    i = [0]
    def rot():
        i[0] = (i[0] + 1) % 4
        _native.fwd_packed(qkv, outs[i[0]], None, 8, 8)
    print("native rot out  (gpu us, host-issue us, wall us):", t(rot))
