"""Developer aid: time forward and backward of the packed op (events, one process)."""
import sys, pathlib, statistics
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "flash-attention-ipu_amd")]
import torch
from flash_attention_amd._impl import _native
CFG = {"cfg2": (128, 4096, 64), "d128": (64, 4096, 128), "short": (512, 1024, 64), "cfg5": (256, 8192, 64)}
DT = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}
for spec in sys.argv[1:] or ["cfg2"]:          # cfg2 | cfg2:f32 | d128:f16 ...
    name, _, dtn = spec.partition(":")
    G, L, D = CFG[name]
    dt = DT[dtn or "bf16"]
    name = spec
    qkv = (torch.randn(3, G, L, D, device="cuda") * 0.5).to(dt)
    dout = torch.randn(G, L, D, device="cuda").to(dt)
    out = torch.empty(G, L, D, device="cuda", dtype=dt)
    lse = torch.empty(G, L, device="cuda", dtype=torch.float32)
    dqkv = torch.empty_like(qkv)
    def fwd(): _native.fwd_packed(qkv, out, lse, 8, 8)
    def bwd(): _native.bwd(qkv[0], qkv[1], qkv[2], out, dout, lse, dqkv[0], dqkv[1], dqkv[2], G, L, D, 1, 1.0)
    def t(fn, n=10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    fwd(); bwd(); t(fwd, 3); t(bwd, 3)
    tf = statistics.median(t(fwd) for _ in range(5)); tb = statistics.median(t(bwd) for _ in range(5))
    fl = 2.0 * G * L * L * D
    print(f"{name}: fwd {tf*1e3:8.1f} us {fl/tf/1e9:7.1f} TF | bwd {tb*1e3:8.1f} us {2.5*fl/tb/1e9:7.1f} TF (5*G*L^2*D) | bwd/fwd {tb/tf:.2f}")
