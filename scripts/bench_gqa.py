"""Grouped-query training step (forward + backward) through the kernels' own head mapping against the expanded-K/V route
(repeat_interleave + the ordinary multi-head kernels + autograd's sum over the copies): ms per step and peak memory.
    python scripts/bench_gqa.py [B H Hkv L D]"""
import pathlib
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "flash-attention-ipu_amd")]
import torch  # noqa: E402
from flash_attention_amd._impl import ops  # noqa: E402

B, H, Hkv, L, D = (int(a) for a in sys.argv[1:6]) if len(sys.argv) > 5 else (4, 32, 8, 4096, 128)
dt = torch.bfloat16
q0 = torch.randn(B, H, L, D, device="cuda").to(dt)
k0, v0 = (torch.randn(B, Hkv, L, D, device="cuda").to(dt) for _ in range(2))
g = torch.randn(B, H, L, D, device="cuda").to(dt)
scale = D ** -0.5
ops.GQA_NATIVE_MIN_WORKGROUPS = 0          # time the kernels' own head mapping whatever the grid size


def native():
    q, k, v = (t.detach().requires_grad_(True) for t in (q0, k0, v0))
    ops.attention_gqa(q, k, v, scale, 8, 8, True).backward(g)
    return q.grad, k.grad, v.grad


def expanded():
    q, k, v = (t.detach().requires_grad_(True) for t in (q0, k0, v0))
    rep = H // Hkv
    ke, ve = k.repeat_interleave(rep, dim=1), v.repeat_interleave(rep, dim=1)
    out = ops.attention(q.reshape(-1, L, D), ke.reshape(-1, L, D), ve.reshape(-1, L, D), scale, 8, 8, True)
    out.view(B, H, L, D).backward(g)
    return q.grad, k.grad, v.grad


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, torch.cuda.max_memory_allocated() / 2 ** 20


a, b = native(), expanded()
for name, x, y in zip(("dq", "dk", "dv"), a, b):
    print(f"{name}: native vs expanded max|d| = {(x.float() - y.float()).abs().max().item():.3e} (max |grad| {y.float().abs().max().item():.2f})")
tn, mn = timed(native)
te, me = timed(expanded)
flops = 7.0 * B * H * L * L * D      # forward 2 + backward 5 GEMM units on the causal half
print(f"B={B} H={H} H_kv={Hkv} L={L} D={D} bf16 causal: native {tn:.3f} ms ({flops / tn / 1e9:.0f} TFLOP/s), peak {mn:.0f} MiB; "
      f"expanded K/V {te:.3f} ms ({flops / te / 1e9:.0f} TFLOP/s), peak {me:.0f} MiB")
