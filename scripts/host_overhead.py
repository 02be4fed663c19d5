"""Developer aid: host-side cost per call of the override (tiny problem: the GPU work is negligible)."""
import sys, pathlib, time
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "flash-attention-ipu_amd")]
import torch, torch.nn.functional as F
import flash_attention_amd.auto  # noqa
from flash_attention_amd._impl import _native, ops
patched = F.scaled_dot_product_attention
stock = patched.__wrapped__
q = torch.randn(2, 4, 64, 64, device="cuda", dtype=torch.bfloat16)
qg = q.clone().requires_grad_(True)
out = torch.empty_like(q)
st = (q.stride(0), q.stride(1), q.stride(2))
def bench(fn, n=2000):
    for _ in range(50): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
print(f"stock sdpa (no grad)            {bench(lambda: stock(q, q, q, is_causal=True)):7.1f} us/call")
print(f"override  (no grad, fwd_bhld)   {bench(lambda: patched(q, q, q, is_causal=True)):7.1f} us/call")
print(f"  ops.fwd_bhld only             {bench(lambda: ops.fwd_bhld(q, q, q, 0.125, 1, 1, True)):7.1f} us/call")
print(f"  _native.fwd_strided only      {bench(lambda: _native.fwd_strided(q, q, q, out, None, 8, 64, 64, 4, st, st, st, st, 1, 1, 0.125)):7.1f} us/call")
print(f"stock sdpa (grad, fwd only)     {bench(lambda: stock(qg, qg, qg, is_causal=True)):7.1f} us/call")
print(f"override  (grad, fwd only)      {bench(lambda: patched(qg, qg, qg, is_causal=True)):7.1f} us/call")
def fb(fn):
    o = fn(qg, qg, qg, is_causal=True); o.sum().backward(); qg.grad = None
print(f"stock sdpa fwd+bwd              {bench(lambda: fb(stock), 500):7.1f} us/call")
print(f"override  fwd+bwd               {bench(lambda: fb(patched), 500):7.1f} us/call")
# the packed op (the reference's own entry point), tiny problem
import flash_attention_amd as fa
qkv = torch.randn(3, 8, 64, 64, device="cuda", dtype=torch.bfloat16, requires_grad=True)
gp = torch.randn(8, 64, 64, device="cuda", dtype=torch.bfloat16)
def fbp():
    o = fa.flash_attention_qkv_packed(qkv, 1, 1); o.backward(gp); qkv.grad = None
with torch.no_grad():
    print(f"packed op (no grad)             {bench(lambda: fa.flash_attention_qkv_packed(qkv, 1, 1)):7.1f} us/call")
print(f"packed op fwd+bwd               {bench(fbp, 500):7.1f} us/call")
