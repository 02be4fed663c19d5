"""Developer aid: where the host time of an eager forward + backward through the override goes (tiny problem, GPU work negligible),
and whether a GPT-2-layer call is host- or GPU-bound (kernel time from events vs wall time per call).
usage: python scripts/host_breakdown.py [out.md]"""
import sys, pathlib, time
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "flash-attention-ipu_amd")]
import torch, torch.nn.functional as F
import flash_attention_amd.auto  # noqa
from flash_attention_amd._impl import _native, ops
patched = F.scaled_dot_product_attention
stock = patched.__wrapped__
rows = []


def bench(fn, n=1000):
    for _ in range(50): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def gpu_us(fn, n=50):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def row(name, us):
    rows.append(f"| {name} | {us:.1f} |")
    print(f"{name:70s} {us:8.1f} us", flush=True)


class Ident(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v):
        return q.clone()
    @staticmethod
    def backward(ctx, g):
        return g, g, g


for tag, shape, headinner in (("tiny [2,4,64,64]", (2, 4, 64, 64), False), ("GPT-2 layer [8,12,1024,64] head-inner", (8, 12, 1024, 64), True),
                              ("GPT-2 layer [16,12,1024,64] head-inner", (16, 12, 1024, 64), True), ("char-GPT [64,6,256,64] head-inner", (64, 6, 256, 64), True)):
    B, H, L, D = shape
    if headinner:
        proj = torch.randn(B, L, 3 * H * D, device="cuda", dtype=torch.bfloat16, requires_grad=True)
        def qkv():
            return tuple(t.view(B, L, H, D).transpose(1, 2) for t in proj.split(H * D, dim=2))
    else:
        q0 = torch.randn(B, H, L, D, device="cuda", dtype=torch.bfloat16, requires_grad=True)
        def qkv():
            return q0, q0, q0
    g = torch.randn(B, H, L, D, device="cuda", dtype=torch.bfloat16)
    rows.append(f"| **{tag}** | us per call (host-side wall time, back-to-back calls) |")
    print("==", tag)
    def fb(f):
        q, k, v = qkv()
        o = f(q, k, v, is_causal=True)
        o.backward(g)
        if headinner: proj.grad = None
        else: q0.grad = None
    def fwd_only(f):
        q, k, v = qkv()
        return f(q, k, v, is_causal=True)
    def ident():
        q, k, v = qkv()
        Ident.apply(q, k, v).backward(g)
        if headinner: proj.grad = None
        else: q0.grad = None
    row("python identity Function fwd+bwd (engine + views floor)", bench(ident))
    row("stock fwd (grad)", bench(lambda: fwd_only(stock)))
    row("override fwd (grad)", bench(lambda: fwd_only(patched)))
    row("stock fwd+bwd", bench(lambda: fb(stock), 500))
    row("override fwd+bwd", bench(lambda: fb(patched), 500))
    row("stock fwd+bwd, GPU time (events, median)", gpu_us(lambda: fb(stock)))
    row("override fwd+bwd, GPU time (events, median; includes host gaps)", gpu_us(lambda: fb(patched)))
    # the native calls alone
    q, k, v = (t.detach() for t in qkv())
    out = torch.empty(B, H, L, D, device="cuda", dtype=torch.bfloat16)
    lse = torch.empty(B * H, L, device="cuda", dtype=torch.float32)
    st = lambda t: (t.stride(0), t.stride(1), t.stride(2))
    row("_native.fwd_strided alone", bench(lambda: _native.fwd_strided(q, k, v, out, lse, B * H, L, D, H, st(q), st(k), st(v), st(out), 1, 1, 0.125)))
    dq, dk, dv = (torch.empty_like(out) for _ in range(3))
    row("_native.bwd alone (2 launches + dsum alloc)", bench(lambda: _native.bwd(q, k, v, out, g, lse, dq, dk, dv, B * H, L, D, H, 0.125, True)))
    row("  fwd kernel GPU time", gpu_us(lambda: _native.fwd_strided(q, k, v, out, lse, B * H, L, D, H, st(q), st(k), st(v), st(out), 1, 1, 0.125)))
    row("  bwd kernels GPU time", gpu_us(lambda: _native.bwd(q, k, v, out, g, lse, dq, dk, dv, B * H, L, D, H, 0.125, True)))
if len(sys.argv) > 1:
    pathlib.Path(sys.argv[1]).write_text("| what | us |\n|---|---|\n" + "\n".join(rows) + "\n")
