"""Developer aid (round 6, verdict item 4): split groups inside the backward's workgroups (fa_bwd_kernels.hip QS) against the unsplit kernels -
gradients vs autograd through the dense fp64 expression, bit-reproducibility, then interleaved in-process timing (debug option bsplit: 1 = unsplit,
-1 = the library's rule, 2 | 3 forced).   usage: python scripts/bwd_split_ab.py [out.txt]"""
import sys, pathlib, statistics
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "flash-attention-ipu_amd")]
import torch
from flash_attention_amd._impl import _native
_native._OPTION_AUTO.setdefault("bsplit", -1)
dev = torch.device("cuda")
lines = []
def say(s):
    print(s, flush=True); lines.append(s)

def grads(q, k, v, g, bsplit, causal=True, scale=None):
    B, H, L, D = q.shape
    scale = D ** -0.5 if scale is None else scale
    out = torch.empty_like(q); lse = torch.empty(B * H, L, device=dev, dtype=torch.float32)
    st = lambda t: (t.stride(0), t.stride(1), t.stride(2))
    _native.fwd_strided(q, k, v, out, lse, B * H, L, D, H, st(q), st(k), st(v), st(out), 1, 1, scale, causal, heads_kv=k.shape[1])
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    with _native.debug_options(bsplit=bsplit):
        _native.bwd(q, k, v, out, g, lse, dq, dk, dv, B * H, L, D, H, scale, causal, heads_kv=k.shape[1])
    torch.cuda.synchronize()
    return out, dq, dk, dv

def check(B, H, Hkv, L, D, dtype, bsplit, causal=True):
    gen = torch.Generator(device=dev).manual_seed(B * 131 + H * 17 + L)
    q = (torch.randn(B, H, L, D, device=dev, generator=gen) * 0.7).to(dtype)
    k, v = ((torch.randn(B, Hkv, L, D, device=dev, generator=gen) * 0.7).to(dtype) for _ in range(2))
    g = torch.randn(B, H, L, D, device=dev, generator=gen).to(dtype)
    o, dq, dk, dv = grads(q, k, v, g, bsplit, causal)
    o2, dq2, dk2, dv2 = grads(q, k, v, g, bsplit, causal)
    _, dq1, dk1, dv1 = grads(q, k, v, g, 1, causal)
    repro = all(torch.equal(a, b) for a, b in ((dq, dq2), (dk, dk2), (dv, dv2)))
    qd, kd, vd = (t.double().requires_grad_(True) for t in (q, k, v))
    ke, ve = (t.repeat_interleave(H // Hkv, dim=1) for t in (kd, vd))
    s = (qd @ ke.transpose(-1, -2)) * D ** -0.5
    if causal:
        s = s.masked_fill(torch.triu(torch.ones(L, L, device=dev, dtype=torch.bool), 1), float("-inf"))
    ref = torch.softmax(s, -1) @ ve
    ref.backward(g.double())
    worst = 0.0
    ok = repro
    for a, b, a1 in ((dq, qd.grad, dq1), (dk, kd.grad, dk1), (dv, vd.grad, dv1)):
        err = (a.double() - b).abs().max().item(); lim = 6e-2 * max(1.0, b.abs().max().item())
        worst = max(worst, err / lim)
        ok &= err <= lim
    dsplit = max((a.float() - b.float()).abs().max().item() for a, b in ((dq, dq1), (dk, dk1), (dv, dv1)))
    say(f"check [{B},{H}/{Hkv},{L},{D}] {str(dtype).split('.')[-1]} causal={causal} bsplit={bsplit}: worst err/limit {worst:.3f}, vs unsplit max|d| {dsplit:.2e}, repro {repro} {'OK' if ok else 'FAIL'}")
    return ok

def t_us(fn, n=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

def ab(G, L, D, variants=(1, -1), dtype=torch.bfloat16):
    q, k, v, g = (torch.randn(1, G, L, D, device=dev).to(dtype) * 0.5 for _ in range(4))
    out = torch.empty_like(q); lse = torch.empty(G, L, device=dev, dtype=torch.float32)
    st = lambda t: (t.stride(0), t.stride(1), t.stride(2))
    _native.fwd_strided(q, k, v, out, lse, G, L, D, G, st(q), st(k), st(v), st(out), 1, 1, D ** -0.5, True)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    def call(): _native.bwd(q, k, v, out, g, lse, dq, dk, dv, G, L, D, G, D ** -0.5, True)
    res = {}
    for _ in range(10): call()
    for rnd in range(7):
        for b in variants:
            with _native.debug_options(bsplit=b):
                call(); call()
                res.setdefault(b, []).append(t_us(call))
    fl = 5.0 * G * L * L * D
    base = statistics.median(res[1])
    for b in variants:
        med = statistics.median(res[b])
        say(f"ab bwd [{G},{L},{D}] {str(dtype).split('.')[-1]}: bsplit={b:2d}  {med:9.1f} us  {fl / med / 1e6:7.1f} TFLOP/s (5GL^2D)  frac {fl / med / 1e6 / 2500:.3f}  x{base / med:.3f}")

allok = True
for args in ((1, 1, 1, 2048, 64, torch.bfloat16, 3), (1, 2, 2, 1500, 64, torch.float16, 3), (1, 4, 2, 1024, 64, torch.bfloat16, 2), (2, 1, 1, 4096, 64, torch.bfloat16, -1),
             (1, 2, 2, 2048, 128, torch.bfloat16, 2), (1, 4, 1, 1024, 128, torch.float16, -1), (1, 3, 3, 1100, 64, torch.bfloat16, 3), (1, 1, 1, 8192, 64, torch.bfloat16, -1)):
    allok &= check(*args)
allok &= check(1, 2, 2, 1024, 64, torch.bfloat16, 3, causal=False)
allok &= check(1, 2, 2, 1024, 128, torch.bfloat16, 2, causal=False)
say("ALL CHECKS " + ("OK" if allok else "FAILED"))
if allok:
    ab(1, 8192, 64, (1, -1, 2, 3))
    ab(2, 16384, 64, (1, -1, 2, 3))
    ab(4, 8192, 128, (1, -1, 2))
    ab(1, 32768, 64, (1, -1))
    ab(8, 2048, 64, (1, -1, 2, 3))
    ab(4, 4096, 64, (1, -1, 2, 3))
    ab(16, 2048, 64, (1, -1, 2, 3))
    ab(8, 4096, 128, (1, -1, 2))
    ab(24, 1024, 64, (1, -1, 3))
    ab(128, 4096, 64, (1, -1))
if len(sys.argv) > 1: pathlib.Path(sys.argv[1]).write_text("\n".join(lines) + "\n")
