"""Developer aid: q-tile subsets (fa_plan_t.xs) on / off for group counts that are not a multiple of 8, against the fp64 oracle rows and each other.
usage: python scripts/xs_ab.py GxLxD[,...] [--dtype bf16|f16|f32] [--kernel dma|pipe|pipe16|kh] [--causal 0|1]"""
import argparse, pathlib, statistics, sys
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "flash-attention-ipu_amd")]
import torch
from flash_attention_amd._impl import _native
ap = argparse.ArgumentParser()
ap.add_argument("shapes")
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--kernel", default=None)
ap.add_argument("--causal", type=int, default=1)
ap.add_argument("--n", type=int, default=10)
a = ap.parse_args()
dt = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[a.dtype]
for shape in a.shapes.split(","):
    G, L, D = (int(x) for x in shape.split("x"))
    torch.manual_seed(1)
    qkv = (torch.randn(3, G, L, D, device="cuda") * (D ** -0.25)).to(dt)
    out = torch.empty(G, L, D, device="cuda", dtype=dt)
    lse = torch.empty(G, L, device="cuda", dtype=torch.float32)
    st = (L * D, 0, D)
    res = {}
    for sub in (0, -1):
        with _native.debug_options(subsets=sub, kernel=a.kernel):
            p = _native.plan(G, L, D, dt, 1, 1, causal=bool(a.causal))
            out.zero_(); lse.zero_()
            _native.fwd_strided(qkv[0], qkv[1], qkv[2], out, lse, G, L, D, 1, st, st, st, st, 1, 1, 1.0, bool(a.causal))
            torch.cuda.synchronize()
            o, l = out.clone(), lse.clone()
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.n):
                    _native.fwd_strided(qkv[0], qkv[1], qkv[2], out, None, G, L, D, 1, st, st, st, st, 1, 1, 1.0, bool(a.causal))
                e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / a.n * 1e3)
            res[sub] = (o, l, statistics.median(ts), p)
    # fp64 reference on a few rows of every group
    rows = sorted({0, 1, L // 2 - 1, L // 2, L - 1} | {min(L - 1, 127 + 128 * k) for k in range(0, (L + 127) // 128, max(1, (L // 128) // 5))})
    q64, k64, v64 = (t.double() for t in qkv)
    s = torch.einsum("grd,gkd->grk", q64[:, rows], k64)
    if a.causal:
        mask = torch.arange(L, device="cuda")[None, :] > torch.tensor(rows, device="cuda")[:, None]
        s = s.masked_fill(mask[None], float("-inf"))
    ref = torch.softmax(s, -1) @ v64
    ref_lse = torch.logsumexp(s, -1)
    flops = (2.0 if a.causal else 4.0) * G * L * L * D
    for sub in (0, -1):
        o, l, t, p = res[sub]
        err = (o[:, rows].double() - ref).abs().max().item()
        lerr = (l[:, rows].double() - ref_lse).abs().max().item()
        print(f"{shape} {a.dtype} subsets={'auto' if sub < 0 else 'off'}: kernel={p.kernel} waves={p.waves} block_q={p.block_q} grid={p.grid} xs={p.xs} sched={p.sched} | {t:9.2f} us {flops / t / 1e6:7.1f} TFLOP/s frac {flops / t / 2.5e9:.3f} | max|O-ref|={err:.2e} max|lse-ref|={lerr:.2e} nan={int(torch.isnan(o.float()).sum())}", flush=True)
    print(f"   off-vs-auto: equal={torch.equal(res[0][0], res[-1][0])} max|dO|={(res[0][0].float() - res[-1][0].float()).abs().max().item():.2e}  speedup x{res[0][2] / res[-1][2]:.2f}")
