"""Developer aid: the persistent work-list launch (fa_fwd_wl.hip) against the library's static plan, interleaved in ONE process.

usage: python scripts/wl_ab.py [--n 20] [--rounds 5] [--dtype bf16|f16] SHAPE[,SHAPE...] [VARIANT ...]
  SHAPE   = GxLxD, e.g. 96x1024x64
  VARIANT = "static" | "wl[:cap=C][:slots=S]"   (default: static wl)
Every variant's output and logsumexp is compared with the static plan's (max |delta|), run twice for bit-reproducibility, and the
workspace's counters are checked to be zero after the launches.  Timing = N back-to-back launches between two events, rounds interleaved.
"""
import argparse
import ctypes
import pathlib
import statistics
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "flash-attention-ipu_amd")]
import torch  # noqa: E402
from flash_attention_amd._impl import _native  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=20)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--scale", type=float, default=0.0, help="0 = the packed op's unscaled logits; else q is pre-divided like the override (D^-1/2)")
ap.add_argument("--causal", type=int, default=1)
ap.add_argument("shapes")
ap.add_argument("variants", nargs="*")
args = ap.parse_args()
dt = {"bf16": torch.bfloat16, "f16": torch.float16}[args.dtype]
lib = _native.load_native_library()
variants = args.variants or ["static", "wl"]


def knobs(v):
    k = {"wl": 0, "wl_cap": 0, "wl_slots": 0}
    if v != "static":
        k["wl"] = 1
        for kv in v.split(":")[1:]:
            a, b = kv.split("=")
            k["wl_" + a] = int(b)
    return k


def launch(q, k, v, out, lse, G, L, D, scale):
    st = (L * D, 0, D)
    _native.fwd_strided(q, k, v, out, lse, G, L, D, 1, st, st, st, st, 1, 1, scale, bool(args.causal))


for shape in args.shapes.split(","):
    G, L, D = (int(x) for x in shape.split("x"))
    torch.manual_seed(0)
    qkv = torch.randn(3, G, L, D, device="cuda").to(dt)
    scale = 1.0
    if args.scale:
        scale = D ** -0.5
    out = torch.empty(G, L, D, device="cuda", dtype=dt)
    lse = torch.empty(G, L, device="cuda", dtype=torch.float32)
    flops = (2.0 if args.causal else 4.0) * G * L * L * D
    ref = ref_lse = None
    times = {v: [] for v in variants}
    info = {}
    for v in variants:
        with _native.debug_options(**knobs(v)):
            p = _native.plan(G, L, D, dt, 1, 1, softmax_scale=scale, causal=bool(args.causal))
            info[v] = f"wl={p.wl} cap={p.wl_cap} units={p.wl_units} grid={p.wl_grid} ws={p.workspace_bytes >> 20}MB" if p.wl else f"kernel={p.kernel} grid={p.grid} waves={p.waves}"
            out.zero_(); lse.zero_()
            launch(qkv[0], qkv[1], qkv[2], out, lse, G, L, D, scale)
            torch.cuda.synchronize()
            o1, l1 = out.clone(), lse.clone()
            out.zero_(); lse.zero_()
            launch(qkv[0], qkv[1], qkv[2], out, lse, G, L, D, scale)
            torch.cuda.synchronize()
            same = torch.equal(o1, out) and torch.equal(l1, lse)
            if ref is None:
                ref, ref_lse = o1, l1
            d_o = (o1.float() - ref.float()).abs().max().item()
            d_l = (l1 - ref_lse).abs().max().item()
            bad = int(torch.isnan(o1.float()).sum().item())
            info[v] += f" | max|dO|={d_o:.3e} max|dLSE|={d_l:.3e} nan={bad} bitrepro={same}"
            for t in _native._WS.values():
                head = t[:256].view(torch.int32)
                if int(head.abs().sum().item()) != 0:
                    info[v] += " | WORKSPACE HEADS NOT ZERO " + str(head[::32][:9].tolist())
    for _ in range(args.rounds):
        for v in variants:
            with _native.debug_options(**knobs(v)):
                launch(qkv[0], qkv[1], qkv[2], out, lse, G, L, D, scale)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.n):
                    launch(qkv[0], qkv[1], qkv[2], out, None, G, L, D, scale)
                e1.record()
                torch.cuda.synchronize()
                times[v].append(e0.elapsed_time(e1) / args.n * 1e3)
    print(f"== {shape} {args.dtype} causal={args.causal} scale={scale:.4f}")
    base = statistics.median(times[variants[0]])
    for v in variants:
        med = statistics.median(times[v])
        print(f"  {v:24s} {med:8.2f} us  min {min(times[v]):8.2f}  {flops / med / 1e6:7.1f} TFLOP/s  frac {flops / med / 1e6 / 2500:.3f}  x{base / med:.3f} | {info[v]}", flush=True)
