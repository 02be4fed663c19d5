"""Developer aid ("tile auto-tuner"): re-measure the kernel-selection rules of fa_capi.cpp on this box.

For one shape, times every forward configuration the library can run it with - kernel (dma | pipe), 32-row query
blocks per wave (1 | 2), waves per workgroup (4 | 8 where built) and groups per L2 bunch - interleaved in one process
under sustained load, and prints them best first next to what the library picks by default.  The defaults in
`pick_kernel` / `pick_waves` were derived from such runs; the knobs are the FA_AMD_* environment variables.

    python scripts/tune.py G L D [bf16|f16] [--scaled]      e.g.  python scripts/tune.py 128 4096 64
"""
import ctypes, itertools, os, pathlib, statistics, sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "flash-attention-ipu_amd")]
import torch
from flash_attention_amd._impl import _native

args = [a for a in sys.argv[1:] if not a.startswith("--")]
G, L, D = map(int, args[:3])
dt = torch.float16 if len(args) > 3 and args[3] == "f16" else torch.bfloat16
scaled = "--scaled" in sys.argv
lib = _native.load_native_library()
q, k, v = (torch.randn(G, L, D, device="cuda").to(dt) for _ in range(3))
out = torch.empty_like(q)
st = (ctypes.c_int64 * 3)(L * D, 0, D)
scale = D ** -0.5 if scaled else 1.0
KNOBS = ("FA_AMD_KERNEL", "FA_AMD_QB", "FA_AMD_WAVES", "FA_AMD_SCHED")


def launch():
    rc = lib.fa_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), None, G, L, D, 1, st, st, st, st, 8, 8,
                    _native.FA_DTYPE[dt], ctypes.c_float(scale), 1, torch.cuda.current_stream().cuda_stream)
    assert rc == 0, lib.fa_last_error().decode()


def timed(env, n):
    for kk in KNOBS:
        os.environ.pop(kk, None)
    os.environ.update(env)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        launch()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


ng = (G + 7) // 8
bunches = sorted({b for b in (1, 2, 4, 8, 16) if b <= ng} | {ng})
cands = [{}]
for kern, qb, w in itertools.product(("dma", "pipe"), (1, 2), (4, 8)):
    if qb == 2 and (D != 64 or w != 4):
        continue
    if kern == "pipe" and w == 8:
        continue
    for b in bunches:
        cands.append({"FA_AMD_KERNEL": kern, "FA_AMD_QB": str(qb), "FA_AMD_WAVES": str(w), "FA_AMD_SCHED": str(b)})
flops = 2.0 * G * L * L * D
n = max(4, int(30e-3 / max(1e-6, flops / 900e12)))          # ~30 ms of load per timing
for c in cands:
    timed(c, 2)
res = {i: [] for i in range(len(cands))}
for _ in range(5):
    for i, c in enumerate(cands):
        res[i].append(timed(c, n))
for kk in KNOBS:
    os.environ.pop(kk, None)
rows = sorted((statistics.median(t), i) for i, t in res.items())
print(f"G={G} L={L} D={D} {dt} scale={scale:.4g}: {n} launches per timing, median of 5 interleaved rounds")
for ms, i in rows[:12]:
    name = " ".join(f"{kk[7:]}={vv}" for kk, vv in cands[i].items()) or "library default"
    print(f"  {flops / ms / 1e9:8.1f} TFLOP/s  {ms * 1e3:9.1f} us   {name}")
d_ms = statistics.median(res[0])
print(f"library default: {flops / d_ms / 1e9:.1f} TFLOP/s = {rows[0][0] / d_ms * 100:.1f} % of the best configuration's time")
