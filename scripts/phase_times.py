"""Where does a forward workgroup spend its life?  Runs the FA_STAMPS measurement build
(`make -C flash-attention-ipu_amd variant NAME=stamps DEFS=-DFA_STAMPS=1`), in which wave-lane 0 of every wave overwrites the
workgroup's LSE rows with s_memtime stamps (entry, first K/V tile landed, [pipelined kernel: first step done, steady groups
done, ragged steps done], loop end, stores retired), the hardware id of the wave and the workgroup's step count - and prints
the phase means, the per-CU residency and the dispatcher's refill gap.

    python scripts/phase_times.py G L D [--kernel dma|pipe] [--waves N] [--qb N]
"""
import argparse
import ctypes
import pathlib
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "flash-attention-ipu_amd")]
import torch  # noqa: E402
from flash_attention_amd._impl import _native  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("G", type=int)
ap.add_argument("L", type=int)
ap.add_argument("D", type=int)
ap.add_argument("--kernel", default=None)
ap.add_argument("--waves", type=int, default=None)
ap.add_argument("--qb", type=int, default=None)
ap.add_argument("--lib", default=str(ROOT / "flash-attention-ipu_amd" / "build" / "variants" / "libfa_stamps.so"))
a = ap.parse_args()

lib = ctypes.CDLL(a.lib)
_native._declare(lib)
G, L, D = a.G, a.L, a.D
for name, val in (("kernel", a.kernel), ("waves", a.waves), ("qb", a.qb)):
    if val is not None:
        v = _native._KERNEL_CODE[val] if name == "kernel" else int(val)
        assert lib.fa_debug_set_option(name.encode(), v) == 0
plan = _native.FaPlan()
assert lib.fa_plan(G, L, D, 1, 1, 1, ctypes.c_float(1.0), 1, ctypes.byref(plan)) == 0
kern = {0: "dma", 2: "pipe", 4: "pipe16"}.get(plan.kernel, str(plan.kernel))
BQ = plan.block_q
print(f"G={G} L={L} D={D}: kernel={kern} waves={plan.waves} qb={plan.qb} block_q={BQ} grid={plan.grid}")
assert kern in ("dma", "pipe") and L % BQ == 0

qkv = torch.randn(3, G, L, D, device="cuda").bfloat16()
out = torch.empty(G, L, D, device="cuda", dtype=torch.bfloat16)
lse = torch.zeros(G, L, device="cuda")
stream = torch.cuda.current_stream().cuda_stream


def launch():
    rc = lib.fa_fwd_qkv_packed(qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), G, L, D, 1, 1, 1, stream)
    assert rc == 0, lib.fa_last_error().decode()


import time  # noqa: E402
t_end = time.time() + 2.5          # the clock the chip holds under this load settles over seconds, not launches (guide, DVFS item 6)
n_warm = 0
while time.time() < t_end:
    for _ in range(50):
        launch()
    torch.cuda.synchronize()
    n_warm += 50
print(f"warm-up: {n_warm} launches over 2.5 s")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
launch()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
W = plan.waves
per = BQ // 2 // W                   # u64 slots per wave
st = lse.cpu().numpy().view(np.uint64).reshape(G, L // BQ, W, per)
t0, t1, t2, t3 = (st[..., i].astype(np.int64) for i in range(4))
rt = (st[..., 11].astype(np.int64) - st[..., 10].astype(np.int64)).ravel()      # 100 MHz constant clock
tk = (st[..., 3].astype(np.int64) - st[..., 0].astype(np.int64)).ravel()
ghz = tk.sum() / (rt.sum() * 10.0)
print(f"s_memtime rate against s_memrealtime (100 MHz): {ghz:.3f} GHz; launch = {ms * 1e3 * ghz * 1e3:.0f} ticks")
rt_all0 = st[..., 10].astype(np.int64)
rt_all1 = st[..., 11].astype(np.int64)
print(f"realtime span of the launch: {(rt_all1.max() - rt_all0.min()) / 100.0:.1f} us; sum of workgroup lifetimes / (span x 256 CUs) = "
      f"{((rt_all1.max(axis=2) - rt_all0.min(axis=2)).sum() / ((rt_all1.max() - rt_all0.min()) * 256.0)):.2f} resident workgroups per CU")
hw = st[..., 4]
xcc = (hw >> 32).astype(np.int64)
# every XCD counts from its own origin: rebase the stamps per XCD
spans = []
for x in np.unique(xcc):
    sel = xcc[..., 0] == x
    b_ = t0[sel].min()
    for t in (t0, t1, t2, t3):
        t[sel] -= b_
    spans.append(t3[sel].max())
base = 0
span = int(np.median(spans))
print(f"per-XCD spans: {spans}")
print(f"one launch: {ms * 1e3:.1f} us by events, {span} stamp ticks -> {span / ms / 1e3:.1f} ticks/us; "
      f"{2.0 * G * L * L * D / ms / 1e9:.0f} TFLOP/s (stamped build)")
nsteps = st[..., 5].astype(np.int64)
hwid = (hw & 0xffffffff).astype(np.int64)
cu = (hwid >> 8) & 0xf
sh = (hwid >> 12) & 1
se = (hwid >> 13) & 7
simd = (hwid >> 4) & 3
cu_key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
print(f"distinct CUs seen: {len(np.unique(cu_key))}, xcc {sorted(np.unique(xcc))[:8]}, simd ids of wave 0: {np.bincount(simd[..., 0].ravel(), minlength=4)}")


def stat(name, x):
    x = x.ravel()
    print(f"  {name:34s} mean {x.mean():9.0f}  p10 {np.percentile(x, 10):8.0f}  p50 {np.percentile(x, 50):8.0f}  p90 {np.percentile(x, 90):8.0f}")


life = t3.max(axis=2) - t0.min(axis=2)          # workgroup lifetime
stat("workgroup lifetime", life)
stat("prologue (entry -> tile 0 landed)", (t1 - t0)[..., W - 1])
if kern == "pipe":
    raw0 = st[..., 0].astype(np.int64)
    ta, tb = st[..., 6].astype(np.int64) - raw0 + t0, st[..., 7].astype(np.int64) - raw0 + t0
    w = W - 1
    stat("first (generic) step, last wave", (ta - t1)[..., w])
    steady = (tb - ta)[..., w]
    stat("steady groups, last wave", steady)
    nst = np.maximum(nsteps[..., w] - 1 - 2 * plan.qb, 1)
    stat("  ... per step (approx)", steady / nst)
    if per >= 16:
        tc = st[..., 8].astype(np.int64) - raw0 + t0
        stat("ragged + diagonal steps, last wave", (tc - tb)[..., w])
        stat("drain", (t2 - tc)[..., w])
    for w_ in range(W):
        stat(f"loop, wave {w_}", (t2 - t1)[..., w_])
else:
    for w_ in range(W):
        stat(f"loop, wave {w_}", (t2 - t1)[..., w_])
    stat("loop per tile (last wave)", (t2 - t1)[..., W - 1] / np.maximum(nsteps[..., W - 1], 1))
stat("epilogue (loop end -> stores done)", (t3 - t2)[..., W - 1])
stat("wave 0 done -> workgroup done", t3.max(axis=2) - t3[..., 0])

# residency and refill gap per CU
keys = cu_key[..., 0].ravel()
start = t0.min(axis=2).ravel() - base
end = t3.max(axis=2).ravel() - base
res, gaps = [], []
for k in np.unique(keys):
    sel = keys == k
    s_, e_ = np.sort(start[sel]), np.sort(end[sel])
    res.append((e_ - s_).sum() if False else (end[sel] - start[sel]).sum() / span)
    # refill gap: each end is followed by the next start at or after it on this CU
    idx = np.searchsorted(s_, e_, side="left")
    ok = idx < len(s_)
    gaps.extend((s_[idx[ok]] - e_[ok]).tolist())
res = np.array(res)
gaps = np.array(gaps)
print(f"resident workgroups per CU (time average): mean {res.mean():.2f} min {res.min():.2f} max {res.max():.2f}")
if len(gaps):
    print(f"gap between a workgroup ending and the next one starting on the same CU: p10 {np.percentile(gaps, 10):.0f} p50 {np.percentile(gaps, 50):.0f} "
          f"p90 {np.percentile(gaps, 90):.0f} ticks")
tail = span - np.percentile(end, 50)
print(f"launch span {span} ticks; first workgroup start spread p90 {np.percentile(start[np.argsort(start)[:512]], 90):.0f}; "
      f"half of the workgroups have ended by {np.percentile(end, 50):.0f}; last 2 % end after {np.percentile(end, 98):.0f}")
