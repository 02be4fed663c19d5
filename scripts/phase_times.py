"""Where does a forward workgroup spend its life?  Runs the FA_STAMPS measurement build
(`make -C flash-attention-ipu_amd variant NAME=stamps DEFS=-DFA_STAMPS=1`), in which wave-lane 0 of every wave writes, into a
buffer of its own (fa_debug_set_stamp_buffer; never an output tensor), s_memtime stamps (entry, first K/V tile landed,
[pipelined kernel: first step done, steady groups done, ragged steps done], loop end, stores retired), the 100 MHz
s_memrealtime at entry and exit, the hardware id of the wave, the workgroup's step count and its block index - and prints the
phase means, the per-CU residency, the dispatcher's refill gap and (grids below one round of slots) the block -> CU placement.
Durations inside a wave come from s_memtime (shader clock; its origin differs per CU, so it is never compared across
workgroups); the launch-wide timeline comes from s_memrealtime (10 ns ticks, one origin for the chip).

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
ap.add_argument("--balance", type=int, default=None, help="0 | 1: placement-balanced block order off / on (default: the library's rule)")
ap.add_argument("--lib", default=str(ROOT / "flash-attention-ipu_amd" / "build" / "variants" / "libfa_stamps.so"))
a = ap.parse_args()

lib = ctypes.CDLL(a.lib)
_native._declare(lib)
G, L, D = a.G, a.L, a.D
for name, val in (("kernel", a.kernel), ("waves", a.waves), ("qb", a.qb), ("balance", a.balance)):
    if val is not None:
        v = _native._KERNEL_CODE[val] if name == "kernel" else int(val)
        assert lib.fa_debug_set_option(name.encode(), v) == 0
plan = _native.FaPlan()
assert lib.fa_plan(G, L, D, 1, 1, 1, ctypes.c_float(1.0), 1, ctypes.byref(plan)) == 0
kern = {0: "dma", 2: "pipe", 4: "pipe16", 6: "kh"}.get(plan.kernel, str(plan.kernel))
BQ = plan.block_q
print(f"G={G} L={L} D={D}: kernel={kern} waves={plan.waves} qb={plan.qb} block_q={BQ} grid={plan.grid}")
assert kern in ("dma", "pipe")

qkv = torch.randn(3, G, L, D, device="cuda").bfloat16()
out = torch.empty(G, L, D, device="cuda", dtype=torch.bfloat16)
lse = torch.zeros(G, L, device="cuda")
W = plan.waves
NQ = (L + BQ - 1) // BQ
G8 = (G + 7) // 8 * 8
stamps = torch.zeros(G8 * NQ * W * 16, dtype=torch.int64, device="cuda")
lib.fa_debug_set_stamp_buffer.restype = ctypes.c_int
lib.fa_debug_set_stamp_buffer.argtypes = [ctypes.c_void_p, ctypes.c_int64]
assert lib.fa_debug_set_stamp_buffer(stamps.data_ptr(), stamps.numel() * 8) == 0
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
stamps.zero_()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
launch()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
st = stamps.cpu().numpy().view(np.uint64).reshape(G8, NQ, W, 16)[:G]
t0, t1, t2, t3 = (st[..., i].astype(np.int64) for i in range(4))     # shader-clock stamps (per-CU origin)
r0, r1 = st[..., 10].astype(np.int64), st[..., 11].astype(np.int64)  # 100 MHz stamps (chip-wide origin)
live = r1 > 0
assert live[..., 0].all(), "a workgroup wrote no stamps"
rt = (r1 - r0)[live]
tk = (t3 - t0)[live]
ghz = tk.sum() / (rt.sum() * 10.0)
origin = r0[live].min()
wg_start = np.where(live, r0, np.iinfo(np.int64).max).min(axis=2) - origin        # 10 ns ticks
wg_end = np.where(live, r1, 0).max(axis=2) - origin
span = wg_end.max()
print(f"in-kernel clock (s_memtime against s_memrealtime): {ghz:.3f} GHz")
print(f"one launch: {ms * 1e3:.1f} us by events, {span / 100.0:.1f} us first entry -> last exit; "
      f"{2.0 * G * L * L * D / ms / 1e9:.0f} TFLOP/s (stamped build, single launch between events)")
print(f"sum of workgroup lifetimes / (span x 256 CUs) = {(wg_end - wg_start).sum() / (span * 256.0):.2f} resident workgroups per CU")
hw = st[..., 4]
xcc = (hw >> 32).astype(np.int64)
nsteps = st[..., 5].astype(np.int64)
hwid = (hw & 0xffffffff).astype(np.int64)
cu = (hwid >> 8) & 0xf
sh = (hwid >> 12) & 1
se = (hwid >> 13) & 7
simd = (hwid >> 4) & 3
cu_key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
print(f"distinct CUs seen: {len(np.unique(cu_key[..., 0]))}, xcc {sorted(np.unique(xcc[..., 0]))[:8]}, simd ids of wave 0: {np.bincount(simd[..., 0].ravel(), minlength=4)}")


def stat(name, x):
    x = np.asarray(x).ravel()
    print(f"  {name:34s} mean {x.mean():9.0f}  p10 {np.percentile(x, 10):8.0f}  p50 {np.percentile(x, 50):8.0f}  p90 {np.percentile(x, 90):8.0f}")


w_last = W - 1
ok = live[..., w_last]
stat("workgroup lifetime (ticks)", (wg_end - wg_start) * 10.0 * ghz)
stat("prologue (entry -> tile 0 landed)", (t1 - t0)[..., w_last][ok])
if kern == "pipe":
    ta, tb, tc = (st[..., i].astype(np.int64) for i in (6, 7, 8))
    stat("first (generic) step, last wave", (ta - t1)[..., w_last][ok])
    steady = (tb - ta)[..., w_last][ok]
    stat("steady groups, last wave", steady)
    nst = np.maximum(nsteps[..., w_last][ok] - 1 - 2 * plan.qb, 1)
    stat("  ... per step (approx)", steady / nst)
    stat("ragged + diagonal steps, last wave", (tc - tb)[..., w_last][ok])
    cnt = st[..., 14].astype(np.int64)
    if cnt.max() > 0:    # -DFA_STAMPS=2 build: inside the hand-ordered steps (body = issue of the step's MFMAs and vector items; publish = DMA wait + barrier)
        sel = cnt > 0
        stat("hand-ordered step: body (ticks per step)", st[..., 12].astype(np.int64)[sel] / cnt[sel])
        stat("hand-ordered step: publish (wait + barrier)", st[..., 13].astype(np.int64)[sel] / cnt[sel])
    stat("drain", (t2 - tc)[..., w_last][ok])
for w_ in range(W):
    stat(f"loop, wave {w_}", (t2 - t1)[..., w_][live[..., w_]])
if kern == "dma":
    stat("loop per tile (last wave)", ((t2 - t1)[..., w_last] / np.maximum(nsteps[..., w_last], 1))[ok])
stat("epilogue (loop end -> stores done)", (t3 - t2)[..., w_last][ok])

# residency and refill gap per CU, on the chip-wide 10 ns timeline
keys = cu_key[..., 0].ravel()
start, end = wg_start.ravel(), wg_end.ravel()
res, gaps, per_cu_n, per_cu_steps = [], [], [], []
steps_wg = nsteps[..., 0].ravel()
for k in np.unique(keys):
    sel = keys == k
    s_, e_ = np.sort(start[sel]), np.sort(end[sel])
    res.append((end[sel] - start[sel]).sum() / span)
    per_cu_n.append(int(sel.sum()))
    per_cu_steps.append(int(steps_wg[sel].sum()))
    idx = np.searchsorted(s_, e_, side="left")   # refill gap: each end is followed by the next start at or after it on this CU
    okg = idx < len(s_)
    gaps.extend((s_[idx[okg]] - e_[okg]).tolist())
res, gaps = np.array(res), np.array(gaps)
print(f"resident workgroups per CU (time average): mean {res.mean():.2f} min {res.min():.2f} max {res.max():.2f}")
if len(gaps):
    print(f"gap between a workgroup ending and the next one starting on the same CU: p10 {np.percentile(gaps, 10) * 10 * ghz:.0f} p50 "
          f"{np.percentile(gaps, 50) * 10 * ghz:.0f} p90 {np.percentile(gaps, 90) * 10 * ghz:.0f} ticks")
print(f"launch span {span / 100.0:.1f} us; first 512 workgroup starts within {np.sort(start)[:512].max() / 100.0:.2f} us; "
      f"half of the workgroups have ended by {np.percentile(end, 50) / 100.0:.1f} us; last 2 % end after {np.percentile(end, 98) / 100.0:.1f} us")
print(f"placement: workgroups per CU min {min(per_cu_n)} max {max(per_cu_n)}; kv steps per CU (sum over its workgroups) min {min(per_cu_steps)} "
      f"mean {np.mean(per_cu_steps):.1f} max {max(per_cu_steps)}")
if plan.grid <= 2048:    # a grid of about one round of slots: where does block b land?  (blocks b = 8 s + x share XCD x)
    blk = st[..., 9].astype(np.int64)[..., 0].ravel()
    order = np.argsort(blk)
    xs = xcc[..., 0].ravel()[order]
    cus = (cu_key[..., 0].ravel() % (8 * 2 * 16))[order]
    b_ = blk[order]
    x0 = xs[0]
    sel = xs == x0
    print(f"XCD of block 0 = {x0}; blocks of that XCD in slot order -> (se*2+sh)*16+cu: {list(zip((b_[sel] >> 3).tolist(), cus[sel].tolist()))[:100]}")
    print(f"distinct XCDs among blocks b = 0 (mod 8): {np.unique(xcc[..., 0].ravel()[blk % 8 == 0])}")
    per = {}
    for k, n_, e_ in zip(keys.tolist(), steps_wg.tolist(), end.tolist()):
        per.setdefault(k, []).append((n_, round(e_ / 100.0, 1)))
    worst = sorted(per.items(), key=lambda kv: -max(e for _, e in kv[1]))[:6]
    print("CUs that finish last: (kv steps, end us) of their workgroups:", [v for _, v in worst])
