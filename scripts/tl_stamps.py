"""Developer aid (round 6): where a team-list workgroup (fa_fwd_tl.hip) spends its life, from the FA_STAMPS measurement build
(`make -C flash-attention-ipu_amd variant NAME=stamps DEFS=-DFA_STAMPS=1`): per workgroup type the prologue, every lock-step round and the
fold / epilogue in shader-clock cycles, next to the per-tile time of the plain 64-key-tile grid on the same shape.
    python scripts/tl_stamps.py G L [tlw]"""
import ctypes, pathlib, sys, time
import numpy as np
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "flash-attention-ipu_amd")]
import torch
from flash_attention_amd._impl import _native
G, L = int(sys.argv[1]), int(sys.argv[2])
tlw = int(sys.argv[3]) if len(sys.argv) > 3 else 0
lib = ctypes.CDLL(str(ROOT / "flash-attention-ipu_amd" / "build" / "variants" / "libfa_stamps.so"))
_native._declare(lib)
lib.fa_debug_set_stamp_buffer.restype = ctypes.c_int
lib.fa_debug_set_stamp_buffer.argtypes = [ctypes.c_void_p, ctypes.c_int64]
D = 64
qkv = torch.randn(3, G, L, D, device="cuda").bfloat16()
out = torch.empty(G, L, D, device="cuda", dtype=torch.bfloat16)
lse = torch.zeros(G, L, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
arr = lambda: (ctypes.c_int64 * 3)(L * D, 0, D)
def launch():
    rc = lib.fa_fwd(qkv[0].data_ptr(), qkv[1].data_ptr(), qkv[2].data_ptr(), out.data_ptr(), lse.data_ptr(), G, L, D, 1, 1, arr(), arr(), arr(), arr(), 1, 1, 1,
                    ctypes.c_float(0.125), 1, stream)
    assert rc == 0, lib.fa_last_error().decode()
def run(kernel):
    assert lib.fa_debug_set_option(b"kernel", kernel) == 0 and lib.fa_debug_set_option(b"tlw", tlw) == 0
    plan = _native.FaPlan()
    assert lib.fa_plan(G, L, D, 1, 1, 1, ctypes.c_float(0.125), 1, ctypes.byref(plan)) == 0
    stamps = torch.zeros(plan.grid * plan.waves * 16, dtype=torch.int64, device="cuda")
    assert lib.fa_debug_set_stamp_buffer(stamps.data_ptr(), stamps.numel() * 8) == 0
    t_end = time.time() + 2.0
    while time.time() < t_end:
        for _ in range(50): launch()
        torch.cuda.synchronize()
    stamps.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); launch(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3
    s = stamps.cpu().numpy().astype(np.uint64).reshape(-1, plan.waves, 16)
    assert lib.fa_debug_set_stamp_buffer(None, 0) == 0
    return plan, us, s
plan, us, s = run(ord("t"))
assert plan.kernel == 7, plan.kernel
s = s[s[:, 0, 0] != 0]
print(f"[{G},{L},64] bf16 team lists: grid {plan.grid}, launch {us:.1f} us (events around ONE launch), {len(s)} workgroups stamped")
w0 = s[:, :, :]
rounds = (w0[:, 0, 5] & 0xff).astype(int); wt = ((w0[:, 0, 5] >> 16) & 0xff).astype(int)
print("type rounds  n_wg | prologue | per round (mean over the type's workgroups, wave 0 of team 0) ... | fold+epilogue | whole (cycles)   wall us (memrealtime)")
for t in sorted(set(wt)):
    m = wt == t
    a = w0[m][:, 0, :].astype(np.int64)        # wave 0 (team 0)
    r = rounds[m][0]
    pro = (a[:, 1] - a[:, 0]).mean()
    per = []
    prev = a[:, 1]
    for i in range(min(r, 10)):
        per.append((a[:, 6 + i] - prev).mean()); prev = a[:, 6 + i]
    tail = (a[:, 3] - a[:, 2]).mean()
    whole = (a[:, 3] - a[:, 0]).mean()
    wall = ((w0[m][:, 0, 5] >> 32).astype(np.int64)).mean() / 100.0
    print(f"  {t:2d}   {r:3d}  {m.sum():4d} | {pro:8.0f} | " + " ".join(f"{x:6.0f}" for x in per) + f" | {tail:7.0f} | {whole:8.0f}   {wall:6.2f}")
allr = []
for i in range(len(s)):
    a = s[i, 0].astype(np.int64); r = int(a[5] & 0xff)
    allr.append((a[2] - a[1]) / r)
print(f"mean cycles per lock-step round over all workgroups: {np.mean(allr):.0f} (min {np.min(allr):.0f}, max {np.max(allr):.0f})")
# the same shape on the plain grid: cycles per tile of a workgroup's loop
plan, us, s = run(ord("d"))
s = s[s[:, 0, 0] != 0]
a = s[:, 0, :].astype(np.int64)
tiles = a[:, 5]
per_tile = (a[:, 2] - a[:, 1]) / np.maximum(tiles, 1)
print(f"plain grid (kernel {plan.kernel}, grid {plan.grid}): launch {us:.1f} us; loop cycles per tile by workgroup tile count:")
for n in sorted(set(tiles)):
    m = tiles == n
    print(f"  {n:3d} tiles x {m.sum():4d} workgroups: prologue {(a[m, 1] - a[m, 0]).mean():7.0f}  loop/tile {per_tile[m].mean():6.0f}  epilogue {(a[m, 3] - a[m, 2]).mean():6.0f}  whole {(a[m, 3] - a[m, 0]).mean():8.0f}")
