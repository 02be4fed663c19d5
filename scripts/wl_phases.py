"""Where does a persistent work-list workgroup (fa_fwd_wl.hip) spend its life?  Needs the stamps build:
    make -C flash-attention-ipu_amd variant NAME=stamps DEFS=-DFA_STAMPS=1
Every wave sums, over its units, the shader-clock time of four phases - wait (unit start -> first K/V tile published: the previous
unit's stores, the next Q rows and the tile's DMA drain here), loop (the kv tiles), fold (publish / ticket / fold of a cut q tile),
epilogue (rows out) - and records the 100 MHz chip-wide clock at entry and exit.

    python scripts/wl_phases.py G L D [--cap C] [--slots S]
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
ap.add_argument("--cap", type=int, default=0)
ap.add_argument("--slots", type=int, default=0)
ap.add_argument("--lib", default=str(ROOT / "flash-attention-ipu_amd" / "build" / "variants" / "libfa_stamps.so"))
a = ap.parse_args()
lib = ctypes.CDLL(a.lib)
_native._declare(lib)
G, L, D = a.G, a.L, a.D
for name, val in (("wl", 1), ("wl_cap", a.cap), ("wl_slots", a.slots)):
    assert lib.fa_debug_set_option(name.encode(), int(val)) == 0
plan = _native.FaPlan()
assert lib.fa_plan(G, L, D, 1, 1, 1, ctypes.c_float(1.0), 1, ctypes.byref(plan)) == 0
assert plan.wl == 1
print(f"G={G} L={L} D={D}: cap={plan.wl_cap} units={plan.wl_units} grid={plan.wl_grid} workspace={plan.workspace_bytes >> 20} MB")
qkv = torch.randn(3, G, L, D, device="cuda").bfloat16()
out = torch.empty(G, L, D, device="cuda", dtype=torch.bfloat16)
ws = torch.zeros(plan.workspace_bytes + 256, dtype=torch.uint8, device="cuda")
stamps = torch.zeros(plan.wl_grid * 4 * 16, dtype=torch.int64, device="cuda")
lib.fa_debug_set_stamp_buffer.restype = ctypes.c_int
lib.fa_debug_set_stamp_buffer.argtypes = [ctypes.c_void_p, ctypes.c_int64]
assert lib.fa_debug_set_stamp_buffer(stamps.data_ptr(), stamps.numel() * 8) == 0
stream = torch.cuda.current_stream().cuda_stream
st = (ctypes.c_int64 * 3)(L * D, 0, D)
n = G * L * D * 2


def run():
    rc = lib.fa_fwd_ws(qkv.data_ptr(), qkv.data_ptr() + n, qkv.data_ptr() + 2 * n, out.data_ptr(), None, G, L, D, 1, 1, st, st, st, st,
                       1, 1, 1, ctypes.c_float(1.0), 1, ws.data_ptr(), ws.numel(), stream)
    assert rc == 0, lib.fa_last_error()


for _ in range(5):
    run()
torch.cuda.synchronize()
stamps.zero_()
run()
torch.cuda.synchronize()
s = stamps.cpu().numpy().reshape(-1, 16)
s = s[s[:, 6] > 0]
wait, loop, fold, epi, tiles, units, folds = (s[:, i].astype(np.float64) for i in (1, 2, 7, 8, 5, 6, 12))
life = (s[:, 3] - s[:, 0]).astype(np.float64)
rt = (s[:, 11] - s[:, 10]).astype(np.float64) * 10.0    # ns
t0 = s[:, 10].min()
clock = life.sum() / rt.sum()                             # cycles per ns
print(f"waves {len(s)}  shader clock {clock:.3f} GHz  launch span {(s[:, 11].max() - t0) * 0.01:.2f} us")
print(f"per wave: units {units.mean():.2f}  tiles {tiles.mean():.2f}  folds {folds.mean():.2f}")
for name, v in (("wait", wait), ("loop", loop), ("fold", fold), ("epilogue", epi), ("life", life)):
    print(f"  {name:9s} mean {v.mean():9.0f} cycles = {v.mean() / clock / 1e3:6.2f} us   per unit {v.sum() / units.sum():8.0f}   max {v.max():9.0f}")
print(f"  loop per tile {loop.sum() / tiles.sum():.0f} cycles;  fold per fold {fold.sum() / max(folds.sum(), 1):.0f} (includes the publishes of the parts that did not fold)")
end = (s[:, 11] - t0) * 0.01
start = (s[:, 10] - t0) * 0.01
print(f"  wave start us: min {start.min():.2f} max {start.max():.2f};  wave end us: p10 {np.percentile(end, 10):.2f} p50 {np.percentile(end, 50):.2f} p90 {np.percentile(end, 90):.2f} max {end.max():.2f}")
