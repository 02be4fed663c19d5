"""Tile auto-tuner: measures every legal forward tile plan on THIS device and writes the table
that `fa_capi.cpp:select_plan` consults (flash-attention-ipu_amd/csrc/fa_tuning_table.h).

The reference picks its chunk counts with a closed-form heuristic (scaled_dot_product_attention.py:46-53) and asks
for "a compiler utility" to do better (demo notebook line 1305).  Here the mapping
    (head_dim, arithmetic mode, mask, L bucket, grid fill)  ->  (kernel, waves per workgroup, query blocks per wave, L2 bunch)
is measured: for each sampled shape every candidate the library can launch is timed in ONE process, interleaved over
rounds under sustained load (cdna guide rule 24), the median decides, and the winner becomes a table row covering
the bucket around the sample.  `num_chunks_q / num_chunks_kv` stay upper bounds on the tile at run time.

    python scripts/tune.py --emit [--quick]      rewrite fa_tuning_table.h (then rebuild the library)
    python scripts/tune.py G L D [bf16|f16] [--scaled] [--noncausal]      rank the candidates of one shape

Candidates are forced through fa_debug_set_option (never the environment); the table is switched off while measuring.
"""
import argparse
import ctypes
import datetime
import itertools
import pathlib
import statistics
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "flash-attention-ipu_amd")]
import torch  # noqa: E402
from flash_attention_amd._impl import _native  # noqa: E402

lib = _native.load_native_library()
TABLE = ROOT / "flash-attention-ipu_amd" / "csrc" / "fa_tuning_table.h"


class Shape:
    def __init__(self, G, L, D, dt, scale, causal):
        self.G, self.L, self.D, self.dt, self.scale, self.causal = G, L, D, dt, scale, causal
        self.q, self.k, self.v = (torch.randn(G, L, D, device="cuda").to(dt) for _ in range(3))
        self.out = torch.empty_like(self.q)
        self.st = (ctypes.c_int64 * 3)(L * D, 0, D)
        self.flops = (2.0 if causal else 4.0) * G * L * L * D

    def launch(self):
        rc = lib.fa_fwd(self.q.data_ptr(), self.k.data_ptr(), self.v.data_ptr(), self.out.data_ptr(), None, self.G, self.L, self.D,
                        1, 1, self.st, self.st, self.st, self.st, 1, 1, _native.FA_DTYPE[self.dt], ctypes.c_float(self.scale),
                        int(self.causal), torch.cuda.current_stream().cuda_stream)
        assert rc == 0, lib.fa_last_error().decode()

    def timed(self, knobs, n):
        with _native.debug_options(**({"table": 0, **knobs} if knobs else {})):   # ({} = the library's own choice, table on)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                self.launch()
            e1.record()
            torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n


def candidates(G, L, D, skip, causal, scheds=True):
    ng = (G + 7) // 8
    auto = max(1, min(ng, (8 << 20) // (2 * L * D * 2)))
    bunches = sorted({b for b in (auto // 2, auto, auto * 2) if 1 <= b <= ng}) if scheds else [auto]
    out = []
    for kern, qb, w in itertools.product(("dma", "pipe"), (1, 2), (2, 4, 8)):
        if kern == "pipe" and not (skip and (w <= 4 or D == 128)):
            continue
        if qb == 2 and (D != 64 or w != 4 or L < 256):
            continue
        if w == 2 and L > 1024:
            continue
        if 32 * w * qb > max(L, 32) * 2:
            continue
        for b in bunches:
            out.append({"kernel": kern, "qb": qb, "waves": w, "sched": b})
    if D == 128 and skip:                 # the pipelined kernel on 16x16x32 MFMAs (4 or 8 waves x 32 rows)
        for w in (4, 8):
            for b in bunches:
                out.append({"kernel": "pipe16", "qb": 1, "waves": w, "sched": b})
    wg128 = (G + 7) // 8 * 8 * ((L + 127) // 128)
    if wg128 <= 2048 and L >= 256:        # under-filled grids: kv split groups inside the dma kernel's workgroup
        for sp in ((2, 4) if D == 64 else (2,)):
            out.append({"kernel": "dma", "qb": 1, "waves": 4, "sched": auto, "split": sp})
    if D == 64 and skip and L >= 256:     # ... at head_dim 64: 4 waves x 32 or 64 rows
        for qb in (1, 2):
            for b in bunches:
                out.append({"kernel": "pipe16", "qb": qb, "waves": 4, "sched": b})
    return out, auto


def measure(shape, cands, rounds=5, ms_per_timing=25.0):
    n = max(3, int(ms_per_timing * 1e-3 / max(1e-6, shape.flops / 900e12)))
    for c in cands:
        shape.timed(c, 2)
    res = {i: [] for i in range(len(cands))}
    for _ in range(rounds):
        for i, c in enumerate(cands):
            res[i].append(shape.timed(c, n))
    return {i: statistics.median(t) for i, t in res.items()}, n


def rank_one(args):
    G, L, D = map(int, args.shape[:3])
    dt = torch.float16 if len(args.shape) > 3 and args.shape[3] == "f16" else torch.bfloat16
    scale = D ** -0.5 if args.scaled else 1.0
    skip = dt == torch.bfloat16 or scale <= 0.25
    shape = Shape(G, L, D, dt, scale, not args.noncausal)
    cands, _ = candidates(G, L, D, skip, shape.causal)
    cands = [{}] + cands
    med, n = measure(shape, cands)
    rows = sorted((ms, i) for i, ms in med.items())
    print(f"G={G} L={L} D={D} {dt} scale={scale:.4g} causal={shape.causal}: {n} launches per timing, median of 5 interleaved rounds")
    for ms, i in rows[:14]:
        name = " ".join(f"{k}={v}" for k, v in cands[i].items()) or "library default (table on)"
        print(f"  {shape.flops / ms / 1e9:8.1f} TFLOP/s  {ms * 1e3:9.1f} us   {name}")
    d_ms = med[0]
    print(f"library default: {shape.flops / d_ms / 1e9:.1f} TFLOP/s = {rows[0][0] / d_ms * 100:.1f} % of the best configuration's time")


def emit(args):
    Ls = [256, 512, 1024, 2048, 4096, 8192, 16384, 32768] if not args.quick else [1024, 4096, 16384]
    fills = [256, 1024, 4096] if not args.quick else [1024, 4096]          # 128-row workgroups the launch amounts to
    modes = [(64, 1, 1), (128, 1, 1), (64, 0, 1), (128, 0, 1), (64, 1, 0), (128, 1, 0)]   # (D, skip, causal)
    rows, log = [], []
    for D, skip, causal in modes:
        dt, scale = (torch.bfloat16, 1.0) if skip else (torch.float16, 1.0)
        for li, L in enumerate(Ls):
            L_lo = 1 if li == 0 else int((Ls[li - 1] * L) ** 0.5) + 1
            L_hi = (1 << 30) if li == len(Ls) - 1 else int((L * Ls[li + 1]) ** 0.5)
            prev = None
            for fi, fill in enumerate(fills):
                G = max(8, (fill * 128 // L + 7) // 8 * 8)
                if G * L * D * 2 * 4 > 6 << 30:
                    continue
                wg = ((G + 7) // 8 * 8) * ((L + 127) // 128)
                if prev is not None and wg == prev:
                    continue                               # the fill level is already covered (long L: G clamps at 8)
                prev = wg
                shape = Shape(G, L, D, dt, scale, bool(causal))
                cands, auto = candidates(G, L, D, skip, causal)
                med, n = measure(shape, cands, rounds=3 if args.quick else 5)
                best = min(med, key=med.get)
                # prefer the automatic L2 bunch unless another one is clearly (> 1 %) faster for the same tile
                same_tile_auto = [i for i, c in enumerate(cands) if all(c.get(k, 1) == cands[best].get(k, 1) for k in ("kernel", "qb", "waves", "split")) and c["sched"] == auto]
                sched = cands[best]["sched"]
                if same_tile_auto and med[same_tile_auto[0]] <= med[best] * 1.01:
                    best, sched = same_tile_auto[0], 0
                c = cands[best]
                rows.append((D, skip, causal, L_lo, L_hi, wg, {"dma": "d", "pipe": "p", "pipe16": "x"}[c["kernel"]], c["waves"], c["qb"], sched, L, G,
                             shape.flops / med[best] / 1e9, c.get("split", 1)))
                log.append(f"D={D} skip={skip} causal={causal} L={L} G={G} wg128={wg}: {c} sched->{sched}  "
                           f"{shape.flops / med[best] / 1e9:.0f} TFLOP/s (runner-up {sorted(med.values())[1] / med[best]:.3f}x time)")
                print(log[-1], flush=True)
                del shape
                torch.cuda.empty_cache()
    # grid-fill buckets: split between consecutive sampled fills of one (mode, L) at their geometric mean
    out_rows = []
    key = lambda r: r[:5]  # noqa: E731
    for k, grp in itertools.groupby(rows, key=key):
        grp = list(grp)
        for i, r in enumerate(grp):
            lo = 0 if i == 0 else int((grp[i - 1][5] * r[5]) ** 0.5) + 1
            hi = (1 << 62) if i == len(grp) - 1 else int((r[5] * grp[i + 1][5]) ** 0.5)
            out_rows.append((r, lo, hi))
    dev = torch.cuda.get_device_name(0)
    stamp = datetime.datetime.now(datetime.timezone.utc).strftime("%Y-%m-%d %H:%M UTC")
    head = TABLE.read_text().split("// generated on:")[0]
    body = [f"// generated on: {stamp}, {dev}, {len(out_rows)} rows, by scripts/tune.py --emit{' --quick' if args.quick else ''}",
            "static const TuneRow kTuneTable[] = {",
            "    //  D  skip causal   L_lo        L_hi        wg_lo                wg_hi          kernel waves qb sched split   measured at (L, G): TFLOP/s"]
    for r, lo, hi in out_rows:
        D, skip, causal, L_lo, L_hi, wg, kern, waves, qb, sched, L, G, tf, split = r
        body.append(f"    {{{D:4d}, {skip}, {causal}, {L_lo:10d}, {L_hi:10d}, {lo:20d}ll, {hi:20d}ll, '{kern}', {waves}, {qb}, {sched:3d}, {split}}},   // L={L} G={G}: {tf:.0f}")
    body += ["    {0, 0, 0, 0, 0, 0, 0, 'd', 0, 0, 0, 0},   // sentinel (D = 0 never matches)", "};", "", "}  // namespace fa", "#endif", ""]
    TABLE.write_text(head + "\n".join(body))
    (ROOT / "gpurun_out").mkdir(exist_ok=True)
    (ROOT / "gpurun_out" / "tune_log.txt").write_text("\n".join(log) + "\n")
    (ROOT / "gpurun_out" / "fa_tuning_table.h").write_text(TABLE.read_text())   # (gpurun merges gpurun_out/ back)
    print(f"wrote {TABLE} ({len(out_rows)} rows)")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("shape", nargs="*")
    ap.add_argument("--emit", action="store_true")
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--scaled", action="store_true")
    ap.add_argument("--noncausal", action="store_true")
    a = ap.parse_args()
    if a.emit:
        emit(a)
    else:
        rank_one(a)
