"""L2 bunch sweep: the forward's `sched` (groups of one XCD whose workgroups are dispatched together, fa_capi.cpp finish_mfma_plan) against the
library's own choice (K + V of a bunch ~ 8 MiB, a rule measured at L = 4096), on shapes with MANY short groups - where the launch runs
several rounds of slots and the fabric traffic is a multiple of the algorithmic bytes (profiles/r06_L1024: 2.0x at [512,1024,64]).

    python scripts/sched_sweep.py [--shapes G,L,D ...] [--rounds 5] [--dtype bf16|f16] [--scale 1.0]

One process, candidates interleaved over rounds under sustained load, medians; the tile plan stays the library's (only the bunch changes).
"""
import argparse
import pathlib
import statistics
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "scripts"), str(ROOT / "flash-attention-ipu_amd")]
import torch  # noqa: E402
from tune import Shape  # noqa: E402
from flash_attention_amd._impl import _native  # noqa: E402

DEFAULT = ["512,1024,64", "1024,1024,64", "2048,1024,64", "256,2048,64", "512,2048,64", "1024,512,64", "2048,512,64", "4096,256,64",
           "256,1024,128", "512,1024,128", "256,2048,128", "1024,512,128", "128,4096,64", "256,8192,64", "64,4096,128"]


def timed(sh, sched, n):
    """n launches of `sh` with the bunch forced (None: the library's own); the tuning table stays on either way"""
    with _native.debug_options(**({"sched": sched} if sched else {})):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            sh.launch()
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", nargs="*", default=DEFAULT)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--noncausal", action="store_true")
    args = ap.parse_args()
    dt = {"bf16": torch.bfloat16, "f16": torch.float16}[args.dtype]
    ramp = Shape(128, 4096, 64, torch.bfloat16, 1.0, True)
    for _ in range(600):
        ramp.launch()
    torch.cuda.synchronize()
    for spec in args.shapes:
        G, L, D = (int(x) for x in spec.split(","))
        sh = Shape(G, L, D, dt, args.scale, not args.noncausal)
        ng = (G + 7) // 8
        plan = _native.plan(G, L, D, dt, 1, 1, softmax_scale=args.scale, causal=not args.noncausal)
        auto = plan.sched
        cands = [None] + [s for s in (1, 2, 4, 8, 16, 32, 64, 128) if s <= ng and s != auto]
        t1 = timed(sh, None, 3)
        n = max(3, int(6.0 / max(t1, 1e-3)))
        res = {c: [] for c in cands}
        for _ in range(args.rounds):
            for c in cands:
                res[c].append(timed(sh, c, n))
        base = statistics.median(res[None])
        row = [f"auto({auto}) {base * 1e3:8.1f} us {sh.flops / base / 1e9:7.1f} TF"]
        for c in cands[1:]:
            m = statistics.median(res[c])
            wins = sum(1 for a, b in zip(res[c], res[None]) if a < b)
            row.append(f"s{c}: x{base / m:5.3f} ({wins}/{args.rounds})")
        print(f"[{G},{L},{D}] {args.dtype} kernel {plan.kernel} {plan.block_q}x{plan.block_kv}/{plan.waves}w: " + "  ".join(row), flush=True)
        del sh
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
