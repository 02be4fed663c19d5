"""Developer aid (test infrastructure: it uses the oracle, so it lives under tests/): run a grid of small cases on
the GPU and print max|delta| vs the oracle per case."""
import sys, pathlib, os
ROOT = pathlib.Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "flash-attention-ipu_amd")]
import torch
import flash_attention_amd as fa
from flash_attention_amd._impl import _native
from oracle import dense_causal_attention_lse, frozen_randn

def run(tag, qkv, nq=1, nk=1):
    ref, _ = dense_causal_attention_lse(qkv.cpu(), compute_dtype=torch.float64)
    out = fa.flash_attention_qkv_packed(qkv.cuda(), nq, nk).cpu().double()
    err = (out - ref).abs()
    bad = (err > 1e-2 + 1e-2 * ref.abs())
    rows_bad = bad.any(-1).nonzero()
    print(f"{tag:40s} max|d|={err.max().item():.3e} bad={int(bad.sum())}/{bad.numel()} first_bad_rows={rows_bad[:6].tolist()}", flush=True)

for kern in os.environ.get("DEBUG_KERNELS", "tile").split(","):      # dma | pipe | pipe16 | tile
    for dt in (torch.bfloat16, torch.float16):
        for waves in ("1", "2", "4", "8"):
            _native.set_debug_option("kernel", kern)      # (knobs are process options now, not per-call environment reads)
            _native.set_debug_option("waves", int(waves))
            for (G, L, D) in [(1, 32, 64), (1, 64, 64), (1, 128, 64), (2, 320, 64), (3, 1000, 64), (1, 128, 128), (2, 320, 128), (2, 777, 128)]:
                x = frozen_randn(7 + L + D, (3, G, L, D)).to(dt)
                run(f"{kern} {dt} w={waves} randn {G}x{L}x{D}", x)
            # structured: zero q,k => uniform softmax => running mean of v
            x = frozen_randn(3, (3, 1, 128, 64)).to(dt); x[0] = 0; x[1] = 0
            run(f"{kern} {dt} w={waves} uniform-softmax", x)
