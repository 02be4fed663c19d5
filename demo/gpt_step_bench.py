"""End-to-end evidence on a GPT-2-small-shaped model (counterpart of the reference's demo
notebook cells that compare standard vs flash attention: step rate nb:826/851, memory nb:351-434).

One forward+backward step (bf16 autocast, AdamW excluded) with
  (a) the override undone          -> stock torch SDPA (whatever backend torch picks on ROCm),
  (b) the override undone + math   -> the dense softmax(QK^T)V path the reference demo starts from,
  (c) `import flash_attention_amd.auto`  -> the HIP forward + backward of this repo,
reporting ms/step and torch.cuda.max_memory_allocated (the reference used the PopVision liveness trace).

    python demo/gpt_step_bench.py [--block 1024] [--batch 8] [--layers 12]
"""
import argparse
import pathlib
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "flash-attention-ipu_amd"), str(ROOT / "demo")]

import torch
import torch.nn.functional as F
from torch.nn.attention import SDPBackend, sdpa_kernel

from model import TinyGPT  # demo/model.py


def run(model, idx, tgt, steps, ctx):
    torch.cuda.reset_peak_memory_stats()
    times = []
    for i in range(steps + 2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model.zero_grad(set_to_none=True)
        with ctx(), torch.autocast("cuda", dtype=torch.bfloat16):
            logits = model(idx)
        loss = F.cross_entropy(logits.float().view(-1, logits.shape[-1]), tgt.view(-1))
        loss.backward()
        torch.cuda.synchronize()
        if i >= 2:
            times.append(time.perf_counter() - t0)
    times.sort()
    return times[len(times) // 2] * 1e3, torch.cuda.max_memory_allocated() / 2**20, loss.item()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--block", type=int, default=1024)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--layers", type=int, default=12)
    ap.add_argument("--steps", type=int, default=8)
    args = ap.parse_args()
    torch.manual_seed(0)
    vocab = 50304
    model = TinyGPT(vocab=vocab, block=args.block, n_layer=args.layers).cuda()
    for p in model.parameters():
        if p.dim() > 1:
            torch.nn.init.normal_(p, std=0.02)
    idx = torch.randint(0, vocab, (args.batch, args.block), device="cuda")
    tgt = torch.randint(0, vocab, (args.batch, args.block), device="cuda")

    import contextlib
    import flash_attention_amd.auto  # noqa: F401
    patched = F.scaled_dot_product_attention
    rows = []
    rows.append(("flash_attention_amd override (HIP fwd+bwd)", *run(model, idx, tgt, args.steps, contextlib.nullcontext)))
    F.scaled_dot_product_attention = patched.__wrapped__
    try:
        rows.append(("stock torch SDPA (default backend)", *run(model, idx, tgt, args.steps, contextlib.nullcontext)))
        try:
            rows.append(("stock torch SDPA, math backend (dense)", *run(model, idx, tgt, args.steps, lambda: sdpa_kernel(SDPBackend.MATH))))
        except torch.OutOfMemoryError:
            rows.append(("stock torch SDPA, math backend (dense)", float("nan"), float("nan"), float("nan")))
    finally:
        F.scaled_dot_product_attention = patched
    print(f"GPT-2-small shape: layers={args.layers} heads=12 n_embd=768 block={args.block} batch={args.batch}, bf16 autocast, fwd+bwd")
    for name, ms, mem, loss in rows:
        print(f"  {name:45s} {ms:9.2f} ms/step   peak {mem:9.0f} MiB   loss {loss:.4f}")


if __name__ == "__main__":
    main()
