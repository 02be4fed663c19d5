"""Character-level GPT training with and without the automatic SDPA override - the counterpart of the reference's
demo (demo/train.py:139-321 trains nanoGPT's `shakespeare_char` config; the notebook then compares standard and
flash attention at block 256 - 6.52 vs 7.16 it/s, same loss curve, nb:826-878 - and at block 1024, where standard
attention runs out of memory on the IPU, nb:736).

Here: the same model family (nanoGPT `shakespeare_char` shape: 6 layers, 6 heads, n_embd 384; `--dropout 0.2` is that config's own
setting - the reference's override raises for attention dropout, sdpa.py:25-28, this one serves it with its dropout kernels; the
default 0 keeps the three modes' loss curves comparable step by step), AdamW, bf16 autocast, on one MI355X, for
    (a) `import flash_attention_amd.auto`  -> HIP forward + backward of this repository,
    (b) the override undone                 -> stock torch SDPA (ROCm's own fused backend),
    (c) the override undone, math backend   -> dense softmax(QK^T)V, what the reference's "standard attention" is,
reporting it/s, torch.cuda.max_memory_allocated (the reference used the PopVision liveness trace,
profile_utils.py:11-60) and the loss curve.  The same seed gives the same batches and the same initial weights, so
the curves must track each other.

There is no network here, so instead of tiny-shakespeare the corpus is ~1 MB of English-like text that every box of
this image has: the sources of Python's standard library (read as bytes, vocab 256); if that is missing, a seeded
synthetic text.

    python demo/train_char_gpt.py [--block 256] [--batch 64] [--steps 200] [--modes override,stock,math] [--dropout 0.2]
"""
import argparse
import contextlib
import glob
import json
import pathlib
import sys
import sysconfig
import time

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "flash-attention-ipu_amd"), str(ROOT / "demo")]

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from torch.nn.attention import SDPBackend, sdpa_kernel  # noqa: E402

from model import TinyGPT, gpt2_init_  # noqa: E402


def load_corpus(max_bytes=1 << 20) -> torch.Tensor:
    """uint8 tensor of text.  Deterministic for a given image: sorted file list, fixed size."""
    files = sorted(glob.glob(str(pathlib.Path(sysconfig.get_paths()["stdlib"]) / "*.py")))
    buf = bytearray()
    for f in files:
        try:
            buf += pathlib.Path(f).read_bytes()
        except OSError:
            continue
        if len(buf) >= max_bytes:
            break
    if len(buf) < 1 << 16:      # no stdlib sources: seeded synthetic "text" with word and sentence structure
        g = torch.Generator().manual_seed(1234)
        words = [bytes(torch.randint(97, 123, (int(n),), generator=g).tolist()) for n in torch.randint(2, 9, (2000,), generator=g)]
        idx = torch.randint(0, 2000, (200000,), generator=g) % torch.randint(1, 2000, (200000,), generator=g).clamp(min=1)
        buf = bytearray(b" ".join(words[int(i)] for i in idx))
    return torch.tensor(list(buf[:max_bytes]), dtype=torch.uint8)


def get_batch(data, block, batch, gen, device):
    ix = torch.randint(0, data.numel() - block - 1, (batch,), generator=gen)
    x = torch.stack([data[i:i + block] for i in ix.tolist()]).long()
    y = torch.stack([data[i + 1:i + 1 + block] for i in ix.tolist()]).long()
    return x.to(device, non_blocking=True), y.to(device, non_blocking=True)


def train(mode="override", block=256, batch=64, steps=200, seed=1337, lr=1e-3, log=None, n_layer=6, n_head=6, n_embd=384, dropout=0.0):
    """Returns {"losses": [...], "it_per_s": ..., "peak_mib": ...}.  mode: override | stock | math."""
    import flash_attention_amd.auto  # noqa: F401  (patches F.scaled_dot_product_attention; undone below for the baselines)

    dev = torch.device("cuda")
    patched = F.scaled_dot_product_attention
    assert hasattr(patched, "__wrapped__")
    data = load_corpus()
    torch.manual_seed(seed)
    gen = torch.Generator().manual_seed(seed)
    model = gpt2_init_(TinyGPT(vocab=256, block=block, n_layer=n_layer, n_head=n_head, n_embd=n_embd, dropout=dropout)).to(dev)
    model.train()
    opt = torch.optim.AdamW(model.parameters(), lr=lr, betas=(0.9, 0.99), weight_decay=0.1)
    ctx = (lambda: sdpa_kernel(SDPBackend.MATH)) if mode == "math" else contextlib.nullcontext
    if mode != "override":
        F.scaled_dot_product_attention = patched.__wrapped__      # the reference's own un-patch idiom (t_sdpa.py:123-125)
    losses = []
    try:
        torch.cuda.reset_peak_memory_stats()
        t0 = None
        for step in range(steps):
            if step == min(5, steps // 2):      # it/s excludes the first steps (allocator warm-up, kernel load)
                torch.cuda.synchronize()
                t0, s0 = time.perf_counter(), step
            x, y = get_batch(data, block, batch, gen, dev)
            with ctx(), torch.autocast("cuda", dtype=torch.bfloat16):
                logits = model(x)
            loss = F.cross_entropy(logits.float().view(-1, 256), y.view(-1))
            opt.zero_grad(set_to_none=True)
            loss.backward()
            torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
            opt.step()
            losses.append(loss.item())
            if log and (step % 25 == 0 or step == steps - 1):
                log(f"[{mode:8s} block {block}] step {step:4d}  loss {losses[-1]:.4f}")
        torch.cuda.synchronize()
        its = (steps - s0) / (time.perf_counter() - t0) if t0 is not None and steps > s0 else float("nan")
    finally:
        F.scaled_dot_product_attention = patched
    return {"mode": mode, "block": block, "batch": batch, "steps": steps, "losses": losses, "it_per_s": its,
            "peak_mib": torch.cuda.max_memory_allocated() / 2 ** 20}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--block", type=int, default=256)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--modes", default="override,stock,math")
    ap.add_argument("--json", default="")
    ap.add_argument("--dropout", type=float, default=0.0, help="attention / residual / embedding dropout (nanoGPT shakespeare_char: 0.2)")
    a = ap.parse_args()
    results = []
    for mode in a.modes.split(","):
        try:
            results.append(train(mode, a.block, a.batch, a.steps, log=print, dropout=a.dropout))
        except torch.OutOfMemoryError:
            results.append({"mode": mode, "block": a.block, "batch": a.batch, "steps": a.steps, "losses": [], "it_per_s": float("nan"),
                            "peak_mib": float("nan"), "oom": True})
            torch.cuda.empty_cache()
    print(f"\nchar-level GPT (6 layers, 6 heads, n_embd 384), block {a.block}, batch {a.batch}, {a.steps} steps, bf16 autocast, AdamW")
    names = {"override": "flash_attention_amd override (HIP fwd+bwd)", "stock": "stock torch SDPA (default backend)",
             "math": "stock torch SDPA, math backend (dense)"}
    for r in results:
        last = f"{r['losses'][-1]:.4f}" if r["losses"] else "OOM"
        first = f"{r['losses'][0]:.4f}" if r["losses"] else "-"
        print(f"  {names[r['mode']]:45s} {r['it_per_s']:7.2f} it/s   peak {r['peak_mib']:9.0f} MiB   loss {first} -> {last}")
    if a.json:
        pathlib.Path(a.json).write_text(json.dumps(results))


if __name__ == "__main__":
    main()
