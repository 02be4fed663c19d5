"""A few optimizer steps of a small GPT on a synthetic next-token task through the automatic SDPA override
(the counterpart, at toy scale, of the reference's nanoGPT demo trainer, train.py - which this
repository does not rebuild; only "does a model train through the HIP forward + backward" is shown here).

The task is learnable only through attention: token t+1 = (token t + token at the start of its 16-token
segment) mod V, so the loss falls well below log(V) only if gradients flow through the causal attention.

    python demo/train_tiny_gpt.py [--steps 60] [--block 256] [--no-override]
"""
import argparse
import pathlib
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "flash-attention-ipu_amd"), str(ROOT / "demo")]

import torch
import torch.nn.functional as F

from model import TinyGPT  # demo/model.py


def make_batch(batch, block, vocab, gen):
    seg = 16
    x = torch.empty(batch, block + 1, dtype=torch.long)
    x[:, 0] = torch.randint(0, vocab, (batch,), generator=gen)
    for t in range(block):
        start = (t // seg) * seg
        fresh = torch.randint(0, vocab, (batch,), generator=gen)
        nxt = (x[:, t] + x[:, start]) % vocab
        x[:, t + 1] = torch.where(torch.tensor((t + 1) % seg == 0), fresh, nxt)
    return x[:, :-1].contiguous(), x[:, 1:].contiguous()


def train(steps=60, block=256, batch=16, vocab=64, override=True, seed=0, log=None):
    if override:
        import flash_attention_amd.auto  # noqa: F401
    torch.manual_seed(seed)
    gen = torch.Generator().manual_seed(seed)
    model = TinyGPT(vocab=vocab, block=block, n_layer=2, n_head=4, n_embd=256).cuda()
    opt = torch.optim.AdamW(model.parameters(), lr=3e-3, weight_decay=0.0)
    losses = []
    for step in range(steps):
        idx, tgt = (t.cuda() for t in make_batch(batch, block, vocab, gen))
        with torch.autocast("cuda", dtype=torch.bfloat16):
            logits = model(idx)
        loss = F.cross_entropy(logits.float().view(-1, vocab), tgt.view(-1))
        opt.zero_grad(set_to_none=True)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        opt.step()
        losses.append(loss.item())
        if log and (step % 10 == 0 or step == steps - 1):
            log(f"step {step:3d}  loss {loss.item():.4f}")
    return losses


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--block", type=int, default=256)
    ap.add_argument("--no-override", action="store_true")
    a = ap.parse_args()
    losses = train(a.steps, a.block, override=not a.no_override, log=print)
    print(f"first {losses[0]:.3f} -> last {losses[-1]:.3f}  (log V = {torch.log(torch.tensor(64.0)).item():.3f})")


if __name__ == "__main__":
    main()
