"""GPT-2-shaped decoder written fresh for the demos, the tests and bench.py's config-4 leg.

The reference's demo trains nanoGPT (demo/train.py:139-321) from a git submodule that is an EMPTY directory in the
reference tree, so there is no model code to follow; this is the standard pre-LN GPT-2 block at the GPT-2-small
shape (12 layers, 12 heads, n_embd 768) whose attention goes through `F.scaled_dot_product_attention(q, k, v,
is_causal=True)` on [B, H, T, hs] views of the fused [B, T, 3C] projection - exactly the call the automatic
override (`import flash_attention_amd.auto`) replaces, as in the reference's demo notebook (nb cell 5)."""

import torch
import torch.nn as nn
import torch.nn.functional as F


class CausalSelfAttention(nn.Module):
    def __init__(self, n_embd, n_head, dropout=0.0):
        super().__init__()
        self.c_attn = nn.Linear(n_embd, 3 * n_embd)
        self.c_proj = nn.Linear(n_embd, n_embd)
        self.n_head, self.n_embd, self.dropout = n_head, n_embd, dropout
        self.resid_dropout = nn.Dropout(dropout)

    def forward(self, x):
        B, T, C = x.shape
        q, k, v = self.c_attn(x).split(self.n_embd, dim=2)
        q, k, v = (t.view(B, T, self.n_head, C // self.n_head).transpose(1, 2) for t in (q, k, v))
        # (nanoGPT's call: attention dropout while training - the reference's override raises for it, sdpa.py:25-28; this one serves it)
        y = F.scaled_dot_product_attention(q, k, v, attn_mask=None, dropout_p=self.dropout if self.training else 0.0, is_causal=True)
        return self.resid_dropout(self.c_proj(y.transpose(1, 2).contiguous().view(B, T, C)))


class Block(nn.Module):
    def __init__(self, n_embd, n_head, dropout=0.0):
        super().__init__()
        self.ln_1, self.ln_2 = nn.LayerNorm(n_embd), nn.LayerNorm(n_embd)
        self.attn = CausalSelfAttention(n_embd, n_head, dropout)
        self.mlp = nn.Sequential(nn.Linear(n_embd, 4 * n_embd), nn.GELU(), nn.Linear(4 * n_embd, n_embd), nn.Dropout(dropout))

    def forward(self, x):
        x = x + self.attn(self.ln_1(x))
        return x + self.mlp(self.ln_2(x))


class TinyGPT(nn.Module):
    def __init__(self, vocab=50304, block=1024, n_layer=12, n_head=12, n_embd=768, dropout=0.0):
        super().__init__()
        self.wte, self.wpe = nn.Embedding(vocab, n_embd), nn.Embedding(block, n_embd)
        self.drop = nn.Dropout(dropout)
        self.h = nn.ModuleList(Block(n_embd, n_head, dropout) for _ in range(n_layer))
        self.ln_f = nn.LayerNorm(n_embd)
        self.lm_head = nn.Linear(n_embd, vocab, bias=False)

    def forward(self, idx):
        pos = torch.arange(idx.shape[1], device=idx.device)
        x = self.drop(self.wte(idx) + self.wpe(pos))
        for blk in self.h:
            x = blk(x)
        return self.lm_head(self.ln_f(x))


def gpt2_init_(model: nn.Module, std: float = 0.02) -> nn.Module:
    """GPT-2 style init: keeps activations O(1) through 12 layers."""
    for p in model.parameters():
        if p.dim() > 1:
            nn.init.normal_(p, std=std)
    return model
