"""BASELINE config 4: a GPT-2-small-shaped forward step (12 layers, 12 heads, n_embd 768, T=1024)
through the automatic SDPA override on one MI355X, against the same model with the override undone
(stock torch SDPA math in fp32 on the GPU).  The reference's demo model (nanoGPT submodule) is an empty
directory in the reference tree, so the block below is written fresh to the GPT-2 small shape."""

import math

import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


import pathlib
import sys

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent / "demo"))
from model import CausalSelfAttention, TinyGPT  # noqa: E402  (demo/model.py: the GPT-2-small-shaped model)


def test_config4_attention_call_vs_oracle():
    """BASELINE config 4's attention call as bench.py's `secondary.cfg4_gpt2_fwd` / `cfg4_attn` issue it: batch 8, 12 heads, T = 1024,
    head_dim 64 - q, k, v the head-inner views `[8, 12, 1024, 64]` of one fused `[B, T, 3C]` projection, bf16, through the patched
    F.scaled_dot_product_attention (G = 96: 768 workgroups, below one round of slots -> the placement-balanced block order).
    The whole launch, groups sampled against the oracle (dense fp64 on the same bf16 values, scale D^-0.5), row 0 = V[0] everywhere."""
    import flash_attention_amd.auto  # noqa: F401
    from flash_attention_amd._impl import _native
    from flash_attention_amd._impl.scaled_dot_product_attention import chunk_heuristic
    from oracle import dense_attention_lse, frozen_randn

    B, H, T, D = 8, 12, 1024, 64
    plan = _native.plan(B * H, T, D, torch.bfloat16, *chunk_heuristic(T, D), softmax_scale=D ** -0.5)
    assert plan.kernel == 0 and plan.grid == 768 and plan.balanced == 1, (plan.kernel, plan.grid, plan.balanced)
    proj = frozen_randn(20260101, (B, T, 3 * H * D)).to(torch.bfloat16)
    q, k, v = (t.view(B, T, H, D).transpose(1, 2) for t in proj.cuda().split(H * D, dim=2))
    with torch.no_grad():
        out = F.scaled_dot_product_attention(q, k, v, is_causal=True)
    assert out.shape == (B, H, T, D) and torch.isfinite(out).all()
    assert torch.equal(out[:, :, 0], v[:, :, 0])
    qc, kc, vc = (t.view(B, T, H, D).transpose(1, 2) for t in proj.split(H * D, dim=2))
    for b, h in ((0, 0), (3, 7), (7, 11), (5, 4)):
        ref, _ = dense_attention_lse(qc[b:b + 1, h:h + 1], kc[b:b + 1, h:h + 1], vc[b:b + 1, h:h + 1], scale=D ** -0.5, causal=True)
        err = (out[b:b + 1, h:h + 1].cpu().double() - ref).abs()
        assert bool((err <= 1e-2 + 1e-2 * ref.abs()).all()), f"config 4 attention call, batch {b} head {h}: max|delta|={err.max().item():.3e}"


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_gpt2_small_forward_through_override(dtype):
    import flash_attention_amd.auto  # noqa: F401
    from flash_attention_amd._impl import _native

    patched = F.scaled_dot_product_attention
    assert hasattr(patched, "__wrapped__") and _native.library_path() is not None
    torch.manual_seed(0)
    model = TinyGPT().cuda().eval()
    for p in model.parameters():           # GPT-2 style init keeps activations O(1) through 12 layers
        if p.dim() > 1:
            nn.init.normal_(p, std=0.02)
    idx = torch.randint(0, 50304, (2, 1024), device="cuda")
    calls = []
    orig_fwd = _native.fwd_strided

    def spy(*a, **kw):
        calls.append((a[5], a[6], a[7], a[8]))   # G, L, D, heads
        return orig_fwd(*a, **kw)

    _native.fwd_strided = spy
    try:
        with torch.no_grad(), torch.autocast("cuda", dtype=dtype):
            logits = model(idx)
    finally:
        _native.fwd_strided = orig_fwd
    assert len(calls) == 12 and all(c == (24, 1024, 64, 12) for c in calls), calls   # the HIP kernel served every layer
    # same model, override undone (the reference's own un-patch idiom, t_sdpa.py:123-125), fp32 math
    F.scaled_dot_product_attention = patched.__wrapped__
    try:
        with torch.no_grad():
            ref = model(idx)
    finally:
        F.scaled_dot_product_attention = patched
    err = (logits.float() - ref).abs().max().item()
    scale = ref.abs().max().item()
    assert math.isfinite(err) and err <= 3e-2 * max(1.0, scale), (err, scale)


def test_gpt_training_step_gradients_through_override():
    """SURVEY 8f-3 (demo/train.py:226 counterpart, reduced): one fwd+bwd step of a 4-layer GPT-2-width model
    with the override (HIP forward + HIP backward) vs the same step with the override undone."""
    import flash_attention_amd.auto  # noqa: F401

    patched = F.scaled_dot_product_attention
    torch.manual_seed(1)
    model = TinyGPT(vocab=8192, block=512, n_layer=4).cuda()
    for p in model.parameters():
        if p.dim() > 1:
            nn.init.normal_(p, std=0.02)
    idx = torch.randint(0, 8192, (4, 512), device="cuda")
    tgt = torch.randint(0, 8192, (4, 512), device="cuda")

    def step():
        model.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            logits = model(idx)
        loss = F.cross_entropy(logits.float().view(-1, 8192), tgt.view(-1))
        loss.backward()
        return loss.item(), {n: p.grad.detach().float().clone() for n, p in model.named_parameters()}

    loss_a, grads_a = step()
    F.scaled_dot_product_attention = patched.__wrapped__
    try:
        loss_b, grads_b = step()
    finally:
        F.scaled_dot_product_attention = patched
    assert abs(loss_a - loss_b) <= 2e-3 * abs(loss_b), (loss_a, loss_b)
    for n in grads_a:
        ga, gb = grads_a[n], grads_b[n]
        denom = gb.norm().item() + 1e-12
        assert (ga - gb).norm().item() / denom <= 3e-2, (n, (ga - gb).norm().item() / denom)


def test_torch_compile_traces_through_the_custom_ops():
    """The kernels are torch.library custom ops with fake implementations: torch.compile (fullgraph, no graph
    break) must carry them for inference and for a training step."""
    import flash_attention_amd as fa
    import flash_attention_amd.auto  # noqa: F401

    torch.manual_seed(2)
    attn = CausalSelfAttention(768, 12).cuda().to(torch.bfloat16)
    x = torch.randn(2, 512, 768, device="cuda", dtype=torch.bfloat16)
    ref = attn(x)
    compiled = torch.compile(attn, fullgraph=True, backend="aot_eager")
    torch.testing.assert_close(compiled(x), ref, atol=2e-2, rtol=2e-2)

    xg = x.clone().requires_grad_(True)
    attn(xg).float().pow(2).sum().backward()
    g_ref = xg.grad.clone()
    xg2 = x.clone().requires_grad_(True)
    compiled(xg2).float().pow(2).sum().backward()
    torch.testing.assert_close(xg2.grad, g_ref, atol=5e-2, rtol=5e-2)

    def packed(qkv):
        return fa.flash_attention_qkv_packed(qkv, 4, 4) * 2.0

    qkv = (torch.randn(3, 4, 256, 64, device="cuda") * 0.5).to(torch.bfloat16)
    torch.testing.assert_close(torch.compile(packed, fullgraph=True, backend="aot_eager")(qkv), packed(qkv))


def test_tiny_gpt_trains_through_the_override():
    """demo/train_tiny_gpt.py: 60 AdamW steps on a synthetic task that needs attention; the loss must fall
    well below log(vocab) - i.e. gradients through the HIP forward + backward are usable for training - and track
    the run with the override undone (stock SDPA) step for step within bf16 noise at the start."""
    import pathlib
    import sys

    sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent / "demo"))
    from train_tiny_gpt import train

    import flash_attention_amd.auto  # noqa: F401
    from flash_attention_amd._impl import utils  # noqa: F401

    ours = train(steps=60, override=True)
    assert ours[0] > 3.5 and ours[-1] < 0.6 * ours[0], (ours[0], ours[-1])
    assert all(l == l for l in ours), "NaN loss"
    wrapped = F.scaled_dot_product_attention
    try:   # undo the override for the comparison run
        F.scaled_dot_product_attention = wrapped.__wrapped__
        stock = train(steps=60, override=False)
    finally:
        F.scaled_dot_product_attention = wrapped
    assert abs(ours[0] - stock[0]) < 5e-2, (ours[0], stock[0])       # same init, same data, same first loss
    assert stock[-1] < 0.6 * stock[0] and abs(ours[-1] - stock[-1]) < 0.35 * stock[0], (ours[-1], stock[-1])


@pytest.mark.parametrize("block,batch,steps", [(256, 32, 40), (1024, 8, 16)])
def test_char_level_training_curves_with_and_without_the_override(block, batch, steps):
    """demo/train_char_gpt.py at reduced steps (the reference's table: nb:826-878 block 256 with/without flash
    attention, nb:736 block 1024): same seed, same batches -> the loss curve through the HIP forward + backward must
    track the stock-SDPA curve, both must fall, and the override must not need more memory than the dense path."""
    from train_char_gpt import train

    ours = train("override", block, batch, steps)
    stock = train("stock", block, batch, steps)
    dense = train("math", block, batch, steps)
    lo, ls, ld = ours["losses"], stock["losses"], dense["losses"]
    assert all(x == x for x in lo), "NaN loss"
    assert abs(lo[0] - ls[0]) < 2e-2 and abs(lo[0] - ld[0]) < 2e-2, (lo[0], ls[0], ld[0])      # same init, same first batch
    assert lo[-1] < 0.75 * lo[0] and ls[-1] < 0.75 * ls[0], (lo[0], lo[-1], ls[-1])           # both learn
    tail = max(1, steps // 4)
    mo, ms = sum(lo[-tail:]) / tail, sum(ls[-tail:]) / tail
    assert abs(mo - ms) < 0.08 * ms, (mo, ms)                                                 # and stay together
    assert ours["peak_mib"] <= 1.05 * dense["peak_mib"], (ours["peak_mib"], dense["peak_mib"])
    assert ours["it_per_s"] > 0 and stock["it_per_s"] > 0


def test_char_level_training_with_attention_dropout_through_the_override():
    """nanoGPT's `shakespeare_char` config trains with dropout 0.2 (attention dropout inside scaled_dot_product_attention): the
    reference's override raises for it (sdpa.py:25-28), this one serves it with the dropout kernels (fa_fwd_ex / fa_bwd_ex).
    demo/train_char_gpt.py --dropout 0.2 at reduced steps: the override's run must learn, must have used the dropout kernels, must
    repeat itself under the same seed, and must end near the stock-SDPA run (other random masks: same distribution, not same values)."""
    from flash_attention_amd._impl import ops
    from train_char_gpt import train

    calls = []
    real = ops.attention_extra
    ops.attention_extra = lambda *a, **kw: (calls.append(a[6]), real(*a, **kw))[1]      # a[6] = dropout_p
    try:
        ours = train("override", 256, 32, 40, dropout=0.2)
        again = train("override", 256, 32, 8, dropout=0.2)
    finally:
        ops.attention_extra = real
    stock = train("stock", 256, 32, 40, dropout=0.2)
    lo, ls = ours["losses"], stock["losses"]
    assert len(calls) >= 40 * 6 and all(abs(p - 0.2) < 1e-6 for p in calls), "the dropout kernels must have served every attention call"
    assert all(x == x for x in lo), "NaN loss"
    assert again["losses"] == lo[:8], "same seed -> same dropout masks -> same losses"
    assert lo[-1] < 0.8 * lo[0] and ls[-1] < 0.8 * ls[0], (lo[0], lo[-1], ls[-1])
    tail = 10
    mo, ms = sum(lo[-tail:]) / tail, sum(ls[-tail:]) / tail
    assert abs(mo - ms) < 0.10 * ms, (mo, ms)
