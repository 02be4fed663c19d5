"""Developer aid (test infrastructure): differential fuzz of the `F.scaled_dot_product_attention` override (flash_attention_amd.auto) against
stock torch SDPA evaluated in fp64 on the same 16-bit-rounded inputs: 3-D / 4-D / 5-D operands, contiguous and head-inner (transposed) views,
head dims inside and outside the MFMA kernels' range (zero-extended, padded, generic kernel, stock hand-off above 256), f16 / bf16 / fp32,
is_causal or not, default and explicit scales, grouped K/V heads with enable_gqa, attn_mask of every broadcast shape (boolean and additive),
with and without a gradient wanted.  Every routing decision of the override - HIP kernel, padding, mask kernels, hand-off to stock - must
give stock's result; where the override raises, it must be the reference's chunk guard (fa.py:28-35 through sdpa.py:52-53: the sequence
length not divisible by the heuristic's chunk counts) and nothing else.

    python tests/tools/fuzz_override.py [cases] [seed]
"""
import pathlib
import random
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "flash-attention-ipu_amd")]
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
import flash_attention_amd.auto  # noqa: F401,E402
from flash_attention_amd._impl.scaled_dot_product_attention import chunk_heuristic  # noqa: E402

dev = torch.device("cuda")
stock = F.scaled_dot_product_attention.__wrapped__
TOL = {torch.float16: 1e-2, torch.bfloat16: 3e-2, torch.float32: 2e-4}


def fuzz(cases, seed, log=print):
    rng = random.Random(seed)
    bad = guarded = 0
    for case in range(cases):
        dtype = rng.choice([torch.bfloat16, torch.float16, torch.bfloat16, torch.float16, torch.float32])
        D = rng.choice([64, 64, 128, 32, 40, 8, 72, 96, 160, 192, 256, 264, 20]) if dtype != torch.float32 else rng.choice([64, 128, 32, 16, 200])
        L = rng.choice([rng.randint(1, 100), 16 * rng.randint(1, 40), 64 * rng.randint(1, 24), 256 * rng.randint(1, 6), rng.randint(100, 900)])
        nd = rng.choice([3, 4, 4, 4, 5])
        B, Hkv = rng.randint(1, 2), rng.randint(1, 2)
        gqa = nd == 4 and rng.random() < 0.25
        H = Hkv * (rng.choice([2, 3]) if gqa else 1)
        causal = rng.random() < 0.6
        scale = rng.choice([None, None, 0.2, 1.0 if D <= 64 else None])
        mask_kind = "none" if causal else rng.choice(["none", "none", "bool_LL", "bool_pad", "bias_HLL", "bias_L", "bool_BHLL"])
        grad = rng.random() < 0.5
        head_inner = nd == 4 and rng.random() < 0.4
        gen = torch.Generator(device=dev).manual_seed(seed * 100003 + case)

        def mk(h):
            shape = {3: (B * h, L, D), 4: (B, h, L, D), 5: (2, B, h, L, D)}[nd]
            if head_inner:
                return (torch.randn(B, L, h, D, device=dev, generator=gen) * 0.6).to(dtype).transpose(1, 2)
            return (torch.randn(*shape, device=dev, generator=gen) * 0.6).to(dtype)
        q, k, v = mk(H), mk(Hkv), mk(Hkv)
        mask = None
        lead = q.shape[:-2]
        if mask_kind == "bool_LL":
            mask = torch.rand(L, L, device=dev, generator=gen) < 0.7
            mask[:, 0] = True
        elif mask_kind == "bool_pad" and nd == 4:
            lens = torch.randint(1, L + 1, (B,), device=dev, generator=gen)
            mask = (torch.arange(L, device=dev)[None, :] < lens[:, None]).view(B, 1, 1, L)
        elif mask_kind == "bias_HLL" and nd == 4:
            mask = (0.7 * torch.randn(1, H, L, L, device=dev, generator=gen)).to(dtype)
        elif mask_kind == "bias_L":
            mask = torch.randn(L, device=dev, generator=gen).to(dtype)
        elif mask_kind == "bool_BHLL":
            mask = torch.rand(*lead, L, L, device=dev, generator=gen) < 0.6
            mask[..., 0] = True
        kw = dict(is_causal=causal)
        if scale is not None:
            kw["scale"] = scale
        if gqa:
            kw["enable_gqa"] = True
        what = (f"case {case}: q{tuple(q.shape)} kv heads {Hkv} {str(dtype).split('.')[-1]} causal={causal} scale={scale} mask={mask_kind if mask is not None else None} "
                f"grad={grad} head_inner={head_inner}")
        try:
            g = torch.randn(q.shape, device=dev, generator=gen).to(dtype)
            qr, kr, vr = (t.double().detach().requires_grad_(grad) for t in (q, k, v))
            mref = mask if mask is None or mask.dtype == torch.bool else mask.double()
            ref = stock(qr, kr, vr, attn_mask=mref, **kw)
            if grad:
                ref.backward(g.double())
            qc, kc, vc = (t.detach().requires_grad_(grad) for t in (q, k, v))
            try:
                out = F.scaled_dot_product_attention(qc, kc, vc, attn_mask=mask, **kw)
            except ValueError as e:
                nq, nk = chunk_heuristic(L, D)
                if mask is None and (L % nq or L % nk) and "divisible" in str(e):
                    guarded += 1
                    continue        # the reference's own guard (fa.py:28-35), as documented
                raise
            ok, why = True, ""
            if out.shape != ref.shape or out.dtype != dtype:
                ok, why = False, f" result {tuple(out.shape)} {out.dtype}"
            else:
                lim = (1e-2 + 1e-2 * ref.detach().abs()) if dtype != torch.float32 else (2e-4 + 1e-3 * ref.detach().abs())
                e_ = (out.detach().double() - ref.detach()).abs()
                nan_ok = torch.equal(torch.isnan(out.detach()), torch.isnan(ref.detach()))
                if not nan_ok or bool((e_[~torch.isnan(e_)] > lim[~torch.isnan(e_)]).any()):
                    ok, why = False, why + f" out max|d|={e_[~torch.isnan(e_)].max().item():.3e} nan_ok={nan_ok}"
                if grad and ok:
                    out.backward(g)
                    for name, got, want in (("dq", qc.grad, qr.grad), ("dk", kc.grad, kr.grad), ("dv", vc.grad, vr.grad)):
                        ee = (got.double() - want).abs()
                        lim = TOL[dtype] * max(1e-2, want.abs().max().item()) + 2e-2 * want.abs()     # (a one-key row's true dQ / dK are exactly 0: an absolute floor)
                        if got.shape != want.shape or not bool(torch.isfinite(got).all()) or bool((ee > lim).any()):
                            ok, why = False, why + f" {name} max|d|={ee.max().item():.3e} (ref max {want.abs().max().item():.2f})"
        except Exception as e:  # noqa: BLE001
            ok, why = False, f" raised {type(e).__name__}: {str(e)[:200]}"
        if not ok:
            bad += 1
            log(f"BAD {what}:{why}")
        if (case + 1) % 25 == 0:
            log(f"... {case + 1} cases, {bad} bad, {guarded} stopped by the reference's chunk guard")
    return bad


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    sd = int(sys.argv[2]) if len(sys.argv) > 2 else 77
    print(f"# tests/tools/fuzz_override.py {n} {sd}", flush=True)
    b = fuzz(n, sd, log=lambda s: print(s, flush=True))
    print(f"{n} cases: {b} bad")
    sys.exit(1 if b else 0)
