"""Developer aid (test infrastructure): differential fuzz of the mask / bias kernels (fa_fwd_ex / fa_bwd_ex through ops.attention_extra).
Random shapes, head dims 64 .. 256 (native and zero-extended), dtypes, grouped-query head counts, with and without the causal mask underneath,
every mask kind a caller may pass, with and without dropout (the keep decisions are oracle/dropout.py's) - boolean / additive, [L, L], [B, H, L, L], [1, H, L, L], key padding, a per-key bias, strided views with
strides that are and are not multiples of four elements (LDS-DMA tile images / wave-private rings at head_dim 256 / element loads) - against
autograd through the dense fp64 expression of oracle/dropout.py (dense_attention_extra) on the same 16-bit-rounded inputs; out, and all three
gradients; every call twice (bit-reproducible).

    python tests/tools/fuzz_extra.py [cases] [seed]
"""
import pathlib
import random
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "flash-attention-ipu_amd")]
import torch  # noqa: E402
from flash_attention_amd._impl import ops  # noqa: E402
from oracle import dense_attention_extra  # noqa: E402

dev = torch.device("cuda")
TOL = {torch.float16: 1e-2, torch.bfloat16: 3e-2}


def make_mask(rng, gen, kind, B, H, L, dtype):
    if kind == "bool_LL":
        m = torch.rand(L, L, device=dev, generator=gen) < 0.7
        m[:, 0] = True
        return m
    if kind == "bool_BHLL":
        m = torch.rand(B, H, L, L, device=dev, generator=gen) < 0.6
        m[..., 0] = True
        return m
    if kind == "bool_padding":
        lens = torch.randint(1, L + 1, (B,), device=dev, generator=gen)
        return (torch.arange(L, device=dev)[None, :] < lens[:, None]).view(B, 1, 1, L)
    if kind == "bias_BHLL":
        b = 0.8 * torch.randn(B, H, L, L, device=dev, generator=gen)
        b[..., 1::7] = float("-inf")
        b[..., 0] = 0.0
        return b.to(dtype)
    if kind == "bias_1HLL":
        b = 0.8 * torch.randn(1, H, L, L, device=dev, generator=gen)
        b[..., 0] = 0.0
        return b.to(dtype)
    if kind == "bias_key":
        b = torch.randn(H, 1, L, device=dev, generator=gen)
        b[..., 0] = 0.0
        return b.to(dtype)
    if kind == "bias_view":          # a strided view of a larger buffer: row stride L + pad, pointer offset off
        pad, off = rng.choice([(24, 4), (8, 8), (3, 1), (5, 4)])
        big = 0.8 * torch.randn(1, H, L + 2, L + pad, device=dev, generator=gen)
        big[..., off] = 0.0
        return big.to(dtype)[:, :, 1:1 + L, off:off + L]
    raise AssertionError(kind)


def fuzz(cases, seed, log=print):
    rng = random.Random(seed)
    bad = 0
    for case in range(cases):
        D = rng.choice([64, 128, 192, 256, 256, 256, 160, 232, 72])
        L = rng.choice([rng.randint(1, 130), rng.randint(130, 700), rng.randint(700, 1500), 64 * rng.randint(1, 20)])
        B = rng.randint(1, 2)
        Hkv = rng.randint(1, 2)
        H = Hkv * rng.choice([1, 1, 2])
        dtype = rng.choice([torch.bfloat16, torch.float16])
        causal = rng.random() < 0.3
        kind = rng.choice(["bool_LL", "bool_BHLL", "bool_padding", "bias_BHLL", "bias_1HLL", "bias_1HLL", "bias_key", "bias_view", "bias_view", "none"])
        drop = rng.choice([0.0, 0.0, 0.0, 0.1, 0.3, 0.5]) if kind != "none" else rng.choice([0.1, 0.25, 0.5])
        if drop > 0.0:               # (the oracle takes its keep decisions element by element in numpy: short sequences)
            L = min(L, rng.randint(1, 450))
        dseed, doff = rng.getrandbits(62), 4 * rng.randint(0, 1 << 20)
        gen = torch.Generator(device=dev).manual_seed(seed * 100003 + case)
        q, k, v = ((torch.randn(B, h, L, D, device=dev, generator=gen) * 0.6).to(dtype) for h in (H, Hkv, Hkv))
        g = torch.randn(B, H, L, D, device=dev, generator=gen).to(dtype)
        mask = make_mask(rng, gen, kind, B, H, L, dtype) if kind != "none" else None
        scale = D ** -0.5
        what = f"case {case}: [{B},{H}/{Hkv},{L},{D}] {str(dtype).split('.')[-1]} causal={causal} {kind} p={drop} strides={tuple(mask.stride()) if mask is not None else None}"
        try:
            qr, kr, vr = (t.cpu().double().requires_grad_(True) for t in (q, k, v))      # (the oracle runs on the host)
            ref, _ = dense_attention_extra(qr, kr, vr, scale, causal, mask.cpu() if mask is not None else None, drop, dseed, doff)
            ref.backward(g.cpu().double())
            mc = None
            if mask is not None:
                mc = mask.expand(B, H, L, L)
                mc = mc.view(torch.uint8) if mc.dtype == torch.bool else mc
                if mc.stride(-1) != 1:
                    mc = mc.contiguous()
            runs = []
            for _ in range(2):
                qc, kc, vc = (t.clone().requires_grad_(True) for t in (q, k, v))
                out = ops.attention_extra(qc, kc, vc, mc, scale, causal, drop, dseed, doff)
                out.backward(g)
                runs.append((out.detach(), qc.grad, kc.grad, vc.grad))
            ok = all(torch.equal(a, b) for a, b in zip(*runs))
            why = "" if ok else " NOT bit-reproducible"
            out, dq, dk, dv = runs[0]
            err = (out.cpu().double() - ref.detach()).abs()
            if not bool((err <= 1e-2 + 1e-2 * ref.detach().abs()).all()):
                ok, why = False, why + f" out max|d|={err.max().item():.3e}"
            for name, got, want in (("dq", dq, qr.grad), ("dk", dk, kr.grad), ("dv", dv, vr.grad)):
                e = (got.cpu().double() - want).abs()
                floor = TOL[dtype] * max(1e-3, want.abs().max().item())
                if not bool(torch.isfinite(got).all()) or bool((e > floor + 2e-2 * want.abs()).any()):
                    ok, why = False, why + f" {name} max|d|={e.max().item():.3e} (ref max {want.abs().max().item():.2f})"
        except Exception as e:  # noqa: BLE001
            ok, why = False, f" raised {type(e).__name__}: {e}"
        if not ok:
            bad += 1
            log(f"BAD {what}:{why}")
        if (case + 1) % 25 == 0:
            log(f"... {case + 1} cases, {bad} bad")
    return bad


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    sd = int(sys.argv[2]) if len(sys.argv) > 2 else 2024
    print(f"# tests/tools/fuzz_extra.py {n} {sd}", flush=True)
    b = fuzz(n, sd, log=lambda s: print(s, flush=True))
    print(f"{n} cases: {b} bad")
    sys.exit(1 if b else 0)
