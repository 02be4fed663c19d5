"""Developer aid (test infrastructure): differential fuzz of the plain forward + backward through the C ABI (fa_fwd / fa_bwd) on operands, results
and gradients that are VIEWS: head-outer or head-inner layouts, rows padded by a random number of elements (multiples of 8: the MFMA kernels'
16-byte rows; sometimes not: the generic kernels), batch / head strides padded, pointer offsets, head dims 8 .. 256 (native and zero-extended),
ragged L, grouped K/V heads, causal or not, scaled and unscaled (f16 unscaled: the exact-max arithmetic), fp32 (the fp32 MFMA and generic kernels).  Checked per case:
  * out, logsumexp and all three gradients against autograd through the dense fp64 expression (oracle/dropout.py dense_attention_extra,
    no mask, no dropout) on the same 16-bit-rounded inputs, elementwise |d| <= tol max|ref| + 2e-2 |ref|;
  * every element of the result buffers OUTSIDE the views (row padding, guard rows) still holds the sentinel: nothing writes out of bounds;
  * two calls are bit-identical.

    python tests/tools/fuzz_strided.py [cases] [seed]
"""
import pathlib
import random
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "flash-attention-ipu_amd")]
import torch  # noqa: E402
from flash_attention_amd._impl import _native  # noqa: E402
from oracle import dense_attention_extra  # noqa: E402

dev = torch.device("cuda")
TOL = {torch.float16: 1e-2, torch.bfloat16: 3e-2, torch.float32: 2e-4}
SENT = 12345.0


def view_of(rng, B, H, L, D, dtype, head_inner, aligned, fill):
    """([B, H, L, D] view, its buffer, boolean map of the buffer's elements that belong to the view)"""
    pad = 8 * rng.randint(0, 3) if aligned else rng.choice([1, 2, 3, 5, 12])
    off = 8 * rng.randint(0, 2) if aligned else rng.choice([0, 1, 4])
    gl, gh = rng.randint(0, 2), rng.randint(0, 1)          # guard rows / guard heads
    if head_inner:      # [B, L + gl, H + gh, D + pad] -> transpose
        buf = torch.full((B, L + gl, H + gh, D + pad + off), fill, device=dev, dtype=dtype)
        v = buf[:, :L, :H, off:off + D].transpose(1, 2)
    else:
        buf = torch.full((B, H + gh, L + gl, D + pad + off), fill, device=dev, dtype=dtype)
        v = buf[:, :H, :L, off:off + D]
    inside = torch.zeros(buf.shape, device=dev, dtype=torch.bool)
    (inside[:, :L, :H, off:off + D] if head_inner else inside[:, :H, :L, off:off + D]).fill_(True)
    return v, buf, inside


def fuzz(cases, seed, log=print):
    rng = random.Random(seed)
    bad = 0
    for case in range(cases):
        D = rng.choice([64, 64, 128, 128, 192, 256, 8, 24, 72, 96, 160, 200, 248])
        L = rng.choice([rng.randint(1, 130), rng.randint(130, 600), rng.randint(600, 1400), 64 * rng.randint(1, 16), 128 * rng.randint(1, 8) + 1])
        B = rng.randint(1, 2)
        Hkv = rng.randint(1, 2)
        H = Hkv * rng.choice([1, 1, 2, 3])
        dtype = rng.choice([torch.bfloat16, torch.float16, torch.bfloat16, torch.float16, torch.float32])
        causal = rng.random() < 0.7
        head_inner = rng.random() < 0.5
        aligned = rng.random() < 0.85
        if dtype == torch.float32:       # (fp32: the fp32 MFMA kernels at 64 / 128, the generic kernels elsewhere - one workgroup per query row: short sequences)
            D = rng.choice([64, 128, 64, 128, 40, 2])
            L = min(L, 700)
        scale = rng.choice([D ** -0.5, D ** -0.5, 1.0 if D <= 64 else D ** -0.5])
        mag = 0.6 if scale < 1.0 else 0.25
        gen = torch.Generator(device=dev).manual_seed(seed * 100003 + case)
        # every third case forces a kernel family / workgroup form (the library's constraint pass turns what a shape cannot launch into what it
        # can) and the backward's split forms: the same checks must hold on every path, not only on the rules' choice
        knobs = {}
        if case % 3 == 2 and dtype != torch.float32:
            knobs = dict(rng.choice([dict(kernel="dma", waves=4), dict(kernel="dma", waves=8), dict(kernel="dma", waves=2), dict(kernel="dma", waves=4, qb=2),
                                     dict(kernel="dma", waves=4, split=2), dict(kernel="dma", waves=4, split=4), dict(kernel="dma", waves=4, pair=1),
                                     dict(kernel="pipe", waves=4, qb=2), dict(kernel="pipe", waves=2), dict(kernel="pipe", waves=1), dict(kernel="pipe16", waves=4),
                                     dict(kernel="pipe16", waves=8), dict(kernel="kh"), dict(kernel="kh", split=2), dict(subsets=0), dict(balance=1)]))
            knobs.update(bsplit=rng.choice([-1, 2, 3]), bbal=rng.choice([-1, 1]))
        what = (f"case {case}: [{B},{H}/{Hkv},{L},{D}] {str(dtype).split('.')[-1]} causal={causal} head_inner={head_inner} aligned={aligned} scale={scale:.4f}"
                + (f" knobs={knobs}" if knobs else ""))
        try:
            ops_ = {}
            for name, h in (("q", H), ("k", Hkv), ("v", Hkv), ("g", H)):
                t, buf, _ = view_of(rng, B, h, L, D, dtype, head_inner, aligned, 0.0)
                t.copy_((torch.randn(B, h, L, D, device=dev, generator=gen) * (1.0 if name == "g" else mag)).to(dtype))
                ops_[name] = t
            q, k, v, g = (ops_[n] for n in "qkvg")
            qr, kr, vr = (t.cpu().double().requires_grad_(True) for t in (q, k, v))
            ref, ref_lse = dense_attention_extra(qr, kr, vr, scale, causal, None)
            # (the oracle's causal term is the reference's finite -10000, the kernels' is -inf: identical at these magnitudes)
            ref.backward(g.cpu().double())
            runs = []
            for _ in range(2):
                res = {n: view_of(random.Random(seed * 7 + case * 13 + i), B, h, L, D, dtype, head_inner, aligned, SENT)
                       for i, (n, h) in enumerate((("out", H), ("dq", H), ("dk", Hkv), ("dv", Hkv)))}
                out, dq, dk, dv = (res[n][0] for n in ("out", "dq", "dk", "dv"))
                lse = torch.full((B * H, L), SENT, device=dev, dtype=torch.float32)
                st = lambda t: (t.stride(0), t.stride(1), t.stride(2))  # noqa: E731
                with _native.debug_options(**knobs):
                    _native.fwd_strided(q, k, v, out, lse, B * H, L, D, H, st(q), st(k), st(v), st(out), 1, 1, scale, causal, heads_kv=Hkv)
                    _native.bwd(q, k, v, out, g, lse, dq, dk, dv, B * H, L, D, H, scale, causal, heads_kv=Hkv)
                torch.cuda.synchronize()
                runs.append((res, lse))
            ok, why = True, ""
            (res, lse), (res2, lse2) = runs
            for n in ("out", "dq", "dk", "dv"):
                if not torch.equal(res[n][0], res2[n][0]):
                    ok, why = False, why + f" {n} NOT bit-reproducible"
                _, buf, inside = res[n]
                if not bool((buf[~inside] == SENT).all()):
                    ok, why = False, why + f" {n}: {int((buf[~inside] != SENT).sum())} elements written OUTSIDE the view"
            for n, want in (("out", ref.detach()), ("dq", qr.grad), ("dk", kr.grad), ("dv", vr.grad)):
                got = res[n][0].cpu().double()
                e = (got - want).abs()
                if dtype == torch.float32:
                    lim = 2e-4 * max(1.0, want.abs().max().item()) + 1e-3 * want.abs()
                else:
                    lim = (1e-2 + 1e-2 * want.abs()) if n == "out" else (TOL[dtype] * max(1e-3, want.abs().max().item()) + 2e-2 * want.abs())
                if not bool(torch.isfinite(got).all()) or bool((e > lim).any()):
                    ok, why = False, why + f" {n} max|d|={e.max().item():.3e} (ref max {want.abs().max().item():.2f})"
            el = (lse.view(B, H, L).cpu().double() - ref_lse).abs().max().item()
            if not el <= (1e-4 if dtype == torch.float32 else 8e-3):
                ok, why = False, why + f" lse max|d|={el:.3e}"
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
    sd = int(sys.argv[2]) if len(sys.argv) > 2 else 11
    print(f"# tests/tools/fuzz_strided.py {n} {sd}", flush=True)
    b = fuzz(n, sd, log=lambda s: print(s, flush=True))
    print(f"{n} cases: {b} bad")
    sys.exit(1 if b else 0)
