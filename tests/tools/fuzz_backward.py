"""Developer aid (test infrastructure): differential fuzz of the backward.  Random shapes, dtypes, layouts (contiguous / head-inner), grouped-query
head counts, causal / none; the split-group forms (debug option bsplit 2 | 3 and the library's rule) against the unsplit kernels (another summation
order: within rounding, and bit-reproducible from call to call) and, for short sequences, against autograd through the dense fp64 expression; the
eager 4-D training path of the override (ops._BhldAttentionFn) against the same gradients.

    python tests/tools/fuzz_backward.py [cases] [seed]
"""
import pathlib, random, sys
ROOT = pathlib.Path(__file__).resolve().parent.parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "flash-attention-ipu_amd")]
import torch
import torch.nn.functional as F
import flash_attention_amd.auto  # noqa: F401
from flash_attention_amd._impl import _native
from flash_attention_amd._impl.scaled_dot_product_attention import chunk_heuristic

dev = torch.device("cuda")


def native_grads(q, k, v, g, scale, causal, bsplit, bbal=0):
    B, H, L, D = q.shape
    out = torch.empty(B, H, L, D, device=dev, dtype=q.dtype)
    lse = torch.empty(B * H, L, device=dev, dtype=torch.float32)
    st = lambda t: (t.stride(0), t.stride(1), t.stride(2))  # noqa: E731
    _native.fwd_strided(q, k, v, out, lse, B * H, L, D, H, st(q), st(k), st(v), st(out), 1, 1, scale, causal, heads_kv=k.shape[1])
    dq, dk, dv = torch.empty_like(out), torch.empty(k.shape, device=dev, dtype=k.dtype), torch.empty(v.shape, device=dev, dtype=v.dtype)
    with _native.debug_options(bsplit=bsplit, bbal=bbal):
        _native.bwd(q, k, v, out, g, lse, dq, dk, dv, B * H, L, D, H, scale, causal, heads_kv=k.shape[1])
    torch.cuda.synchronize()
    return out, dq, dk, dv


def fuzz(cases, seed, log=print):
    rng = random.Random(seed)
    bad = n_override = 0
    for case in range(cases):
        D = rng.choice([64, 64, 128])
        L = rng.choice([rng.randint(1, 300), rng.randint(300, 1500), rng.randint(1024, 4200), 512 * rng.randint(2, 8) - rng.choice([0, 0, 1, 77])])
        B = rng.randint(1, 2)
        Hkv = rng.randint(1, 3)
        H = Hkv * rng.choice([1, 1, 2, 4])
        dtype = rng.choice([torch.bfloat16, torch.float16])
        causal = rng.random() < 0.8
        head_inner = rng.random() < 0.4
        bsplit = rng.choice([2, 3, -1])
        bbal = rng.choice([1, -1])        # balanced dQ units: forced where the form exists / the library's rule
        gen = torch.Generator(device=dev).manual_seed(seed * 100003 + case)
        mk = lambda h: (torch.randn(B, L, h, D, device=dev, generator=gen).transpose(1, 2) if head_inner else torch.randn(B, h, L, D, device=dev, generator=gen))  # noqa: E731
        q, k, v = (mk(H) * 0.7).to(dtype), (mk(Hkv) * 0.7).to(dtype), (mk(Hkv) * 0.7).to(dtype)
        g = torch.randn(B, H, L, D, device=dev, generator=gen).to(dtype)
        scale = D ** -0.5
        what = f"case {case}: [{B},{H}/{Hkv},{L},{D}] {str(dtype).split('.')[-1]} causal={causal} head_inner={head_inner} bsplit={bsplit} bbal={bbal}"
        try:
            o1, *plain = native_grads(q, k, v, g, scale, causal, 1)
            _, *a = native_grads(q, k, v, g, scale, causal, bsplit, bbal)
            _, *b = native_grads(q, k, v, g, scale, causal, bsplit, bbal)
            ok = all(torch.equal(x, y) for x, y in zip(a, b))
            lims = [6e-2 * max(1.0, x.float().abs().max().item()) for x in plain]
            ok &= all((x.float() - y.float()).abs().max().item() <= 0.5 * lim for x, y, lim in zip(a, plain, lims))
            # the override's eager training path on the same operands (where the reference's chunk heuristic divides L: sdpa.py:52-53 -> fa.py:28-35
            # raise for the other lengths, and so does the override)
            nq_, nk_ = chunk_heuristic(L, D)
            if L % nq_ == 0 and L % nk_ == 0:
                qa, ka, va = (t.detach().clone().requires_grad_(True) if not head_inner else t.detach().requires_grad_(True) for t in (q, k, v))
                oo = F.scaled_dot_product_attention(qa, ka, va, is_causal=causal, enable_gqa=H != Hkv)
                go = torch.autograd.grad(oo, (qa, ka, va), g)
                ok &= (oo.float() - o1.float()).abs().max().item() <= 2e-2
                ok &= all((x.float() - y.float()).abs().max().item() <= 0.5 * lim for x, y, lim in zip(go, plain, lims))
                n_override += 1
            if L <= 1200:
                qd, kd, vd = (t.double().requires_grad_(True) for t in (q, k, v))
                ke, ve = (t.repeat_interleave(H // Hkv, dim=1) for t in (kd, vd))
                s = (qd @ ke.transpose(-1, -2)) * scale
                if causal:
                    s = s.masked_fill(torch.triu(torch.ones(L, L, device=dev, dtype=torch.bool), 1), float("-inf"))
                ref = torch.softmax(s, -1) @ ve
                ref.backward(g.double())
                ok &= all((x.double() - y).abs().max().item() <= 6e-2 * max(1.0, y.abs().max().item()) for x, y in zip(a, (qd.grad, kd.grad, vd.grad)))
            if not ok:
                bad += 1
                log(f"MISMATCH {what}")
        except Exception as e:  # noqa: BLE001
            bad += 1
            log(f"ERROR {what}: {type(e).__name__}: {e}")
        if case % 25 == 24:
            log(f"... {case + 1} cases, {bad} bad")
    log(f"fuzz_backward: {cases} cases ({n_override} of them also through the override's eager training path), seed {seed}: {bad} mismatches")
    return bad


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    sd = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    sys.exit(1 if fuzz(n, sd) else 0)
