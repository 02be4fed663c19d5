"""One-off check at sequence lengths far beyond the tests' 65 536: forward, logsumexp and gradients of ONE group at L = 2^18 and 2^20 against fp64
rows / columns computed directly (row i of out and dQ from its own keys; columns j of dK / dV from all rows, with the kernels' logsumexp -
itself checked on the sample rows).  Index arithmetic, buffer extents (K / V of a group: up to 128 MB) and the grids (8 192 row blocks) at sizes
no parity test reaches.      python scripts/long_context_check.py [L ...]
"""
import pathlib
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "flash-attention-ipu_amd")]
import torch  # noqa: E402
from flash_attention_amd._impl import _native  # noqa: E402

dev = torch.device("cuda")


def check(L, D, dtype, log=print):
    """True when out, logsumexp, dQ rows and dK / dV columns of one causal group of L rows agree with the fp64 expressions"""
    if True:   # (one block: the tensors below are released together at its end)
        G, scale = 1, D ** -0.5
        gen = torch.Generator(device=dev).manual_seed(L + D)
        q, k, v, g = ((torch.randn(G, L, D, device=dev, generator=gen) * 0.7).to(dtype) for _ in range(4))
        out, lse = torch.empty_like(q), torch.empty(G, L, device=dev, dtype=torch.float32)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
        st = (L * D, 0, D)
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        _native.fwd_strided(q, k, v, out, lse, G, L, D, 1, st, st, st, st, 1, 1, scale, True)
        e1.record()
        _native.bwd(q, k, v, out, g, lse, dq, dk, dv, G, L, D, 1, scale, True)
        e2.record()
        torch.cuda.synchronize()
        rows = torch.tensor(sorted({0, 1, 63, 64, 12345, L // 3, L // 2 + 1, L - 129, L - 2, L - 1}), device=dev)
        K, V, Q, Gd = k[0].double(), v[0].double(), q[0].double(), g[0].double()
        s = (Q[rows] @ K.T) * scale
        s = s.masked_fill(torch.arange(L, device=dev)[None, :] > rows[:, None], float("-inf"))
        ref_lse = torch.logsumexp(s, -1)
        P = torch.exp(s - ref_lse[:, None])
        ref_o = P @ V
        dP = Gd[rows] @ V.T
        dsum = (Gd[rows] * ref_o).sum(-1, keepdim=True)
        ref_dq = (P * (dP - dsum)) @ K * scale
        e_o = (out[0][rows].double() - ref_o).abs().max().item()
        e_l = (lse[0][rows].double() - ref_lse).abs().max().item()
        e_q = (dq[0][rows].double() - ref_dq).abs().max().item() / max(1e-9, ref_dq.abs().max().item())
        # columns of dK / dV: keys j, over all rows in slabs (P from the kernels' logsumexp, Dsum from the kernels' out)
        keys = torch.tensor(sorted({0, 1, 127, 128, L // 2, L - 130, L - 1}), device=dev)
        acc_v = torch.zeros(len(keys), D, device=dev, dtype=torch.float64)
        acc_k = torch.zeros_like(acc_v)
        for r0 in range(0, L, 1 << 16):
            sl = slice(r0, r0 + (1 << 16))
            sc = (Q[sl] @ K[keys].T) * scale                                            # [rows, keys]
            sc = sc.masked_fill(keys[None, :] > torch.arange(r0, r0 + sc.shape[0], device=dev)[:, None], float("-inf"))
            Pc = torch.exp(sc - lse[0][sl].double()[:, None])
            ds_ = Pc * ((Gd[sl] @ V[keys].T) - (Gd[sl] * out[0][sl].double()).sum(-1, keepdim=True))
            acc_v += Pc.T @ Gd[sl]
            acc_k += ds_.T @ Q[sl] * scale
        e_v = (dv[0][keys].double() - acc_v).abs().max().item() / max(1e-9, acc_v.abs().max().item())
        e_k = (dk[0][keys].double() - acc_k).abs().max().item() / max(1e-9, acc_k.abs().max().item())
        ok = e_o <= 1e-2 and e_l <= 8e-3 and e_q <= 3e-2 and e_v <= 3e-2 and e_k <= 3e-2 and bool(torch.isfinite(out).all()) and all(bool(torch.isfinite(t).all()) for t in (dq, dk, dv))
        fl = 2.0 * G * L * L * D
        log(f"L={L} D={D} {str(dtype).split('.')[-1]}: fwd {e0.elapsed_time(e1):8.1f} ms ({fl / e0.elapsed_time(e1) / 1e9:6.0f} TFLOP/s)  bwd {e1.elapsed_time(e2):8.1f} ms ({2.5 * fl / e1.elapsed_time(e2) / 1e9:6.0f})  "
              f"max|d| out {e_o:.2e} lse {e_l:.2e}; rel dq {e_q:.2e} dk {e_k:.2e} dv {e_v:.2e}  {'OK' if ok else 'BAD'}")
        del q, k, v, g, out, dq, dk, dv, K, V, Q, Gd
        torch.cuda.empty_cache()
        return ok


if __name__ == "__main__":
    bad = 0
    for L_ in [int(a) for a in sys.argv[1:]] or [1 << 18, 1 << 20]:
        for D_, dt_ in ((64, torch.bfloat16), (128, torch.float16)):
            bad += not check(L_, D_, dt_, log=lambda s_: print(s_, flush=True))
    sys.exit(1 if bad else 0)
