"""Forward / backward time of the mask / bias / dropout kernels (fa_fwd_ex / fa_bwd_ex) against the plain kernels on one shape.
    python scripts/bench_extra.py [B H L D] [--only SUBSTRING] [--ours]
--only keeps the cases whose name contains SUBSTRING; --ours leaves stock SDPA out (for a kernel trace of this repo's launches alone)."""
import pathlib
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "flash-attention-ipu_amd")]
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
import flash_attention_amd.auto  # noqa: F401,E402
from flash_attention_amd._impl import ops  # noqa: E402

ARGS = sys.argv[1:]
ONLY = ARGS[ARGS.index("--only") + 1] if "--only" in ARGS else None
OURS = "--ours" in ARGS
NUMS = [a for a in ARGS if a.isdigit()]
B, H, L, D = (int(x) for x in NUMS[:4]) if len(NUMS) >= 4 else (8, 16, 4096, 64)
dt = torch.bfloat16
q, k, v = (torch.randn(B, H, L, D, device="cuda").to(dt) for _ in range(3))
g = torch.randn(B, H, L, D, device="cuda").to(dt)
stock = F.scaled_dot_product_attention.__wrapped__
pad = (torch.arange(L, device="cuda")[None, :] < torch.randint(L // 2, L + 1, (B,), device="cuda")[:, None]).view(B, 1, 1, L)
bias = torch.randn(1, H, L, L, device="cuda").to(dt)
full = torch.rand(B, H, L, L, device="cuda") < 0.9 if B * H * L * L < 3e9 else None


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n)
    return sorted(ts)[1]


def fwd_bwd(fn):
    qq, kk, vv = (t.detach().requires_grad_(True) for t in (q, k, v))

    def run():
        fn(qq, kk, vv).backward(g)
    return run


cases = [("plain causal", dict(is_causal=True)), ("plain unmasked", dict()),
         ("key-padding mask [B,1,1,L] (bool)", dict(attn_mask=pad)), ("bias [1,H,L,L]", dict(attn_mask=bias)),
         ("dropout 0.1, causal", dict(dropout_p=0.1, is_causal=True)), ("key-padding mask + dropout 0.1", dict(attn_mask=pad, dropout_p=0.1))]
if full is not None:
    cases.insert(4, ("mask [B,H,L,L] (bool)", dict(attn_mask=full)))
print(f"B={B} H={H} L={L} D={D} bf16; ms per call, this repo's override | stock torch SDPA")
flops_c, flops_n = 2.0 * B * H * L * L * D, 4.0 * B * H * L * L * D
for name, kw in cases:
    if ONLY is not None and ONLY not in name:
        continue
    fl = flops_c if kw.get("is_causal") else flops_n
    with torch.no_grad():
        a = timeit(lambda: F.scaled_dot_product_attention(q, k, v, **kw))
        try:
            b = float("nan") if OURS else timeit(lambda: stock(q, k, v, **kw))
        except Exception as e:  # noqa: BLE001
            b = float("nan")
    a2 = timeit(fwd_bwd(lambda x, y, z: F.scaled_dot_product_attention(x, y, z, **kw)), 5)
    try:
        b2 = float("nan") if OURS else timeit(fwd_bwd(lambda x, y, z: stock(x, y, z, **kw)), 5)
    except Exception as e:  # noqa: BLE001
        b2 = float("nan")
    print(f"  {name:38s} fwd {a:7.3f} ({fl / a / 1e9:6.0f} TFLOP/s) | {b:7.3f}    fwd+bwd {a2:7.3f} | {b2:7.3f}")
