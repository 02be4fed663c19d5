import sys, pathlib, os
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "flash-attention-ipu_amd")]
import torch
torch.set_printoptions(linewidth=250, precision=3, sci_mode=False)
import flash_attention_amd as fa
from oracle import dense_causal_attention_lse
os.environ["FA_AMD_WAVES"] = "1"
L, D = 64, 64
def run(tag, qkv):
    ref, _ = dense_causal_attention_lse(qkv.float().cpu(), compute_dtype=torch.float64)
    out = fa.flash_attention_qkv_packed(qkv.cuda(), 1, 1).cpu().double()
    err = (out - ref).abs()
    print(f"== {tag}: max|d|={err.max().item():.3e}")
    return out, ref
# 1) v[key][d] = d  (q=k=0): every output row must equal [0..D-1]
x = torch.zeros(3, 1, L, D); x[2, 0] = torch.arange(D).float()[None, :].expand(L, D)
out, ref = run("v=d", x.bfloat16())
print("row0", out[0, 0].int().tolist()); print("row5", out[0, 5].int().tolist()); print("row40", out[0, 40].int().tolist())
# 2) v[key][d] = key: output row r = mean(0..r) for every d
x = torch.zeros(3, 1, L, D); x[2, 0] = torch.arange(L).float()[:, None].expand(L, D)
out, ref = run("v=key", x.bfloat16())
print("col0 out", out[0, :, 0].tolist()); print("col0 ref", ref[0, :, 0].tolist())
print("row 33 across d", out[0, 33].tolist())
# 3) one-hot v: v[key=k0][:] = 1 -> out[r][:] = 1/(r+1) for r>=k0
for k0 in (0, 1, 4, 8, 17, 35):
    x = torch.zeros(3, 1, L, D); x[2, 0, k0] = 1.0
    out, ref = run(f"onehot key {k0}", x.bfloat16())
    got = (out[0, :, 0] * (torch.arange(L) + 1)).round().int().tolist()
    print("   rows where key seen (col0):", got)
# 4) S test: identity-ish q/k: q[r] = e_(r%D) * 8, k[c] = e_(c%D) * 8 -> s[r][c] = 64 if r%D==c%D
x = torch.zeros(3, 1, L, D)
for r in range(L):
    x[0, 0, r, r % D] = 4.0; x[1, 0, r, r % D] = 4.0
x[2, 0] = torch.arange(L).float()[:, None].expand(L, D)
out, ref = run("diag qk, v=key", x.bfloat16())
print("col0 out", out[0, :, 0].tolist()); print("col0 ref", [round(v, 2) for v in ref[0, :, 0].tolist()])
