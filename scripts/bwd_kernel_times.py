"""Developer aid: run the backward of one shape in the modes of debug option bsplit (for rocprofv3 --kernel-trace --stats)."""
import sys, pathlib
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "flash-attention-ipu_amd")]
import torch
from flash_attention_amd._impl import _native
G, L, D = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
mode = int(sys.argv[4])
dev = torch.device("cuda")
q, k, v, g = (torch.randn(1, G, L, D, device=dev).bfloat16() * 0.5 for _ in range(4))
out = torch.empty_like(q); lse = torch.empty(G, L, device=dev, dtype=torch.float32)
st = lambda t: (t.stride(0), t.stride(1), t.stride(2))
_native.fwd_strided(q, k, v, out, lse, G, L, D, G, st(q), st(k), st(v), st(out), 1, 1, D ** -0.5, True)
dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
with _native.debug_options(bsplit=mode):
    for _ in range(30):
        _native.bwd(q, k, v, out, g, lse, dq, dk, dv, G, L, D, G, D ** -0.5, True)
torch.cuda.synchronize()
