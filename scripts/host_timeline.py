"""Developer aid: timeline of ONE eager forward + backward through the override on the host (ns stamps at the stations of the call), median of many.
usage: python scripts/host_timeline.py"""
import sys, pathlib, time, statistics
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "flash-attention-ipu_amd")]
import torch, torch.nn.functional as F
import flash_attention_amd.auto  # noqa
from flash_attention_amd._impl import _native, ops
patched = F.scaled_dot_product_attention
stock = patched.__wrapped__
now = time.perf_counter_ns
T = {}
def wrap(obj, name, tag):
    fn = getattr(obj, name)
    def w(*a, **k):
        T.setdefault(tag + ".in", []).append(now())
        r = fn(*a, **k)
        T.setdefault(tag + ".out", []).append(now())
        return r
    setattr(obj, name, w)
wrap(_native, "fwd_strided", "native_fwd")
wrap(_native, "bwd", "native_bwd")
wrap(ops, "_bwd_impl", "bwd_impl")
wrap(ops, "_fwd_impl", "fwd_impl")
for nm in ("_bwd_gqa_impl", "_fwd_gqa_impl", "_fwd_bhld_train_impl", "_bwd_bhld_impl"):
    if hasattr(ops, nm):
        wrap(ops, nm, nm.strip("_"))

for tag, shape in (("tiny [2,4,64,64] contiguous leafs", (2, 4, 64, 64)), ("GPT-2 layer [8,12,1024,64] contiguous leafs", (8, 12, 1024, 64))):
    B, H, L, D = shape
    q, k, v = (torch.randn(B, H, L, D, device="cuda", dtype=torch.bfloat16, requires_grad=True) for _ in range(3))
    g = torch.randn(B, H, L, D, device="cuda", dtype=torch.bfloat16)
    for f, fname in ((patched, "override"), (stock, "stock")):
        rows = []
        for it in range(300):
            T.clear()
            if it % 20 == 0: torch.cuda.synchronize()
            t0 = now()
            o = f(q, k, v, is_causal=True)
            t1 = now()
            o.backward(g)
            t2 = now()
            q.grad = k.grad = v.grad = None
            t3 = now()
            st = {"fwd call": t1 - t0, "backward() call": t2 - t1, "grad reset": t3 - t2, "total": t3 - t0}
            if "native_fwd.in" in T:
                st["  fwd: entry -> native call"] = T["native_fwd.in"][0] - t0
                st["  fwd: native call (ctypes + launch)"] = T["native_fwd.out"][0] - T["native_fwd.in"][0]
                st["  fwd: native return -> override return"] = t1 - T["native_fwd.out"][0]
            for key in [k_ for k_ in T if k_.endswith(".in") and ("bwd" in k_)]:
                base = key[:-3]
                st[f"  bwd: backward() -> {base} entry"] = T[key][0] - t1
                st[f"  bwd: inside {base}"] = T[base + ".out"][0] - T[key][0]
                st[f"  bwd: {base} exit -> backward() returns"] = t2 - T[base + ".out"][0]
            if it >= 50: rows.append(st)
        torch.cuda.synchronize()
        print(f"== {tag}: {fname} (median us over {len(rows)} calls; the GPU runs behind - these are host stamps)")
        for key in rows[0]:
            print(f"  {key:55s} {statistics.median(r[key] for r in rows) / 1e3:8.1f}")
