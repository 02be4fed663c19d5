"""Developer aid: two builds of libflash_attention.so must agree BIT FOR BIT - used after a source clean-up that is not supposed to change
a single instruction of the kernels that stay (round 5: the experimental axes left fa_fwd_pipe.hip / fa_fwd_pipe16.hip).
usage: python scripts/bit_identity.py OLD.so NEW.so [cases]   - random shapes x forced kernels x dtypes, forward (out + logsumexp) and backward."""
import ctypes, random, sys, pathlib
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "flash-attention-ipu_amd")]
import torch
from flash_attention_amd._impl import _native

old, new = (ctypes.CDLL(p) for p in sys.argv[1:3])
cases = int(sys.argv[3]) if len(sys.argv) > 3 else 400
for lib in (old, new):
    lib.fa_fwd.restype = ctypes.c_int
    lib.fa_fwd.argtypes = _native.load_native_library().fa_fwd.argtypes
    lib.fa_bwd.restype = ctypes.c_int
    lib.fa_bwd.argtypes = _native.load_native_library().fa_bwd.argtypes
    lib.fa_debug_set_option.restype = ctypes.c_int
    lib.fa_debug_set_option.argtypes = [ctypes.c_char_p, ctypes.c_int32]
rng = random.Random(5)
KERN = {"auto": 0, "dma": ord("d"), "pipe": ord("p"), "pipe16": ord("x"), "kh": ord("h")}
stream = torch.cuda.current_stream().cuda_stream
bad = 0
for case in range(cases):
    D = rng.choice([64, 64, 128, 96, 192, 256])
    dt = rng.choice([torch.bfloat16, torch.float16])
    G = rng.choice([1, 2, 5, 8, 12, 16, 33])
    L = rng.choice([32, 96, 200, 257, 512, 1000, 1024, 2048, 2304, 4096])
    causal = rng.choice([1, 1, 0])
    scale = rng.choice([1.0, D ** -0.5])
    kern = rng.choice(list(KERN))
    knobs = {"kernel": KERN[kern], "qb": rng.choice([0, 0, 1, 2]), "waves": rng.choice([0, 0, 4, 8, 2]), "split": rng.choice([0, 0, 0, 2, 4]), "sched": rng.choice([0, 1, 4])}
    q, k, v, do = ((torch.randn(G, L, D, device="cuda") * (0.7 if scale == 1.0 else 1.5)).to(dt) for _ in range(4))
    st = (ctypes.c_int64 * 3)(L * D, 0, D)
    res = []
    for lib in (old, new):
        for name, val in knobs.items():
            assert lib.fa_debug_set_option(name.encode(), val) == 0, (name, val)
        out = torch.full((G, L, D), 7.0, device="cuda", dtype=dt)
        lse = torch.full((G, L), 7.0, device="cuda")
        rc = lib.fa_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), lse.data_ptr(), G, L, D, 1, 1, st, st, st, st, 1, 1,
                        _native.FA_DTYPE[dt], ctypes.c_float(scale), causal, stream)
        assert rc == 0, rc
        dq, dk, dv = (torch.full((G, L, D), 7.0, device="cuda", dtype=dt) for _ in range(3))
        dsum = torch.empty(G, L, device="cuda")
        arr = (ctypes.c_int64 * 24)(*([L * D, 0, D] * 8))
        rc = lib.fa_bwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), do.data_ptr(), lse.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr(),
                        dsum.data_ptr(), G, L, D, 1, 1, ctypes.cast(arr, ctypes.POINTER(ctypes.c_int64 * 3)), _native.FA_DTYPE[dt], ctypes.c_float(scale), causal, stream)
        assert rc == 0, rc
        torch.cuda.synchronize()
        res.append((out, lse, dq, dk, dv))
    same = all(torch.equal(a, b) for a, b in zip(*res))
    if not same:
        bad += 1
        print(f"case {case}: DIFFERENT  G={G} L={L} D={D} {dt} causal={causal} scale={scale:.3f} {kern} {knobs}", flush=True)
print(f"{cases} cases, {bad} different")
sys.exit(1 if bad else 0)
