"""Developer aid: time forward and backward of the packed op (events, one process); with extra library names
(build/variants/libfa_<name>.so) the backward of each library is timed interleaved with the shipped one (A/B).

    python scripts/bench_bwd.py cfg2 d128:f16 cfg2:f32 [--ab oldbwd]"""
import ctypes, sys, pathlib, statistics
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "flash-attention-ipu_amd")]
import torch
from flash_attention_amd._impl import _native
CFG = {"cfg2": (128, 4096, 64), "d128": (64, 4096, 128), "short": (512, 1024, 64), "cfg5": (256, 8192, 64), "gpt": (96, 1024, 64),
       "b1l8k": (1, 8192, 64), "g2l16k": (2, 16384, 64), "b1l32k": (1, 32768, 64), "g4l4k": (4, 4096, 64), "g8l2k": (8, 2048, 64), "g4l8kd128": (4, 8192, 128)}
DT = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}
args = sys.argv[1:]
ab = []
if "--ab" in args:
    i = args.index("--ab")
    ab = args[i + 1].split(",")
    args = args[:i] + args[i + 2:]
libs = {"": _native.load_native_library()}
for name in ab:
    lib = ctypes.CDLL(str(ROOT / "flash-attention-ipu_amd" / "build" / "variants" / f"libfa_{name}.so"))
    _native._declare(lib)
    libs[name] = lib
for spec in args or ["cfg2"]:          # cfg2 | cfg2:f32 | d128:f16 ...
    name, _, dtn = spec.partition(":")
    G, L, D = CFG[name]
    dt = DT[dtn or "bf16"]
    qkv = (torch.randn(3, G, L, D, device="cuda") * 0.5).to(dt)
    dout = torch.randn(G, L, D, device="cuda").to(dt)
    out = torch.empty(G, L, D, device="cuda", dtype=dt)
    lse = torch.empty(G, L, device="cuda", dtype=torch.float32)
    dsum = torch.empty(G, L, device="cuda", dtype=torch.float32)
    dqkv = torch.empty_like(qkv)
    arr = ((ctypes.c_int64 * 3) * 8)()
    for i in range(8):
        arr[i][0], arr[i][1], arr[i][2] = L * D, 0, D
    def fwd(): _native.fwd_packed(qkv, out, lse, 8, 8)
    def bwd(lib):
        rc = lib.fa_bwd(qkv[0].data_ptr(), qkv[1].data_ptr(), qkv[2].data_ptr(), out.data_ptr(), dout.data_ptr(), lse.data_ptr(),
                        dqkv[0].data_ptr(), dqkv[1].data_ptr(), dqkv[2].data_ptr(), dsum.data_ptr(), G, L, D, 1, 1, arr,
                        _native.FA_DTYPE[dt], ctypes.c_float(1.0), 1, torch.cuda.current_stream().cuda_stream)
        assert rc == 0
    def t(fn, n=10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    fwd(); t(fwd, 3)
    ref = None
    for k, lib in libs.items():
        bwd(lib); t(lambda: bwd(lib), 3)
        g = dqkv.clone()
        if ref is None: ref = g
        elif not torch.equal(g, ref): print(f"note: lib '{k}' gradients differ from the shipped library: max|d|={(g.float() - ref.float()).abs().max().item():.3e}")
    tf = statistics.median(t(fwd) for _ in range(5))
    res = {k: [] for k in libs}
    for _ in range(7):
        for k, lib in libs.items():
            res[k].append(t(lambda: bwd(lib)))
    fl = 2.0 * G * L * L * D
    for k in libs:
        tb = statistics.median(res[k])
        print(f"{spec} [{k or 'shipped'}]: fwd {tf*1e3:8.1f} us {fl/tf/1e9:7.1f} TF | bwd {tb*1e3:8.1f} us {2.5*fl/tb/1e9:7.1f} TF (5*G*L^2*D) | bwd/fwd {tb/tf:.2f}")
