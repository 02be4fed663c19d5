"""Developer aid: interleaved A/B timing of kernel builds/variants in ONE process (cdna guide rule 24).

usage: python scripts/sweep.py [cfg] SPEC [SPEC ...]
  SPEC = "[lib:]K=pipe,W=4,S=4"   lib = name of build/variants/libfa_<lib>.so (default: the shipped library)
         K kernel (dma|pipe|pipe16|kh), W waves per workgroup, S groups per L2 bunch, Q 32-row query blocks per wave (1|2),
         P kv split groups per workgroup (dma: 2|4), R q-tile pairs per workgroup (dma: 0|1, default auto), B placement-balanced order (0|1, default auto),
Each timing = N back-to-back launches between two events; rounds are interleaved across specs.
"""
import ctypes, os, sys, pathlib, statistics
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "flash-attention-ipu_amd")]
import torch

CFG = {"cfg2": (128, 4096, 64, torch.bfloat16, 8, 8), "cfg2h": (128, 4096, 64, torch.float16, 8, 8),
       "cfg3": (16, 32768, 128, torch.bfloat16, 16, 16), "d128": (64, 4096, 128, torch.bfloat16, 8, 8),
       "cfg5": (256, 8192, 64, torch.bfloat16, 16, 8), "short": (512, 1024, 64, torch.bfloat16, 4, 4), "l1k": (2048, 1024, 64, torch.bfloat16, 8, 8), "g8l4k": (8, 4096, 64, torch.bfloat16, 8, 8), "g8l8k": (8, 8192, 64, torch.bfloat16, 8, 8), "g4l8kd128": (4, 8192, 128, torch.bfloat16, 8, 8), "g32l2k": (32, 2048, 64, torch.bfloat16, 8, 8),
       "small1": (8, 2048, 64, torch.bfloat16, 1, 1), "small2": (16, 1024, 64, torch.bfloat16, 1, 1),
       "small3": (32, 4096, 64, torch.bfloat16, 1, 1), "small4": (12, 1024, 64, torch.bfloat16, 1, 1),
       "gpt": (96, 1024, 64, torch.bfloat16, 1, 1), "long64": (16, 32768, 64, torch.bfloat16, 16, 16), "g64l8k": (64, 8192, 64, torch.bfloat16, 8, 8), "g32l16k": (32, 16384, 64, torch.bfloat16, 8, 8), "g32l4k": (32, 4096, 64, torch.bfloat16, 8, 8), "g256l2k": (256, 2048, 64, torch.bfloat16, 8, 8), "d128l2k": (128, 2048, 128, torch.bfloat16, 8, 8), "d128l8k": (32, 8192, 128, torch.bfloat16, 8, 8), "d96l8k": (32, 8192, 96, torch.bfloat16, 8, 8), "d96l32k": (16, 32768, 96, torch.bfloat16, 16, 16), "d32l8k": (64, 8192, 32, torch.bfloat16, 8, 8), "d128g16": (16, 4096, 128, torch.bfloat16, 8, 8), "d128h": (64, 4096, 128, torch.float16, 8, 8), "g64l4k": (64, 4096, 64, torch.bfloat16, 8, 8), "small5": (4, 8192, 128, torch.bfloat16, 1, 1),
       "d192": (128, 2048, 192, torch.bfloat16, 1, 1), "d256": (128, 2048, 256, torch.bfloat16, 1, 1), "d256l8k": (32, 8192, 256, torch.bfloat16, 1, 1), "d192l8k": (32, 8192, 192, torch.bfloat16, 1, 1), "d256h": (128, 2048, 256, torch.float16, 1, 1),
       "g64l1k": (64, 1024, 64, torch.bfloat16, 1, 1), "g48l2k": (48, 2048, 64, torch.bfloat16, 1, 1), "g24l4k": (24, 4096, 64, torch.bfloat16, 1, 1),
       "g96l1kd128": (96, 1024, 128, torch.bfloat16, 1, 1), "g128l1k": (128, 1024, 64, torch.bfloat16, 1, 1), "g192l1k": (192, 1024, 64, torch.bfloat16, 1, 1)}
args = sys.argv[1:]
cfg = "cfg2"
if args and args[0] in CFG:
    cfg = args.pop(0)
G, L, D, dt, nq, nk = CFG[cfg]
specs = args or ["K=tile,W=4", "K=pipe,W=4"]
LONG = int(os.environ.get("SWEEP_N", "20"))
qkv = torch.randn(3, G, L, D, device="cuda").to(dt)
out = torch.empty(G, L, D, device="cuda", dtype=dt)
flops = 2.0 * G * L * L * D
dcode = {torch.float16: 0, torch.bfloat16: 1}[dt]
libs = {}

def getlib(name):
    if name not in libs:
        path = ROOT / "flash-attention-ipu_amd" / "build" / ("libflash_attention.so" if name == "" else f"variants/libfa_{name}.so")
        lib = ctypes.CDLL(str(path))
        lib.fa_fwd_qkv_packed.restype = ctypes.c_int
        lib.fa_fwd_qkv_packed.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int64] * 3 + [ctypes.c_int32] * 3 + [ctypes.c_void_p]
        lib.fa_debug_set_option.restype = ctypes.c_int
        lib.fa_debug_set_option.argtypes = [ctypes.c_char_p, ctypes.c_int32]
        libs[name] = lib
    return libs[name]

def parse(spec):
    name, _, rest = spec.rpartition(":")
    return name, rest

def setknobs(lib, rest):
    """knobs go through fa_debug_set_option of THAT library (the environment is only read once, at load)"""
    want = {"kernel": 0, "waves": 0, "sched": 0, "qb": 0, "split": 0, "pair": -1, "balance": -1}
    for kv in rest.split(","):
        if kv:
            k, val = kv.split("=")
            name = {"K": "kernel", "W": "waves", "S": "sched", "Q": "qb", "P": "split", "R": "pair", "B": "balance"}[k]
            want[name] = (ord("x") if val in ("pipe16", "x") else ord("h") if val in ("kh", "h") else ord(val[0])) if name == "kernel" else int(val)
    for name, val in want.items():
        assert lib.fa_debug_set_option(name.encode(), val) == 0, (name, val)

def timeit(spec, n):
    name, rest = parse(spec)
    lib = getlib(name)
    setknobs(lib, rest)
    stream = torch.cuda.current_stream().cuda_stream
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        rc = lib.fa_fwd_qkv_packed(qkv.data_ptr(), out.data_ptr(), None, G, L, D, nq, nk, dcode, stream)
        assert rc == 0, rc
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

ref = None
for s in specs:
    timeit(s, 3)
    o = out.clone()
    if ref is None:
        ref = o
    elif not torch.equal(o, ref):
        print(f"note: {s} differs from {specs[0]}: max|d|={(o.float()-ref.float()).abs().max().item():.3e}")
res = {s: [] for s in specs}
for rnd in range(7):
    for s in specs:
        res[s].append(timeit(s, LONG))
for s in specs:
    ts = res[s]
    print(f"{cfg} {s:34s} median {statistics.median(ts)*1e3:8.1f} us  min {min(ts)*1e3:8.1f} us  "
          f"{flops/statistics.median(ts)/1e9:8.1f} TF (max {flops/min(ts)/1e9:7.1f})", flush=True)
