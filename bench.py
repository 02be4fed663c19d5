#!/usr/bin/env python3
"""Headline benchmark: forward attention TFLOP/s, causal, L=4096, D=64 (BASELINE.json configs[1]).

    python bench.py --gpus N --steps K --warmup W [--config cfg2|cfg2_f16|cfg3|cfg5_shard|cfg5]

A step is one pass of the hot path over one batch of synthetic packed qkv that is already resident in HBM:
`flash_attention_qkv_packed(qkv[3, 128, 4096, 64] bf16, 8, 8)` per GPU (batch 8 x 16 heads; chunk counts = the
reference heuristic for L=4096, D=64).

N > 1: one process per GPU over RCCL.  Started either by the driver's `python -m torch.distributed.run ... bench.py
--gpus N` (WORLD_SIZE set) or plainly as `python bench.py --gpus N`: then this process - BEFORE it touches the GPU -
starts N fresh rank processes itself (torch.distributed.run as a child; never an exec) and relays rank 0's JSON
line and the exit code.
  * default (cfg2, "scaling": "weak"): every rank owns its own 128 groups (groups are independent, FA.cpp:186); the
    only exchange is the gather of O, double-buffered so the gather of step s overlaps the kernel of step s+1.
  * right after that headline, in the same launch and process group, BASELINE configs[4] ("scaling": "strong": G = 64*32 = 2048 groups
    in total, L = 8192, D = 64, G/N groups per rank, cut into slabs so the gather of slab s overlaps the kernel of slab s+1) runs as
    `secondary.cfg5_strong` {value, kernel_only_value, ms_per_step, exchange_only_ms, achieved_GBps_per_link, expected_link_bound_ms,
    n1_anchor_kernel_tflops, scaling_efficiency, kernel_scaling_efficiency} with its own exchange selection (strong_record);
  * --config cfg5 runs configs[4] as the headline instead.
`value` is end-to-end (gather included); `roofline.achieved` is kernel-only (HIP events around the launches) and
`kernel_only_value` repeats it as a whole-job figure; `scaling_efficiency` = value / (N x the one-GPU kernel-only rate of the launch shape,
measured in the same run on every rank): 1.0 = the exchange is free.

N = 1 also reports, in `secondary`, a small-grid shape (split-KV plan), the backward at cfg2 and on two grids that cannot fill the chip
(`bwd_small_grids`: the library's split groups + balanced dQ units against the plain grid), and the other single-GPU configs of BASELINE.json (cfg3 long context, the cfg5
per-GPU shard shape, `cfg5_full` = ALL 2048 groups of configs[4] on this one GPU - the anchor of its 1/2/4/8 strong-scaling curve - and cfg4: a
GPT-2-small forward step through the automatic SDPA override with the attention share) - the headline stays cfg2.

FLOPs = 2*G*L^2*D (causal convention); roofline = MFMA, dense bf16 peak 2.5 PFLOP/s.
The cpu_baseline leg times the oracle's dense restatement of the reference CPU path
(flash_attention_ipu/_impl/flash_attention_qkv_packed.py:48-53) on the host cores of the same box, fp32 and bf16.
"""

import argparse
import hashlib
import json
import os
import pathlib
import re
import socket
import subprocess
import sys
import threading
import time

ROOT = pathlib.Path(__file__).resolve().parent
for _p in (ROOT, ROOT / "flash-attention-ipu_amd", ROOT / "demo"):
    if str(_p) not in sys.path:
        sys.path.insert(0, str(_p))

CONFIGS = {
    # name: (G per GPU, L, D, dtype, (nq, nk))  -  per-GPU work fixed as N grows ("weak")
    "cfg2": (128, 4096, 64, "bf16", (8, 8)),         # BASELINE configs[1] - the headline
    "cfg2_f16": (128, 4096, 64, "f16", (8, 8)),
    "cfg3": (16, 32768, 128, "bf16", (16, 16)),      # configs[2] long context
    "cfg5_shard": (256, 8192, 64, "bf16", (16, 8)),  # configs[4] per-GPU shard at 8 GPUs
    "small_grid": (8, 4096, 64, "bf16", (8, 8)),     # not a BASELINE config: 256 workgroups on 256 CUs (split-KV plan)
    "l1024": (512, 1024, 64, "bf16", (4, 4)),        # not a BASELINE config: the short-sequence regime on a full grid
    "few_groups": (2, 32768, 64, "bf16", (16, 16)),  # not a BASELINE config: batch 1 x 2 heads, long context - fewer groups than XCDs (q-tile subsets)
    "cfg4_attn": (96, 1024, 64, "bf16", (4, 4)),     # configs[3]'s attention call alone: SDPA override on head-inner [8,12,1024,64] views
    "nb_65536": (8, 65536, 64, "f16", (64, 64)),     # the reference notebook's longest demonstrated context (nb:939, :952-963)
    "cfg2_sdpa": (128, 4096, 64, "bf16", (8, 8)),    # the headline shape through the SDPA override ([8,16,4096,64], scale 1/sqrt(D): sdpa.py:55-59)
    # total work fixed ("strong"): G is the TOTAL group count, G/N per rank
    "cfg5": (2048, 8192, 64, "bf16", (16, 8)),       # configs[4]: batch 64 x heads 32, seq 8192, head_dim 64
}
STRONG = {"cfg5"}
PEAK_TFLOPS = 2500.0   # dense bf16/f16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md


_emit_lock = threading.Lock()
_emitted = [False]
_json_fd = [None]       # the process's ORIGINAL stdout, once keep_stdout_for_the_json_line() has pointed fd 1 at stderr


def keep_stdout_for_the_json_line() -> None:
    """stdout carries rank 0's ONE JSON line and nothing else: RCCL prints a five-line banner (version, host, library path) on fd 1 when its
    first communicator comes up, and any library may chat there.  Everything that writes to fd 1 from here on - C libraries included - lands on
    stderr; the JSON line goes to a private duplicate of the original stdout."""
    if _json_fd[0] is None:
        sys.stdout.flush()
        _json_fd[0] = os.dup(1)
        os.dup2(2, 1)


def emit_once(line) -> None:
    """rank 0's ONE JSON line (the N > 1 watchdog and the normal path may race for it)."""
    with _emit_lock:
        if not _emitted[0]:
            _emitted[0] = True
            text = (json.dumps(line) + "\n").encode()
            if _json_fd[0] is None:
                sys.stdout.flush()
                os.write(1, text)
            else:
                os.write(_json_fd[0], text)


def csrc_fingerprint(kernels=None) -> str:
    """sha256 over kernel sources: ties a profiled HBM-traffic figure to the code that was profiled.  With `kernels` (the kernel names of a
    PMC record, e.g. "fa_fwd_pipe_kernel" or "fa_bwd_dq_kernel + fa_bwd_dkdv_kernel") only the device sources those kernels are built
    from count - the .hip files that define them and every csrc header / .hip they include, transitively - so an edit to another kernel's
    file does not orphan the figure; without it, every file under csrc/."""
    csrc = ROOT / "flash-attention-ipu_amd" / "csrc"
    files = sorted(f for f in csrc.glob("*") if f.suffix in (".hip", ".h", ".cpp"))
    if kernels:
        names = set(re.findall(r"fa_\w+_kernel", kernels))
        text = {f: f.read_text() for f in files}
        todo = [f for f in files if f.suffix == ".hip" and any(re.search(r"void\s+" + n + r"\s*\(", text[f]) for n in names)]
        seen = set()
        while todo:
            f = todo.pop()
            if f in seen:
                continue
            seen.add(f)
            todo += [csrc / inc for inc in re.findall(r'#include\s+"([^"]+)"', text[f]) if (csrc / inc) in text]
        files = sorted(seen)
    h = hashlib.sha256()
    for f in files:
        h.update(f.name.encode())
        h.update(f.read_bytes())
    return h.hexdigest()[:16]


def self_launch(args) -> int:
    """`python bench.py --gpus N` without a launcher: N fresh rank processes, started before any GPU call."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(pathlib.Path(__file__).resolve())] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode      # stdout/stderr inherited: rank 0's JSON line passes through


def cpu_baseline(L, D, budget_s=9.0):
    """Reference CPU path (dense torch, restated in oracle/) on this box's host cores: fp32 leg (the reported
    value) and bf16 leg, each a bounded sample."""
    import torch
    from oracle import dense_causal_attention

    # every core the box GRANTS this process: its affinity mask, cut to the cgroup's CPU quota where one is set (threads beyond a
    # quota are throttled, not run: oversubscription only slows the sample down)
    cores = os.cpu_count() or 1
    granted_by = "os.cpu_count"
    try:
        cores = len(os.sched_getaffinity(0))
        granted_by = "sched_getaffinity"
    except AttributeError:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = pathlib.Path(path).read_text().split()
            quota = int(txt[0]) if txt[0] != "max" else -1
            period = int(txt[1]) if len(txt) > 1 else int(pathlib.Path("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read_text())
            if quota > 0 and period > 0 and -(-quota // period) < cores:
                cores = max(1, -(-quota // period))
                granted_by = f"cgroup quota ({path})"
            break
        except (OSError, ValueError, IndexError):
            continue
    torch.set_num_threads(cores)
    slab = 8
    gen = torch.Generator().manual_seed(0)
    qkv32 = torch.randn(3, slab, L, D, generator=gen)
    flops = 2.0 * slab * L * L * D
    legs = {}
    for name, qkv in (("fp32", qkv32), ("bf16", qkv32.bfloat16())):
        dense_causal_attention(qkv)  # warm-up
        times = []
        t_start = time.perf_counter()
        while time.perf_counter() - t_start < budget_s and len(times) < 30:
            t0 = time.perf_counter()
            dense_causal_attention(qkv)
            times.append(time.perf_counter() - t0)
        times.sort()
        med = times[len(times) // 2]
        legs[name] = {"value": round(flops / med / 1e12, 5), "reps": len(times), "median_ms_per_slab": round(med * 1e3, 1)}
    return {
        "value": legs["fp32"]["value"], "unit": "TFLOP/s", "cores": cores, "cores_granted_by": granted_by, "kind": "port",
        "sample": f"{legs['fp32']['reps']} reps of a {slab}-group slab (of 128) at L={L}, D={D}, fp32 dense torch "
                  f"restatement of fa.py:48-53, median {legs['fp32']['median_ms_per_slab']} ms/slab, scaled linearly in G",
        "bf16": {"value": legs["bf16"]["value"], "unit": "TFLOP/s",
                 "sample": f"{legs['bf16']['reps']} reps of the same slab in bf16 (the reference CPU path run on bf16 "
                           f"tensors), median {legs['bf16']['median_ms_per_slab']} ms/slab"},
    }


def tool_preloaded(env=None) -> bool:
    """True when ANYTHING is preloaded into this process or a ROCm tool interface is configured: a profiler's preloaded library (rocprofv3
    with --pmc does) initialises the GPU before main(), and a GPU box refuses a fork + exec from a process that has - 24 refused `execvp`
    in round 5 (gpurun_out/.graft_exec_refused).  Deliberately broad: any non-empty LD_PRELOAD, HSA_TOOLS_LIB, any ROCP_* / ROCPROFILER_* /
    ROCPROF_* / ROCTRACER_* variable - not a match on the names one SDK version happens to use."""
    env = os.environ if env is None else env
    if env.get("LD_PRELOAD", "").strip() or env.get("HSA_TOOLS_LIB", "").strip():
        return True
    return any(k.startswith(("ROCP_", "ROCPROFILER_", "ROCPROF_", "ROCTRACER_")) for k in env)


def sysfs_max_sclk_mhz(index, root="/sys/class/drm"):
    """Highest engine-clock level of the index-th AMD GPU from sysfs (`pp_dpm_sclk`: "0: 500Mhz\n1: 2400Mhz *"): a FILE READ - no child
    process, so it is safe under a profiler and after the first HIP call.  None where the file is absent or unreadable."""
    try:
        cards = sorted((p for p in pathlib.Path(root).glob("card[0-9]*") if (p / "device" / "pp_dpm_sclk").exists()),
                       key=lambda p: int(re.sub(r"\D", "", p.name) or 0))
        vals = [int(v) for v in re.findall(r"(\d+)\s*Mhz", (cards[index] / "device" / "pp_dpm_sclk").read_text(), re.I)]
        return max(vals) if vals else None
    except (OSError, IndexError, ValueError):
        return None


def smi_clock_mhz(index):
    """Engine clock (top sclk level) of device `index` in MHz, or None - decoration for `config.device`, never a measurement.  sysfs first
    (no child process at all); `rocm-smi -s` - a script, i.e. a fork + exec - only when nothing is preloaded and no tool interface is
    configured (tool_preloaded), and only BEFORE this process makes its first HIP call (main() asks first thing)."""
    mhz = sysfs_max_sclk_mhz(index)
    if mhz is not None or tool_preloaded():
        return mhz
    try:   # the highest SUPPORTED sclk level (`--showclocks` reports the current speeds: an idle device shows 95 MHz there, and its 2 000 MHz is the memory clock)
        return parse_max_sclk(subprocess.run(["rocm-smi", "-d", str(index), "-s"], capture_output=True, text=True, timeout=20).stdout)
    except Exception:  # noqa: BLE001
        return None


def parse_max_sclk(text):
    """the highest level of the 'Supported sclk frequencies' block of `rocm-smi -s`, in MHz, or None"""
    m = re.search(r"Supported sclk frequencies(.*?)(?:Supported \w+ frequencies|$)", text, re.S)
    vals = [int(v) for v in re.findall(r"(\d+)Mhz", m.group(1))] if m else []
    return max(vals) if vals else None


def device_clock_mhz(torch, dev, smi_mhz=None):
    """Engine clock of the device in MHz: torch's device properties where they carry it, else what rocm-smi listed at start-up, else None."""
    props = torch.cuda.get_device_properties(dev)
    for attr in ("clock_rate", "clockRate"):
        v = getattr(props, attr, 0) or 0
        if v:
            return int(v) // 1000
    return smi_mhz


def make_op(name, dev, seed=1, backward=False):
    """(fn, flops, kernel names, description) for one launch of a config: the packed op (default), the SDPA override on the
    head-inner views of a fused [B, T, 3C] projection (cfg4_attn: exactly GPT-2 small's attention call), or fa_bwd."""
    import torch
    import flash_attention_amd as fa
    from flash_attention_amd._impl import _native

    G, L, D, dname, (nq, nk) = CONFIGS[name]
    dtype = {"bf16": torch.bfloat16, "f16": torch.float16}[dname]
    gen = torch.Generator(device=dev).manual_seed(seed)
    if backward:
        qkv = (torch.randn(3, G, L, D, device=dev, generator=gen) * 0.5).to(dtype)
        dout = torch.randn(G, L, D, device=dev, generator=gen).to(dtype)
        from flash_attention_amd._impl.flash_attention_qkv_packed import flash_attention_qkv_packed_with_lse
        out, lse = flash_attention_qkv_packed_with_lse(qkv, nq, nk)
        dqkv = torch.empty_like(qkv)
        fn = lambda: _native.bwd(qkv[0], qkv[1], qkv[2], out, dout, lse, dqkv[0], dqkv[1], dqkv[2], G, L, D, 1, 1.0, True)  # noqa: E731
        return fn, 5.0 * G * L * L * D, "fa_bwd_dq_kernel + fa_bwd_dkdv_kernel", \
            f"{name}: gradients of packed qkv [3,{G},{L},{D}] {dname}, causal (fa_bwd: dQ + dK/dV kernels)", None
    if name == "cfg2_sdpa":
        import torch.nn.functional as F
        import flash_attention_amd.auto  # noqa: F401
        B, H = 8, 16
        q, k, v = (torch.randn(B, H, L, D, device=dev, generator=gen).to(dtype) for _ in range(3))

        def fn():
            with torch.no_grad():
                return F.scaled_dot_product_attention(q, k, v, is_causal=True)
        plan = _native.plan(G, L, D, dtype, nq, nk, softmax_scale=D ** -0.5)
        return fn, 2.0 * G * L * L * D, kernel_name(plan), \
            f"{name}: F.scaled_dot_product_attention(is_causal=True) through the override on [{B},{H},{L},{D}] {dname}, scale 1/sqrt(D)", plan
    if name == "cfg4_attn":
        import torch.nn.functional as F
        import flash_attention_amd.auto  # noqa: F401
        B, H = 8, 12
        proj = torch.randn(B, L, 3 * H * D, device=dev, generator=gen).to(dtype)
        q, k, v = (t.view(B, L, H, D).transpose(1, 2) for t in proj.split(H * D, dim=2))

        def fn():
            with torch.no_grad():
                return F.scaled_dot_product_attention(q, k, v, is_causal=True)
        plan = _native.plan(G, L, D, dtype, nq, nk, softmax_scale=D ** -0.5)
        return fn, 2.0 * G * L * L * D, kernel_name(plan), \
            f"{name}: F.scaled_dot_product_attention(is_causal=True) through the override on head-inner [{B},{H},{L},{D}] {dname} views of a [B,T,3C] projection", plan
    qkv = torch.randn(3, G, L, D, device=dev, generator=gen).to(dtype)
    plan = _native.plan(G, L, D, dtype, nq, nk)
    return (lambda: fa.flash_attention_qkv_packed(qkv, nq, nk)), 2.0 * G * L * L * D, kernel_name(plan), \
        f"{name}: packed qkv [3,{G},{L},{D}] {dname}, causal, num_chunks=({nq},{nk})", plan


def per_step_ms(fn, steps):
    """Per-step durations from HIP events between consecutive steps (a pass of its own: the headline's timed region carries no
    events inside it) -> (median, min) in ms."""
    import torch

    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    ev[0].record()
    for i in range(steps):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    d = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(steps))
    return d[len(d) // 2], d[0]


def time_kernel(fn, steps, warmup, ramp_ms=0.0):
    """avg ms per call of fn() on the current stream: untimed ramp + warm-up, then HIP events around `steps` calls."""
    import torch

    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    i = 0
    while (time.perf_counter() - t0) * 1e3 < ramp_ms:
        fn()
        i += 1
        if i % 16 == 0:
            torch.cuda.synchronize()
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


def secondary_attention(name, dev, steps=20, warmup=5, backward=False):
    """One of the other configs through the same op (or fa_bwd): a COMPACT record - the driver keeps an 8 KB tail of stdout, and the
    whole JSON line must fit in it (round-3 verdict: a 10.9 KB line lost BASELINE config 3).  value = TFLOP/s on the causal convention
    (5 G L^2 D for the backward), from the MEDIAN of three `steps`-long timed regions: each record starts from a chip that idled while
    its tensors were built, and ~0.35-0.5 s after load onset the device stalls once for ~40 ms."""
    G, L, D, dname, (nq, nk) = CONFIGS[name]
    fn, flops, kname, desc, plan = make_op(name, dev, backward=backward)
    means = sorted([time_kernel(fn, steps, warmup, ramp_ms=60.0)] + [time_kernel(fn, steps, 0) for _ in range(2)])
    ms = means[1]
    med, mn = per_step_ms(fn, steps)
    tf = flops / (ms * 1e-3) / 1e12
    traffic, _ = profiled_traffic(name + ("_bwd" if backward else ""), plan, kname)
    rec = {"value": round(tf, 1), "frac": round(tf / PEAK_TFLOPS, 4), "ms": round(ms, 4), "median_ms": round(med, 4), "min_ms": round(mn, 4),
           "kernel": kname.replace("fa_", "").replace("_kernel", ""), "dtype": dname,
           "shape": [3, G, L, D] if name not in ("cfg2_sdpa", "cfg4_attn") else desc.split(" on ")[1].split(" ")[0 if name == "cfg2_sdpa" else 1],
           "traffic": traffic, "alg_bytes": (8 if backward else 4) * G * L * D * 2}
    if backward:
        rec["flops"] = "5GL^2D"
    elif plan is not None:
        rec["tile"] = f"{plan.block_q}x{plan.block_kv}/{plan.waves}w" + (f"/split{32 * plan.waves * plan.qb // plan.block_q}" if 32 * plan.waves * plan.qb > plan.block_q else "") + (f"/xs{plan.xs}" if plan.xs > 1 else "")
    return rec


def secondary_backward_small_grids(dev, steps=10, warmup=3):
    """fa_bwd on two causal grids that cannot fill the chip (batch-1 long-context shapes: the round-5 verdict's item 4): the library's own choice -
    split groups inside the workgroups, balanced dQ units - against the plain grid of unsplit kernels in the same process.
    value = TFLOP/s on 5 G L^2 D."""
    import torch
    from flash_attention_amd._impl import _native
    from flash_attention_amd._impl.flash_attention_qkv_packed import flash_attention_qkv_packed_with_lse

    rec = {}
    for G, L, D in ((1, 8192, 64), (4, 8192, 128)):
        gen = torch.Generator(device=dev).manual_seed(7)
        qkv = (torch.randn(3, G, L, D, device=dev, generator=gen) * 0.5).to(torch.bfloat16)
        dout = torch.randn(G, L, D, device=dev, generator=gen).to(torch.bfloat16)
        out, lse = flash_attention_qkv_packed_with_lse(qkv, 1, 1)
        dqkv = torch.empty_like(qkv)
        fn = lambda: _native.bwd(qkv[0], qkv[1], qkv[2], out, dout, lse, dqkv[0], dqkv[1], dqkv[2], G, L, D, 1, 1.0, True)  # noqa: E731
        ms = sorted([time_kernel(fn, steps, warmup, ramp_ms=60.0)] + [time_kernel(fn, steps, 0) for _ in range(2)])[1]
        with _native.debug_options(bsplit=1, bbal=0):
            plain = sorted(time_kernel(fn, steps, warmup if i == 0 else 0) for i in range(3))[1]
        tf = 5.0 * G * L * L * D / (ms * 1e-3) / 1e12
        rec[f"[{G},{L},{D}]"] = {"value": round(tf, 1), "frac": round(tf / PEAK_TFLOPS, 4), "ms": round(ms, 4), "plain_grid_ms": round(plain, 4),
                                 "speedup": round(plain / ms, 3)}
    return rec


def secondary_gpt2_forward(dev, batch=8, steps=10, warmup=3):
    """BASELINE configs[3]: GPT-2 small (12 layers, 12 heads, n_embd 768, T = 1024) forward step through the
    automatic SDPA override; whole-step time and the attention share (HIP events around the 12 op calls)."""
    import torch
    import torch.nn.functional as F
    import flash_attention_amd.auto  # noqa: F401  (patches F.scaled_dot_product_attention)
    from model import TinyGPT, gpt2_init_

    T, H, hs, layers = 1024, 12, 64, 12
    torch.manual_seed(0)
    model = gpt2_init_(TinyGPT(vocab=50304, block=T, n_layer=layers, n_head=H, n_embd=H * hs)).to(dev).eval()
    idx = torch.randint(0, 50304, (batch, T), device=dev)
    patched = F.scaled_dot_product_attention
    spans = []

    def timed_sdpa(*a, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = patched(*a, **kw)
        e1.record()
        spans.append((e0, e1))
        return out

    def fwd():
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            return model(idx)

    def run(n, sdpa):
        """median step time (HIP events between consecutive steps: robust against the device's one-time stall)"""
        F.scaled_dot_product_attention = sdpa
        try:
            for _ in range(warmup):
                fwd()
            spans.clear()
            return per_step_ms(fwd, n)[0]
        finally:
            F.scaled_dot_product_attention = patched

    step_ms = run(steps, patched)
    run(steps, timed_sdpa)
    per_call = sorted(a.elapsed_time(b) for a, b in spans)
    attn_ms = layers * per_call[len(per_call) // 2]          # 12 calls per step x the median call
    stock_ms = run(steps, patched.__wrapped__)
    # the same step captured once in a HIP graph and replayed (the override's launches are capture-safe: caller's stream, no host sync):
    # what a serving loop would run; a failed capture only drops the field
    graph_ms = None
    try:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fwd()
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            fwd()
        for _ in range(warmup):
            graph.replay()
        graph_ms = per_step_ms(graph.replay, steps)[0]
        del graph
    except Exception:  # noqa: BLE001
        graph_ms = None
    flops = layers * 2.0 * batch * H * T * T * hs
    tf = flops / (attn_ms * 1e-3) / 1e12
    from flash_attention_amd._impl import _native
    from flash_attention_amd._impl.scaled_dot_product_attention import chunk_heuristic

    plan = _native.plan(batch * H, T, hs, torch.bfloat16, *chunk_heuristic(T, hs), softmax_scale=hs ** -0.5)
    return {"value": round(step_ms, 3), "unit": "ms/step", "stock_sdpa_ms": round(stock_ms, 3), "hip_graph_ms": None if graph_ms is None else round(graph_ms, 3),
            "tokens_per_s": round(batch * T / (step_ms * 1e-3)),
            "attn_ms": round(attn_ms, 3), "attn_share": round(attn_ms / step_ms, 4), "attn_tflops": round(tf, 1), "attn_frac": round(tf / PEAK_TFLOPS, 4),
            "kernel": kernel_name(plan).replace("fa_", "").replace("_kernel", ""), "calls_per_step": layers,
            "shape": f"GPT-2 small fwd, B={batch} T={T} 12 layers x 12 heads, bf16 autocast, through flash_attention_amd.auto"}


def strong_record(name, ctx, steps, warmup, slabs, gather):
    """A strong-scaling config (BASELINE configs[4]: `cfg5`) as ONE compact record, measured the way the `--config cfg5` headline is:
    CONFIGS[name] groups IN TOTAL, G/W per rank, `slabs` launches per step through `SlabGather` (slab-cyclic ownership, the gather of
    slab s under the kernel of slab s + 1), K steps between barrier + synchronize, maximum over ranks.  EVERY rank calls this (it
    carries collectives); every rank returns the same record.

      value                whole-job TFLOP/s on 2 G L^2 D, exchange included          ms_per_step   the step it comes from
      kernel_only_value    W x the rate of a gather-free repeat of the same launches  exchange_only_ms   the step's `slabs` gathers alone
      achieved_GBps_per_link   one shard per link and direction / exchange_only_ms    expected_link_bound_ms   shard / 76 GB/s
      n1_anchor_kernel_tflops  this GPU's kernel-only rate on the launch shape a ONE-GPU run of the config uses (G_total / slabs groups)
      scaling_efficiency       value / (W x anchor);  kernel_scaling_efficiency = kernel_only_value / (W x anchor)
    At W = 1 (`cfg5_full`, the anchor of the 1/2/4/8 curve) nothing is exchanged and the anchor is the run itself."""
    torch, dist, dev, world, rank, Ev, sync = (ctx[k] for k in ("torch", "dist", "dev", "world", "rank", "Ev", "sync"))
    import flash_attention_amd as fa
    from flash_attention_amd._impl import _native
    from flash_attention_amd.distributed import SlabGather, select_gather_mode

    Gc, L, D, dname, (nq, nk) = CONFIGS[name]
    if Gc % (world * slabs) != 0:
        raise ValueError(f"{name}: {Gc} groups do not split over {world} ranks x {slabs} slabs")
    G, dtype = Gc // world, ctx["DT"][dname]
    gs = G // slabs
    gen = torch.Generator(device=dev).manual_seed(1000 + rank)
    qkv = torch.empty(3, G, L, D, dtype=dtype, device=dev)
    for t3 in range(3):                       # (slab by slab: the fp32 draw of the whole shard would be twice its size again)
        for s_ in range(slabs):
            qkv[t3, s_ * gs:(s_ + 1) * gs] = torch.randn(gs, L, D, device=dev, generator=gen).to(dtype)
    do_gather = ctx["exchange"] and ctx["dist_on"]
    mode, table, selection = gather, None, "forced (--gather)"
    if do_gather and gather == "auto":
        q_sel = qkv[:, :gs].contiguous()
        step_sel = lambda: fa.flash_attention_qkv_packed(q_sel, nq, nk)   # noqa: E731
        try:
            mode, table = select_gather_mode(step_sel(), step=step_sel, reps=3)
            selection = "auto: fastest kernel + exchange step on this config's slab (maximum over ranks)"
        except Exception as e:  # noqa: BLE001
            mode, selection = "allgather", f"auto FAILED ({type(e).__name__}: {e})"[:160] + " -> all-gather"
        del q_sel
    elif gather == "auto":
        mode = "allgather"
    sg = SlabGather(G, L, D, dtype, dev, slabs=slabs, mode=mode, gather=do_gather, gather_at_world1=ctx["force_dist"])

    def barrier():
        if ctx["dist_on"]:
            dist.barrier()

    for _ in range(max(1, warmup)):
        sg.run(qkv, nq, nk, wait=False)
    sg.wait()
    sync()
    barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        sg.run(qkv, nq, nk, wait=False)
    sg.wait()
    sync()
    barrier()
    sync()
    elapsed = time.perf_counter() - t0

    def timed(fn, n):
        fn()
        sync()
        e0, e1 = Ev(), Ev()
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        sync()
        return e0.elapsed_time(e1) / n

    n_k = max(3, steps // 2)
    kernel_ms = elapsed / steps * 1e3
    x_ms = 0.0
    if sg.gather:      # the same launches with no exchange, and the step's exchanges with no launches
        sgk = SlabGather(G, L, D, dtype, dev, slabs=slabs, gather=False)
        kernel_ms = timed(lambda: sgk.run(qkv, nq, nk), n_k)
        del sgk
        from flash_attention_amd.distributed import gather_output

        def exchanges():
            for s_ in range(slabs):
                gather_output(sg.local[s_], out=sg.fulls[0][s_], mode=mode)
        barrier()
        t1 = time.perf_counter()
        for _ in range(3):
            exchanges()
        sync()
        x_ms = (time.perf_counter() - t1) / 3 * 1e3
    anchor_ms, ga = kernel_ms, gs
    if world > 1:      # the launch a one-GPU run of this config makes: G_total / slabs groups (or this rank's whole shard if that is smaller)
        ga = min(G, Gc // slabs)
        qa = qkv[:, :ga].contiguous()
        anchor_ms = timed(lambda: fa.flash_attention_qkv_packed(qa, nq, nk), max(3, min(8, steps)))
        del qa
    red = torch.tensor([elapsed, kernel_ms, x_ms, anchor_ms], dtype=torch.float64, device=ctx["reduce_dev"])
    if ctx["dist_on"]:
        dist.all_reduce(red, op=dist.ReduceOp.MAX)
    elapsed, kernel_ms, x_ms, anchor_ms = red.tolist()
    flops_total = 2.0 * Gc * L * L * D
    value = flops_total * steps / elapsed / 1e12
    kernel_only = world * 2.0 * G * L * L * D / (max(kernel_ms, 1e-9) * 1e-3) / 1e12
    anchor = 2.0 * ga * L * L * D / (max(anchor_ms, 1e-9) * 1e-3) / 1e12 if world > 1 else kernel_only
    shard_bytes = G * L * D * qkv.element_size()
    plan = _native.plan(gs, L, D, dtype, nq, nk) if dev.type == "cuda" else None
    rec = {"value": round(value, 2), "unit": "TFLOP/s", "scaling": "strong", "n_gpus": world, "steps": steps,
           "ms_per_step": round(elapsed / steps * 1e3, 4), "kernel_only_value": round(kernel_only, 2),
           "frac": round(value / (world * PEAK_TFLOPS), 4), "ms": round(elapsed / steps * 1e3, 4),
           "kernel": kernel_name(plan).replace("fa_", "").replace("_kernel", "") if plan is not None else "cpu",
           "exchange_only_ms": round(x_ms, 4) if sg.gather else None,
           "achieved_GBps_per_link": round(shard_bytes / (max(x_ms, 1e-9) * 1e-3) / 1e9, 2) if sg.gather and world > 1 else None,
           "expected_link_bound_ms": round(shard_bytes / 76e9 * 1e3, 3) if world > 1 else None,
           "n1_anchor_kernel_tflops": round(anchor, 2),
           "scaling_efficiency": round(value / (world * anchor), 4) if anchor > 0 else None,
           "kernel_scaling_efficiency": round(kernel_only / (world * anchor), 4) if anchor > 0 else None,
           "exchange_selected": mode if sg.gather else None, "exchange_selection": selection if sg.gather else None,
           "groups_total": Gc, "groups_per_gpu": G, "slabs": slabs, "groups_per_launch": gs,
           "shape_per_gpu": [3, G, L, D], "dtype": dname}
    if table is not None:
        rec["gather_modes"] = {m: [round(r["exchange_only_ms"], 4), round(r["ms_per_step"], 4)] for m, r in table.items()}
    del sg, qkv
    return {k: v for k, v in rec.items() if v is not None}      # (a one-GPU record has no exchange fields)


def kernel_name(plan):
    return {0: "fa_fwd_dma_kernel", 1: "fa_fwd_generic_kernel", 2: "fa_fwd_pipe_kernel", 3: "fa_fwd_f32_kernel",
            4: "fa_fwd_pipe16_kernel", 6: "fa_fwd_kh_kernel"}[plan.kernel]


def profiled_traffic(config, plan, kname=None):
    """HBM bytes per launch from the PMC passes of scripts/profile.sh (profiles/hbm_traffic.json), valid only for
    the kernel sources and the kernel it was measured on - otherwise null with the reason."""
    tfile = ROOT / "profiles" / "hbm_traffic.json"
    if not tfile.exists():
        return None, "no profiles/hbm_traffic.json"
    try:
        rec = json.loads(tfile.read_text()).get(config)
    except Exception as e:  # noqa: BLE001
        return None, f"unreadable profiles/hbm_traffic.json: {e}"
    if not rec:
        return None, f"no PMC record for {config}"
    if rec.get("csrc_sha256_16") != csrc_fingerprint(rec.get("kernel")):
        return None, f"PMC record is for kernel sources {rec.get('csrc_sha256_16')}, this build is {csrc_fingerprint(rec.get('kernel'))}: re-run scripts/profile.sh"
    kname = kname or kernel_name(plan)
    names = lambda t: sorted(x.split(" ")[0] for x in t.split(" + "))  # noqa: E731
    if rec.get("kernel") and names(rec["kernel"]) != names(kname):
        return None, f"PMC record is for {rec.get('kernel')}, this run launched {kname}"
    return rec.get("hbm_bytes_per_launch"), f"rocprofv3 PMC passes ({rec.get('source', 'profiles/')}), FETCH_SIZE doubled per the gfx950 rule + WRITE_SIZE"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--ramp-ms", type=float, default=1000.0,
                    help="untimed load before the W warm-up steps.  An idle MI355X needs ~60 ms of load to come within 1.5 %% of its "
                         "sustained clock (scripts/clock_ramp.py: 838 TFLOP/s after 7 ms, 1031 after 60 ms, 1045 after 3 s), and about "
                         "0.5 s after the load starts the device stalls ONCE for ~50 ms (seen on every box: one 20-launch group at "
                         "~100 TFLOP/s, none later in 8 s): a 20-step timed region that lands on it reads 850 instead of 1030, so the "
                         "ramp runs well past it")
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="N=1: skip the cfg3 / cfg4 / cfg5-shard records")
    ap.add_argument("--no-gather", action="store_true", help="N>1: skip the gather of O (kernel only)")
    ap.add_argument("--gather", default="auto", choices=("auto", "allgather", "direct", "root"),
                    help="N>1: how O is moved - auto (default): RCCL all-gather, direct point-to-point gather and gather to rank 0 are timed on "
                         "the real shard before the ramp, every rank takes the all-reduced maximum and the fastest runs the headline; or force one")
    ap.add_argument("--slabs", type=int, default=8, help="cfg5: slabs per rank (gather of slab s under the kernel of slab s+1)")
    ap.add_argument("--no-strong", action="store_true", help="N>1: skip the BASELINE configs[4] strong-scaling record (secondary.cfg5_strong) after the weak headline")
    ap.add_argument("--backward", action="store_true", help="N=1: time fa_bwd (dQ + dK/dV kernels) of the config instead of the forward")
    ap.add_argument("--no-probes", action="store_true",
                    help="skip the MFMA-peak probes, the clock sampler and the per-step event pass after the timed region: the LAST --steps "
                         "dispatches of the kernel are then the timed region (scripts/profile.sh computes its kernel statistics over them)")
    ap.add_argument("--print-csrc-fingerprint", nargs="?", const="", default=None, metavar="KERNELS",
                    help="print the sha256 of the kernel sources (of the named kernels' sources only, when given) and exit")
    args = ap.parse_args()
    if args.print_csrc_fingerprint is not None:
        print(csrc_fingerprint(args.print_csrc_fingerprint or None))
        return 0

    gather_requested = args.gather
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args)          # (nothing above has imported torch.cuda or made a HIP call)

    keep_stdout_for_the_json_line()
    smi_mhz = smi_clock_mhz(int(os.environ.get("LOCAL_RANK", "0"))) if int(os.environ.get("RANK", "0")) == 0 else None   # (before the first HIP call)
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # FA_BENCH_FORCE_DIST=1: run the N > 1 control flow - process group on RCCL, gathers of O, per-launch events, agreed ramp count, the
    # exchange comparison - at world size 1 (tests/test_gpu_distributed.py: the first multi-GPU driver run must not be the first RCCL run
    # of this code).  Its `value` is a one-GPU number with a one-rank "exchange" in it.
    dist_on = world > 1 or os.environ.get("FA_BENCH_FORCE_DIST") == "1"
    if dist_on and world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            with socket.socket() as s_:
                s_.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(s_.getsockname()[1])
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # FA_BENCH_REHEARSE=1: developer rehearsal of the N>1 control flow on a ONE-GPU box (all ranks on
    # cuda:0, gloo instead of RCCL).  Never set by the driver; the numbers it prints mean nothing.
    # FA_BENCH_REHEARSE=cpu: the same control flow with NO GPU at all - CPU tensors of a tiny fp32 config through the op's own CPU
    # branch (the reference's device dispatch, fa.py:47-53), gloo exchanges, wall-clock "events": what tests/test_distributed.py
    # runs at world size 2 to check the N > 1 launch path and the JSON line's schema.  Its numbers mean nothing either.
    rehearse_cpu = os.environ.get("FA_BENCH_REHEARSE") == "cpu"
    rehearse = os.environ.get("FA_BENCH_REHEARSE") == "1" or rehearse_cpu
    if rehearse:
        local_rank = 0
    if rehearse_cpu:
        CONFIGS["tiny"] = (4, 64, 16, "f32", (2, 2))
        CONFIGS["tiny_strong"] = (max(16, 2 * world * args.slabs), 64, 16, "f32", (2, 2))
        STRONG.add("tiny_strong")
        args.config = "tiny_strong" if args.config in STRONG else "tiny"
        args.no_probes = args.no_secondary = args.no_cpu_baseline = True
        args.ramp_ms = min(args.ramp_ms, 20.0)
        dev = torch.device("cpu")

        class Ev:      # wall-clock stand-in for a HIP event
            def record(self):
                self.t = time.perf_counter()

            def elapsed_time(self, other):
                return (other.t - self.t) * 1e3

        def sync():
            return None
    else:
        if not torch.cuda.is_available():
            raise SystemExit(f"bench.py needs a GPU (the HIP path has no fallback) [rank {rank} of {world}]")
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)

        def Ev():
            return torch.cuda.Event(enable_timing=True)

        sync = torch.cuda.synchronize
    if dist_on:
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            import datetime
            # (a collective that never completes is aborted by the watchdog after 3 minutes instead of the default 10: the exchange
            # paths have only ever run at world size 1 on hardware)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=datetime.timedelta(seconds=180))

    import flash_attention_amd as fa
    from flash_attention_amd._impl import _native
    from flash_attention_amd.distributed import SlabGather, gather_output

    assert _native.library_path() is not None
    DT = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}
    Gc, L, D, dname, (nq, nk) = CONFIGS[args.config]
    strong = args.config in STRONG
    if strong and Gc % (world * args.slabs) != 0:
        raise SystemExit(f"{args.config}: {Gc} groups do not split over {world} ranks x {args.slabs} slabs")
    G = Gc // world if strong else Gc              # groups on this rank
    dtype = DT[dname]
    gen = torch.Generator(device=dev).manual_seed(rank)
    qkv = torch.randn(3, G, L, D, device=dev, generator=gen).to(dtype)
    flops_step = 2.0 * G * L * L * D               # per GPU per step
    g_launch = G // args.slabs if strong else G    # groups per kernel launch
    plan = _native.plan(g_launch, L, D, dtype, nq, nk)

    if dist_on and (args.backward or args.config in ("cfg4_attn", "cfg2_sdpa")):
        raise SystemExit("--backward / cfg4_attn / cfg2_sdpa are single-GPU records")
    op_fn, op_flops, op_kernel, op_desc = None, None, None, None
    if not dist_on and not strong and (args.backward or args.config in ("cfg4_attn", "cfg2_sdpa")):
        del qkv
        op_fn, op_flops, op_kernel, op_desc, plan_ = make_op(args.config, dev, seed=rank, backward=args.backward)
        plan = plan_ or plan
        flops_step = op_flops

    do_gather = dist_on and not args.no_gather
    use_gloo = rehearse and not rehearse_cpu      # (one-GPU rehearsal: no exchange at all; the CPU rehearsal exchanges CPU tensors over gloo)
    launches_per_step = args.slabs if strong else 1

    # ---- N > 1: which way to move O?  Measured HERE, before the ramp, on the real shard with the real kernel under it - the driver gets one
    # 8-GPU run, and its headline must already run on the exchange this node is fastest with ("prefer direct / grouped
    # point-to-point and measure").  Every rank runs the same sequence and takes the all-reduced maximum, so all agree on the minimum.
    exchange_table, exchange_selection = None, "forced (--gather)"
    if args.gather == "auto":
        exchange_selection = "auto"
        if do_gather and not use_gloo:
            from flash_attention_amd.distributed import select_gather_mode

            def give_up():      # a rank stuck in a collective leaves no headline to print: say so, and fail visibly
                if rank == 0:
                    emit_once({"metric": "fwd_attention_tflops_causal_L4096_D64", "value": None, "n_gpus": world,
                               "error": "the exchange selection did not finish within 180 s (a rank left a collective?)"})
                sys.stdout.flush()
                os._exit(3)

            watchdog = threading.Timer(180.0, give_up)
            watchdog.daemon = True
            watchdog.start()
            try:
                q_sel = qkv[:, :g_launch].contiguous() if strong else qkv
                step_sel = (lambda: fa.flash_attention_qkv_packed(q_sel, nq, nk)) if op_fn is None else op_fn
                # ("root" is timed for information only: it completes rank 0's buffer alone, so it never runs the headline unless forced)
                args.gather, exchange_table = select_gather_mode(step_sel(), step=step_sel, reps=max(2, min(args.steps, 5)), also_time=("root",))
                exchange_selection = "auto: fastest kernel + exchange step of all-gather / direct (every rank gets all of O), timed before the ramp (maximum over ranks)"
            except Exception as e:  # noqa: BLE001  (the selection must never cost the headline)
                args.gather = "allgather"
                exchange_selection = f"auto FAILED ({type(e).__name__}: {e})"[:200] + " -> all-gather"
            finally:
                watchdog.cancel()
        else:
            args.gather = "allgather"
    kernel_events = []                             # (start, end) per launch inside the timed region (N > 1)
    record = [False]

    if strong:
        sg = SlabGather(G, L, D, dtype, dev, slabs=args.slabs, mode=args.gather, gather=do_gather and not use_gloo,
                        gather_at_world1=os.environ.get("FA_BENCH_FORCE_DIST") == "1")

        def step(i):
            if record[0]:
                e0, e1 = Ev(), Ev()
                e0.record()
            out = sg.run(qkv, nq, nk, wait=False)      # its last exchanges run under the next step's first kernels
            if record[0]:
                e1.record()
                kernel_events.append((e0, e1))
            return out

        def drain():
            sg.wait()
    else:
        gathered = [torch.empty(world * G, L, D, dtype=dtype, device=dev) for _ in range(2)] if do_gather and not use_gloo else None
        pending = [None, None]

        def step(i):
            b = i & 1
            if gathered is not None and pending[b] is not None:
                pending[b][1].wait()           # buffer b is free again (gather of step i-2 done)
            if record[0]:
                e0, e1 = Ev(), Ev()
                e0.record()
            out = op_fn() if op_fn is not None else fa.flash_attention_qkv_packed(qkv, nq, nk)
            if record[0]:
                e1.record()
                kernel_events.append((e0, e1))
            if gathered is not None:
                _, work = gather_output(out, out=gathered[b], async_op=True, mode=args.gather)
                pending[b] = (out, work)       # keep `out` alive until its gather completes
            return out

        def drain():
            for b in range(2):
                if pending[b] is not None:
                    pending[b][1].wait()
                    pending[b] = None

    ramp_info = None
    step(0)                      # first call pays one-off host costs (library load, op registration)
    drain()
    sync()
    # untimed ramp: bring the chip to its sustained clock.  Every rank must issue the SAME number of steps (each step
    # carries a collective), so with N > 1 the count is agreed first: 3 steps are timed, the slowest rank's time decides.
    if dist_on:
        t_est = time.perf_counter()
        for i in range(3):
            step(i)
        drain()
        sync()
        est = torch.tensor([(time.perf_counter() - t_est) / 3 * 1e3], dtype=torch.float64, device="cpu" if use_gloo else dev)
        dist.all_reduce(est, op=dist.ReduceOp.MAX)
        n_ramp = int(min(2000, max(1, args.ramp_ms / max(est.item(), 1e-3))))
        for i in range(n_ramp):
            step(i)
            if (i + 1) % 16 == 0:
                sync()
    else:
        # groups of 16 steps between events: the slowest group stays visible in the JSON line (`ramp`): the idle -> sustained
        # clock transition of the device, including the one-time stall DESIGN.md 3.8 describes, happens here and not in the timed region
        t_ramp = time.perf_counter()
        i_ramp = 0
        g_ev = [(0.0, Ev())]
        g_ev[0][1].record()
        while (time.perf_counter() - t_ramp) * 1e3 < args.ramp_ms:
            step(i_ramp)
            i_ramp += 1
            if i_ramp % 16 == 0:
                e = Ev()
                e.record()
                sync()
                g_ev.append(((time.perf_counter() - t_ramp) * 1e3, e))
        if len(g_ev) > 1:
            durs = [(g_ev[i][1].elapsed_time(g_ev[i + 1][1]) / 16, g_ev[i + 1][0]) for i in range(len(g_ev) - 1)]
            slow = max(durs)
            srt = sorted(d for d, _ in durs)
            ramp_info = {"ms": args.ramp_ms, "groups_of_16_steps": len(durs), "median_group_ms_per_step": round(srt[len(srt) // 2], 4),
                         "slowest_group_ms_per_step": round(slow[0], 4), "slowest_group_ended_at_ms": round(slow[1], 1),
                         "first_group_ms_per_step": round(durs[0][0], 4)}
    drain()
    sync()
    for i in range(args.warmup):
        step(i)
    drain()
    sync()
    if dist_on:
        dist.barrier()
    sync()

    record[0] = dist_on and not strong          # per-launch events only where a collective shares the stream
    ev0, ev1 = Ev(), Ev()
    t0 = time.perf_counter()
    ev0.record()
    for i in range(args.steps):
        step(i)
    ev1.record()
    drain()
    sync()
    if dist_on:
        dist.barrier()
    sync()
    elapsed = time.perf_counter() - t0
    record[0] = False
    if kernel_events:      # N > 1, weak: kernel-only = events right around each launch (the waits on gathers sit before them)
        kernel_ms = sum(a.elapsed_time(b) for a, b in kernel_events) / len(kernel_events)
        kernel_note = "HIP events around each launch (exchange stalls excluded)"
    else:
        kernel_ms = ev0.elapsed_time(ev1) / (args.steps * launches_per_step)   # avg launch duration incl. launch gaps (HIP events)
        kernel_note = "HIP events over the timed region / launches" + (" (includes stalls on the slab gathers)" if strong and do_gather else "")
    if strong and do_gather:
        # kernel-only for the slab pipeline: the same slabs, no gather, a few untimed-region repeats
        sgk = SlabGather(G, L, D, dtype, dev, slabs=args.slabs, gather=False)
        n_k = max(3, args.steps // 4)
        sgk.run(qkv, nq, nk)
        sync()
        ek0, ek1 = Ev(), Ev()
        ek0.record()
        for _ in range(n_k):
            sgk.run(qkv, nq, nk)
        ek1.record()
        sync()
        kernel_ms = ek0.elapsed_time(ek1) / n_k / args.slabs
        kernel_note = "HIP events over a gather-free repeat of the same slab launches"

    t = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device=dev)
    if dist_on and not use_gloo:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elif dist_on:
        tc = t.cpu()
        dist.all_reduce(tc, op=dist.ReduceOp.MAX)
        t = tc
    elapsed, kernel_ms = t[0].item(), t[1].item()

    ctx = {"torch": torch, "dist": dist, "dev": dev, "world": world, "rank": rank, "Ev": Ev, "sync": sync, "DT": DT, "dist_on": dist_on,
           "exchange": not args.no_gather and not use_gloo, "force_dist": os.environ.get("FA_BENCH_FORCE_DIST") == "1",
           "reduce_dev": "cpu" if (use_gloo or rehearse_cpu) else dev}
    # ---- N > 1, the driver's plain `bench.py --gpus N`: BASELINE configs[4] (strong: 2048 groups in total, SlabGather, its own exchange
    # selection) right after the weak headline, in the same launch and process group -> secondary.cfg5_strong.  Every rank runs it.
    cfg5_strong = None
    if dist_on and not strong and op_fn is None and not args.no_strong:
        drain()
        name5 = "tiny_strong" if rehearse_cpu else "cfg5"

        def strong_gave_up():      # a stuck collective must not cost the weak headline, which is already measured
            if rank == 0 and line_box[0] is not None:
                line_box[0].setdefault("secondary", {})["cfg5_strong"] = {"error": "did not finish within 300 s (a rank left a collective?)"}
                emit_once(line_box[0])
            sys.stdout.flush()
            os._exit(4)

        strong_watchdog = threading.Timer(300.0, strong_gave_up)
        strong_watchdog.daemon = True
    else:
        strong_watchdog = None
    line_box = [None]

    rc = 0
    line = None
    shard_bytes = G * L * D * qkv.element_size() if dist_on else 0
    median_ms = min_ms = None
    if not dist_on and not strong and not args.no_probes:
        median_ms, min_ms = per_step_ms(lambda: step(0), args.steps)     # median and min, a pass of its own
    if rank == 0:
        lib_ = _native.load_native_library()
        measured_peak = measured_peak_random = clock_mhz = None
        if not args.no_probes:
          with torch.cuda.device(dev):
            mp = lib_.fa_probe_mfma_peak(_native.FA_DTYPE[dtype], 4000, 0)
            mpr = lib_.fa_probe_mfma_peak(_native.FA_DTYPE[dtype], 40000, 1)   # ~30 ms per repetition: clock settles
          measured_peak = round(mp, 1) if mp > 0 else None
          measured_peak_random = round(mpr, 1) if mpr > 0 else None
        # which core clock does the device hold under this load?  (untimed repeat of the kernel with a one-wave clock
        # sampler on a side stream: the kernels are power-bound; nominal peak assumes 2400 MHz)
        if not args.no_probes:
            if op_fn is not None:
                probe_fn = op_fn
            else:
                qk = qkv[:, :g_launch].contiguous() if strong else qkv
                probe_fn = lambda: fa.flash_attention_qkv_packed(qk, nq, nk)   # noqa: E731  (kernel only: no collective on one rank)
            side = torch.cuda.Stream(device=dev)
            buf = torch.zeros(2, dtype=torch.int64, device=dev)
            n_rep = max(8 + 2, int(60.0 / max(kernel_ms, 1e-3)))                   # >= 60 ms of load
            for i in range(n_rep):
                probe_fn()
                if i == 8:                                                          # the load is running: start sampling
                    lib_.fa_probe_clock(buf.data_ptr(), 15000, side.cuda_stream)   # ~30-40 ms
            sync()
            cyc, ticks = (int(x) for x in buf.tolist())
            clock_mhz = round(cyc / ticks * 100.0) if ticks > 0 else None

        total_tflops = world * flops_step * args.steps / elapsed / 1e12
        flops_launch = op_flops if op_fn is not None else 2.0 * g_launch * L * L * D
        achieved = flops_launch / (kernel_ms * 1e-3) / 1e12
        kname = op_kernel if op_fn is not None else kernel_name(plan)
        tkey = "cfg5_slab" if strong else args.config + ("_bwd" if args.backward else "")
        traffic, traffic_note = profiled_traffic(tkey, plan, kname)
        metric = "fwd_attention_tflops_causal_L4096_D64" if args.config.startswith("cfg2") else f"fwd_attention_tflops_causal_L{L}_D{D}_{args.config}"
        if args.backward:
            metric = f"bwd_attention_tflops_causal_L{L}_D{D}_{args.config}"
        exchange = "none"
        if do_gather:
            exchange = {"allgather": "rccl all-gather of O", "direct": "direct point-to-point gather of O (rccl send/recv)",
                        "root": "gather of O to rank 0 (rccl send/recv)"}[args.gather]
            exchange += f", {args.slabs} slabs per rank, gather of slab s under the kernel of slab s+1" if strong else ", double-buffered: gather of step s under the kernel of step s+1"
        line = {
            "metric": metric,
            "value": round(total_tflops, 2), "unit": "TFLOP/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": dname,
            "data": "synthetic",
            "median_ms": round(median_ms, 4) if median_ms is not None else None,
            "min_ms": round(min_ms, 4) if min_ms is not None else None,
            "ramp": ramp_info,
            "config": {"workload": op_desc if op_fn is not None else
                                   f"{args.config}: packed qkv [3,{G},{L},{D}] {dname} per GPU"
                                   + (f" ({Gc} groups in total over {world} GPUs)" if strong else "")
                                   + f", causal, num_chunks=({nq},{nk}), flash_attention_qkv_packed forward",
                       "groups_per_gpu": G, "seq_len": L, "head_dim": D,
                       "tile": f"{plan.block_q} query rows x {plan.block_kv} keys per workgroup step ({plan.waves} waves, {plan.qb} query blocks per wave)",
                       "plan_source": {0: "built-in rules", 1: "measured tuning table", 2: "debug override"}.get(plan.source),
                       "exchange": exchange,
                       "exchange_selected": args.gather if do_gather else None,
                       "exchange_selection": exchange_selection if do_gather else None,
                       "flops_per_step_per_gpu": flops_step,
                       "device": {"name": "cpu (FA_BENCH_REHEARSE=cpu: numbers are meaningless)"} if rehearse_cpu else
                                 {"name": torch.cuda.get_device_name(dev),
                                  "compute_units": torch.cuda.get_device_properties(dev).multi_processor_count,
                                  "clock_mhz": device_clock_mhz(torch, dev, smi_mhz)}},
            "exchange_bytes_per_rank": {"sent": shard_bytes if do_gather else 0,
                                        "received": ((world - 1) * shard_bytes if args.gather != "root" or rank == 0 else 0) if do_gather else 0} if dist_on else None,
            # what the exchange alone needs when every GPU pair has its own xGMI link and each link carries ONE shard per direction
            # (~76 GB/s per direction): the floor of a step's gather - read `value` against max(kernel, this)
            "expected_link_bound_ms": round(shard_bytes / 76e9 * 1e3, 3) if do_gather else None,
            "pct_mfma_peak": round(100.0 * total_tflops / (world * PEAK_TFLOPS), 2),
            "kernel_only_value": round(world * achieved, 2),
            # value / (N x the one-GPU kernel-only rate of this launch shape, measured in THIS run on every rank: events around the launches,
            # slowest rank): how much of N kernels' worth of work the job delivers once the exchange is in (1.0 = the exchange is free)
            "n1_anchor_kernel_tflops": round(achieved, 2),
            "scaling_efficiency": round(total_tflops / (world * achieved), 4) if achieved > 0 else None,
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK_TFLOPS, 4), "traffic": traffic,
                         "traffic_note": traffic_note if traffic is None else "rocprofv3 PMC passes (profiles/), FETCH_SIZE doubled per the gfx950 rule + WRITE_SIZE",
                         "hbm_gbps": round(traffic / (kernel_ms * 1e-3) / 1e9, 1) if traffic else None,
                         "alg_gbps": round(4 * g_launch * L * D * 2 / (kernel_ms * 1e-3) / 1e9, 1),
                         "shader_clock_mhz_under_load": clock_mhz,
                         "frac_at_that_clock": round(achieved / (PEAK_TFLOPS * clock_mhz / 2400.0), 4) if clock_mhz else None,
                         "measured_mfma_peak": measured_peak,
                         "measured_mfma_peak_random_operands": measured_peak_random,
                         "frac_of_measured_peak_random_operands": round(achieved / measured_peak_random, 4) if measured_peak_random else None,
                         # the same two from the per-step MEDIAN (a pass of its own, one event pair per step): the figure to quote when the two differ
                         "achieved_median": round(flops_launch / (median_ms * 1e-3) / 1e12, 2) if median_ms else None,
                         "frac_median": round(flops_launch / (median_ms * 1e-3) / 1e12 / PEAK_TFLOPS, 4) if median_ms else None,
                         "kernel": kname, "avg_launch_ms": round(kernel_ms, 4), "avg_launch_note": kernel_note,
                         "flops_per_launch": flops_launch,
                         "algorithmic_bytes": 4 * g_launch * L * D * 2},
        }
        if not dist_on and not args.no_secondary and not strong:
            # BASELINE configs LAST (the driver keeps the tail of stdout), every record compact; the same numbers once more, shorter
            # still, under roofline.other_configs - a key the driver stores verbatim
            sec = {}
            for name in ("few_groups", "small_grid", "l1024", "nb_65536", "cfg2_sdpa", "cfg2_f16", "cfg4_attn", "cfg5_shard", "cfg3"):
                if name != args.config:
                    try:
                        sec[name] = secondary_attention(name, dev)
                    except Exception as e:  # noqa: BLE001  (a secondary record must never cost the headline)
                        sec[name] = {"error": f"{type(e).__name__}: {e}"[:200]}
            # the one-GPU anchor of BASELINE configs[4]'s 1/2/4/8 strong-scaling curve: all 2 048 groups on this GPU, the same slab launches
            try:
                torch.cuda.empty_cache()
                sec["cfg5_full"] = strong_record("cfg5", ctx, steps=5, warmup=1, slabs=args.slabs, gather="auto")
                torch.cuda.empty_cache()
            except Exception as e:  # noqa: BLE001
                sec["cfg5_full"] = {"error": f"{type(e).__name__}: {e}"[:200]}
            sec["cfg3"] = sec.pop("cfg3")          # (BASELINE configs stay last among the forward records)
            try:
                sec["cfg2_bwd"] = secondary_attention("cfg2", dev, steps=10, warmup=3, backward=True)
            except Exception as e:  # noqa: BLE001
                sec["cfg2_bwd"] = {"error": f"{type(e).__name__}: {e}"[:200]}
            try:
                sec["bwd_small_grids"] = secondary_backward_small_grids(dev)
            except Exception as e:  # noqa: BLE001
                sec["bwd_small_grids"] = {"error": f"{type(e).__name__}: {e}"[:200]}
            try:
                sec["cfg4_gpt2_fwd"] = secondary_gpt2_forward(dev)
            except Exception as e:  # noqa: BLE001
                sec["cfg4_gpt2_fwd"] = {"error": f"{type(e).__name__}: {e}"[:200]}
            line["secondary"] = sec
            line["roofline"]["other_configs"] = {
                k: ({"tflops": v["value"], "frac": v["frac"], "kernel": v["kernel"], "ms": v["ms"]} if "frac" in v else
                    {s_: [r_["ms"], r_["plain_grid_ms"]] for s_, r_ in v.items()} if k == "bwd_small_grids" and "error" not in v else
                    {"step_ms": v.get("value"), "stock_ms": v.get("stock_sdpa_ms"), "graph_ms": v.get("hip_graph_ms"), "attn_frac": v.get("attn_frac")} if "error" not in v else v)
                for k, v in sec.items()}
        if not args.no_cpu_baseline and not dist_on:
            line["cpu_baseline"] = cpu_baseline(4096, 64) if args.config.startswith("cfg2") else cpu_baseline(min(L, 4096), D)
        else:
            line["cpu_baseline"] = None
        # the driver keeps an 8 KB tail of stdout: the line must fit (shed detail, never the contract keys)
        if len(json.dumps(line)) > 7600 and isinstance(line.get("secondary"), dict):
            for v in line["secondary"].values():
                if isinstance(v, dict):
                    for k2 in ("median_ms", "min_ms", "alg_bytes", "shape", "tile", "shape_per_gpu", "unit", "steps", "ms"):
                        v.pop(k2, None)
        if len(json.dumps(line)) > 7600:
            line.pop("ramp", None)
    line_box[0] = line
    if strong_watchdog is not None:
        strong_watchdog.start()
        try:
            cfg5_strong = strong_record(name5, ctx, steps=max(3, min(args.steps, 10)), warmup=2, slabs=args.slabs, gather=gather_requested)
        except Exception as e:  # noqa: BLE001  (deterministic host-side failures - a group count that does not divide - hit every rank alike)
            cfg5_strong = {"error": f"{type(e).__name__}: {e}"[:200]}
        finally:
            strong_watchdog.cancel()
    # N > 1: the exchange table of the selection above (per mode: the exchange alone - what the links deliver - and kernel + exchange per
    # step, overlapped the way the headline overlaps them; maxima over ranks)
    gather_modes = None
    if exchange_table is not None:
        gather_modes = {}
        for mode, rec_ in exchange_table.items():
            x_ms, s_ms = rec_["exchange_only_ms"], rec_["ms_per_step"]
            gather_modes[mode] = {
                "exchange_only_ms": round(x_ms, 4), "ms_per_step": round(s_ms, 4),
                "value_tflops": round(world * 2.0 * g_launch * L * L * D / (max(s_ms, 1e-9) * 1e-3) / 1e12, 2),
                "received_bytes_busiest_rank": (world - 1) * (shard_bytes // launches_per_step),
                # one shard per link and direction: what one xGMI link carried (0 at world size 1: nothing crosses a link)
                "achieved_GBps_per_link": round((shard_bytes // launches_per_step) / (max(x_ms, 1e-9) * 1e-3) / 1e9, 2) if world > 1 else 0.0,
                "selected": mode == args.gather, "eligible": bool(rec_.get("eligible", True)),   # ("root" completes rank 0's buffer only: timed, never auto-selected)
            }
    if rank == 0:
        if gather_modes is not None:
            line.setdefault("secondary", {})["gather_modes"] = gather_modes
        if cfg5_strong is not None:
            line.setdefault("secondary", {})["cfg5_strong"] = cfg5_strong
        emit_once(line)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()
    return rc


if __name__ == "__main__":
    sys.exit(main())
