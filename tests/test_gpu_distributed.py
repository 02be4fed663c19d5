"""GPU tests of the multi-GPU path on the ONE GPU of a test box: the RCCL call path of `gather_output`
(`torch.distributed` backend "nccl" = RCCL) executes with world_size 1, and the BASELINE config 5 per-GPU shard
shape runs through the sharded op.  (Real N > 1 runs belong to the driver's 8-GPU node; the N > 1 control flow is
covered on CPU by tests/test_distributed.py with gloo.)"""

import os
import socket

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.fixture
def rccl_world1():
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        yield dev
    finally:
        dist.destroy_process_group()


def test_gather_output_runs_on_rccl_world1(rccl_world1):
    from flash_attention_amd.distributed import (SlabGather, flash_attention_qkv_packed_sharded, gather_output,
                                                 shard_range)
    from oracle import dense_causal_attention, frozen_randn

    dev = rccl_world1
    G, L, D = 4, 256, 64
    assert shard_range(G, 0, 1) == (0, G)
    qkv = frozen_randn(5, (3, G, L, D)).to(torch.bfloat16)
    ref = dense_causal_attention(qkv, compute_dtype=torch.float64)
    local = flash_attention_qkv_packed_sharded(qkv.to(dev), 2, 2, gather=False)
    for mode in ("allgather", "direct", "root"):
        out = gather_output(local, mode=mode)                    # the collective / p2p API path on RCCL
        assert out.shape == (G, L, D) and torch.equal(out, local), mode
        out2, work = gather_output(local, mode=mode, async_op=True)
        work.wait()
        torch.cuda.synchronize()
        assert torch.equal(out2, local), mode
    err = (out.cpu().double() - ref).abs()
    assert bool((err <= 1e-2 + 1e-2 * ref.abs()).all())
    # slab-overlapped variant (cfg5 mode of bench.py): gather of slab s under the kernel of slab s + 1
    sg = SlabGather(G, L, D, torch.bfloat16, dev, slabs=2)
    full = sg.run(qkv.to(dev), 2, 2)
    torch.cuda.synchronize()
    assert torch.equal(full, local)


@pytest.mark.parametrize("extra", [[], ["--gather", "direct"], ["--config", "cfg5", "--slabs", "8"]],
                         ids=["weak_allgather_with_mode_comparison", "weak_direct", "strong_cfg5_slab_gather"])
def test_bench_multi_gpu_control_flow_on_rccl_at_world_size_1(extra):
    """bench.py's N > 1 flows end to end ON RCCL - process group, agreed ramp count, double-buffered gather of O under the next
    kernel (weak) / SlabGather.run(wait=False) (strong, BASELINE config 5), per-launch events, max-over-ranks reduction, the
    gather-mode comparison, rank 0's one JSON line - forced at world size 1 (FA_BENCH_FORCE_DIST=1), so that the first 8-GPU
    driver run is not also the first RCCL run of this code (round-3 verdict, item 8).  The numbers are one-GPU numbers."""
    import json
    import pathlib
    import subprocess
    import sys

    root = pathlib.Path(__file__).resolve().parent.parent
    env = dict(os.environ, FA_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("MASTER_PORT", None)
    cmd = [sys.executable, str(root / "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--ramp-ms", "30",
           "--no-probes", "--no-cpu-baseline", "--no-secondary"] + extra
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    assert res.stdout.strip().splitlines() == lines, "stdout carries the JSON line and nothing else (RCCL's banner, library chatter: stderr)"
    assert len(lines[0]) < 8000, "the driver keeps an 8 KB tail: the line must fit"
    rec = json.loads(lines[0])
    strong = "cfg5" in extra
    assert rec["n_gpus"] == 1 and rec["scaling"] == ("strong" if strong else "weak") and rec["value"] > 0
    assert rec["kernel_only_value"] >= rec["value"] * 0.5
    assert rec["exchange_bytes_per_rank"]["sent"] > 0 and rec["expected_link_bound_ms"] > 0
    assert "rccl" in rec["config"]["exchange"] or "point-to-point" in rec["config"]["exchange"]
    if "--gather" not in extra:       # auto: the exchanges timed before the ramp, all ranks agreed on the fastest FULL gather ("root": information only)
        modes = rec["secondary"]["gather_modes"]
        assert set(modes) == {"allgather", "direct", "root"}, modes
        for m in modes.values():
            assert m["ms_per_step"] > 0 and m["exchange_only_ms"] >= 0
        assert [k for k, m in modes.items() if m["selected"]] == [rec["config"]["exchange_selected"]]
        assert rec["config"]["exchange_selected"] in ("allgather", "direct") and not modes["root"]["eligible"]
        assert rec["config"]["exchange_selection"].startswith("auto: fastest")
    else:
        assert rec["config"]["exchange_selected"] == "direct"
    assert 0.5 < rec["scaling_efficiency"] <= 1.05
    if not strong:
        # round-5 verdict, item 1: the plain N > 1 command carries BASELINE configs[4] too - all 2 048 groups (here on ONE GPU), SlabGather with
        # its exchange calls on RCCL, its own selection, the efficiency figures
        s5 = rec["secondary"]["cfg5_strong"]
        assert "error" not in s5, s5
        assert s5["groups_total"] == 2048 and s5["shape_per_gpu"] == [3, 2048, 8192, 64] and s5["slabs"] == 8 and s5["scaling"] == "strong"
        assert s5["value"] > 500 and s5["kernel_only_value"] >= s5["value"] * 0.9 and s5["exchange_only_ms"] > 0
        assert s5["exchange_selected"] == ("direct" if "--gather" in extra else s5["exchange_selected"]) and s5["exchange_selected"] in ("allgather", "direct")
        assert 0.5 < s5["scaling_efficiency"] <= 1.05 and s5["kernel"] in ("fwd_pipe", "fwd_pipe16", "fwd_dma")


def test_bench_two_rank_processes_on_the_one_gpu_with_the_real_configs():
    """`FA_BENCH_REHEARSE=1 python bench.py --gpus 2`: two rank PROCESSES started by bench.py itself (one torch.distributed.run child), both on
    cuda:0, gloo for the barriers and reductions, no exchange - the multi-process control flow of the driver's N > 1 command on the REAL configs:
    the cfg2 weak headline (128 groups per rank), then BASELINE config 5 split over two ranks (1 024 groups each, 8 slabs of 128), the
    max-over-ranks reductions, rank 0's one JSON line.  (RCCL itself needs one GPU per rank: its call path runs at world size 1 above.)"""
    import json
    import pathlib
    import subprocess
    import sys

    root = pathlib.Path(__file__).resolve().parent.parent
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(FA_BENCH_REHEARSE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--ramp-ms", "30", "--no-probes"],
                         env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    assert res.stdout.strip().splitlines() == lines, "stdout carries the JSON line and nothing else (RCCL's banner, library chatter: stderr)"
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["scaling"] == "weak" and rec["value"] > 200 and rec["config"]["groups_per_gpu"] == 128
    s5 = rec["secondary"]["cfg5_strong"]
    assert "error" not in s5, s5
    assert s5["n_gpus"] == 2 and s5["groups_total"] == 2048 and s5["groups_per_gpu"] == 1024 and s5["groups_per_launch"] == 128
    assert s5["value"] > 200 and s5["kernel_only_value"] > 0 and s5["scaling_efficiency"] > 0 and s5["n1_anchor_kernel_tflops"] > 200
