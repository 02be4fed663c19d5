"""world_size-2 (and 3) gloo tests of the sharded path on CPU (runs without a GPU): the all-gather, the direct
point-to-point gather, the gather to rank 0 and the slab-overlapped variant, plus bench.py's self-launch."""

import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, G, L, D, q):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "flash-attention-ipu_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    torch.set_num_threads(1)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from flash_attention_amd.distributed import flash_attention_qkv_packed_sharded, gather_output, shard_range
        from oracle import dense_causal_attention, frozen_randn

        qkv = frozen_randn(11, (3, G, L, D))
        lo, hi = shard_range(G, rank, world)
        full = flash_attention_qkv_packed_sharded(qkv[:, lo:hi].contiguous(), 2, 2)
        ref = dense_causal_attention(qkv)
        ok = full.shape == (G, L, D) and torch.allclose(full, ref, atol=1e-6)
        local = flash_attention_qkv_packed_sharded(qkv[:, lo:hi].contiguous(), 2, 2, gather=False)
        ok = ok and torch.equal(local, full[lo:hi])
        out, work = gather_output(local, async_op=True)
        work.wait()
        ok = ok and torch.equal(out, full)
        for mode in ("direct", "root"):
            got = flash_attention_qkv_packed_sharded(qkv[:, lo:hi].contiguous(), 2, 2, mode=mode)
            if mode == "direct" or rank == 0:
                ok = ok and torch.equal(got, full)
            else:
                ok = ok and torch.equal(got[lo:hi], full[lo:hi])
            out, work = gather_output(local, async_op=True, mode=mode)
            work.wait()
            ok = ok and (torch.equal(out, full) if (mode == "direct" or rank == 0) else torch.equal(out[lo:hi], local))
        # slab-overlapped gather: slab-cyclic ownership, the gathered buffer is in global group order
        from flash_attention_amd.distributed import SlabGather

        for mode in ("allgather", "direct"):
            sg = SlabGather(G // world, L, D, torch.float32, "cpu", slabs=G // world, mode=mode)
            mine = sg.owned_groups(rank)
            first = sg.run(qkv[:, mine].contiguous(), 2, 2)                  # (default: complete on return)
            ok = ok and torch.allclose(first, ref, atol=1e-6)
            got = sg.run(qkv[:, mine].contiguous(), 2, 2, wait=False)      # second call: waits for the first call's exchanges per slab
            sg.wait()
            ok = ok and got.shape == (G, L, D) and torch.allclose(got, ref, atol=1e-6)
            ok = ok and got.data_ptr() != first.data_ptr() and torch.allclose(first, ref, atol=1e-6)   # two result buffers, used alternately
        owners = torch.cat([SlabGather(G // world, L, D, torch.float32, "cpu", slabs=G // world).owned_groups(r) for r in range(world)])
        ok = ok and sorted(owners.tolist()) == list(range(G))
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_forward_gloo(world):
    G, L, D = 6, 32, 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, G, L, D, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(results) == [(r, True) for r in range(world)]


def _select_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from flash_attention_amd.distributed import select_gather_mode

        # a fake clock: rank 0 alone would pick "allgather" (1 / 5 / 9 ms per step), rank 1 alone "direct" (9 / 2 / 3) - the all-reduced
        # maxima are 9 / 5 / 9, so BOTH must come out with "direct" and with the same table
        per_step_ms = [[1.0, 5.0, 9.0], [9.0, 2.0, 3.0]][rank]
        ticks = []
        for ms in per_step_ms:
            ticks += [0.0, 0.5e-3 * 2, 0.0, ms * 1e-3 * 2]     # exchange-only interval, then the overlapped-step interval (reps = 2)
        now, it = [0.0], iter(ticks)

        def clock():
            now[0] += next(it)
            return now[0]

        local = torch.full((2, 4, 8), float(rank))
        calls = [0]

        def step():
            calls[0] += 1
            return local.clone()

        mode, table = select_gather_mode(local, step=step, reps=2, clock=clock, also_time=("root",))
        q.put((rank, mode, {k: (round(v["exchange_only_ms"], 6), round(v["ms_per_step"], 6)) for k, v in table.items()}, calls[0]))
    finally:
        dist.destroy_process_group()


def test_exchange_selection_agrees_across_ranks_on_the_all_reduced_maximum():
    """bench.py --gather auto / distributed.select_gather_mode at world size 2 on gloo: ranks whose own clocks prefer DIFFERENT exchanges
    must leave the selection with the same mode and the same table (every time is the maximum over ranks)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_select_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, m0, t0, c0), (_, m1, t1, c1) = res
    assert m0 == m1 == "direct" and t0 == t1
    assert t0["allgather"][1] == 9.0 and t0["direct"][1] == 5.0 and t0["root"][1] == 9.0       # ("root": timed for information, never selected)
    assert c0 == c1 == 6           # two overlapped steps per mode, every rank the same sequence of collectives


def test_shard_range():
    from flash_attention_amd.distributed import shard_range

    assert [shard_range(128, r, 8) for r in (0, 7)] == [(0, 16), (112, 128)]
    with pytest.raises(ValueError):
        shard_range(10, 0, 4)


def test_bench_self_launches_rank_processes_before_touching_the_gpu():
    """`python bench.py --gpus 2` with no launcher around it: the parent starts 2 fresh rank processes (one
    torch.distributed.run child) and relays their exit code.  On this CPU-only box each rank stops at "needs a GPU" -
    which proves the ranks were started (RANK/WORLD_SIZE set) and that the parent itself never needed the GPU."""
    import pathlib
    import subprocess

    root = pathlib.Path(__file__).resolve().parent.parent
    if torch.cuda.is_available():
        pytest.skip("CPU-only check (on a GPU box the N > 1 control flow is rehearsed with FA_BENCH_REHEARSE=1)")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    res = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert res.returncode != 0
    # (the launcher stops the surviving rank as soon as the first one fails, so at least one of them reports)
    assert "bench.py needs a GPU" in res.stderr and " of 2]" in res.stderr, res.stderr[-2000:]
    assert "but WORLD_SIZE" not in res.stderr


@pytest.mark.parametrize("config,extra", [("cfg2", []), ("cfg2", ["--gather", "direct"]), ("cfg5", ["--slabs", "2"])])
def test_bench_n2_control_flow_and_json_schema_on_cpu(config, extra):
    """`FA_BENCH_REHEARSE=cpu python bench.py --gpus 2`: the WHOLE N > 1 path of the benchmark with no GPU - self-launch of the two
    rank processes, gloo process group, agreed ramp step count, double-buffered exchange under the next step's op (weak) or slab
    pipeline (strong), max-over-ranks timing, rank 0's one JSON line - on CPU tensors of a tiny config through the op's own CPU
    branch.  Checks the launch path and the line's schema (incl. the all-gather / direct / root comparison measured in the same
    launch); the numbers mean nothing."""
    import json
    import pathlib
    import subprocess

    root = pathlib.Path(__file__).resolve().parent.parent
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["FA_BENCH_REHEARSE"] = "cpu"
    res = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--config", config] + extra,
                         capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    assert res.stdout.strip().splitlines() == lines, "stdout carries the JSON line and nothing else (RCCL's banner, library chatter: stderr)"           # rank 0 alone prints
    line = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "kernel_only_value", "exchange_bytes_per_rank"):
        assert key in line, key
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1 and line["value"] >= 0 and line["ms_per_step"] > 0
    assert line["scaling"] == ("strong" if config == "cfg5" else "weak")
    assert line["kernel_only_value"] >= 0 and line["roofline"]["avg_launch_ms"] > 0
    assert line["exchange_bytes_per_rank"]["sent"] > 0 and line["exchange_bytes_per_rank"]["received"] > 0
    assert "workload" in line["config"] and "exchange" in line["config"]
    if not extra or config == "cfg5":     # --gather auto (the default): the exchanges were timed before the ramp and the fastest FULL gather ran the headline
        modes = line["secondary"]["gather_modes"]
        assert set(modes) == {"allgather", "direct", "root"}
        for m in modes.values():
            assert m["exchange_only_ms"] > 0 and m["ms_per_step"] > 0 and m["achieved_GBps_per_link"] >= 0 and m["value_tflops"] >= 0
        chosen = [k for k, m in modes.items() if m["selected"]]
        assert chosen == [line["config"]["exchange_selected"]] and line["config"]["exchange_selection"].startswith("auto: fastest")
        assert not modes["root"]["eligible"] and chosen[0] != "root"     # rank 0's buffer alone is complete after "root": information only
        assert modes[chosen[0]]["ms_per_step"] == min(m["ms_per_step"] for m in modes.values() if m["eligible"])
    else:
        assert line["config"]["exchange_selected"] == "direct" and line["config"]["exchange_selection"].startswith("forced")
        assert "gather_modes" not in line["secondary"]
    assert line["scaling_efficiency"] > 0 and line["n1_anchor_kernel_tflops"] >= 0
    if config == "cfg2":       # the plain `--gpus N` command also answers BASELINE configs[4]: the strong record after the weak headline
        _check_strong_record(line["secondary"]["cfg5_strong"], 2, forced="direct" if extra else None)


def _check_strong_record(rec, world, forced=None):
    assert "error" not in rec, rec
    for key in ("value", "kernel_only_value", "ms_per_step", "exchange_only_ms", "achieved_GBps_per_link", "expected_link_bound_ms",
                "n1_anchor_kernel_tflops", "scaling_efficiency", "kernel_scaling_efficiency", "exchange_selected", "groups_total", "slabs"):
        assert key in rec, key
    assert rec["scaling"] == "strong" and rec["n_gpus"] == world and rec["ms_per_step"] > 0 and rec["exchange_only_ms"] > 0
    assert rec["groups_total"] == rec["groups_per_gpu"] * world and rec["groups_per_gpu"] == rec["groups_per_launch"] * rec["slabs"]
    assert rec["scaling_efficiency"] > 0 and rec["kernel_scaling_efficiency"] > 0
    if forced:
        assert rec["exchange_selected"] == forced and "gather_modes" not in rec
    else:
        assert rec["exchange_selected"] in ("allgather", "direct") and set(rec["gather_modes"]) == {"allgather", "direct"}


@pytest.mark.parametrize("config", ["cfg2", "cfg5"])
def test_bench_rehearsal_at_8_ranks_on_cpu(config):
    """The driver's 8-GPU command, rehearsed with no hardware (`FA_BENCH_REHEARSE=cpu python bench.py --gpus 8`, gloo, tiny fp32 stand-ins):
    the default command = weak cfg2 headline THEN BASELINE configs[4] (`secondary.cfg5_strong`: 8 x 8 slabs dividing the groups, slab-cyclic
    ownership, its own exchange selection, efficiency figures), and `--config cfg5` = configs[4] as the headline.  Eight ranks have to
    agree on every collective of the launch for the one JSON line to appear."""
    import json
    import pathlib
    import subprocess

    root = pathlib.Path(__file__).resolve().parent.parent
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(FA_BENCH_REHEARSE="cpu", OMP_NUM_THREADS="1")
    res = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1", "--config", config],
                         capture_output=True, text=True, timeout=900, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and len(lines[0]) < 8000, res.stdout[-2000:]
    assert res.stdout.strip().splitlines() == lines, "stdout carries the JSON line and nothing else"
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["scaling"] == ("strong" if config == "cfg5" else "weak")
    assert line["config"]["exchange_selected"] in ("allgather", "direct")
    assert line["exchange_bytes_per_rank"]["received"] == 7 * line["exchange_bytes_per_rank"]["sent"]
    modes = line["secondary"]["gather_modes"]
    assert set(modes) == {"allgather", "direct", "root"} and not modes["root"]["selected"]
    if config == "cfg2":
        _check_strong_record(line["secondary"]["cfg5_strong"], 8)
        assert line["secondary"]["cfg5_strong"]["groups_total"] % 64 == 0       # 8 ranks x 8 slabs
    else:
        assert "(128 groups in total over 8 GPUs)" in line["config"]["workload"]


def _worker8(rank, world, port, q):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "flash-attention-ipu_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    torch.set_num_threads(1)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from flash_attention_amd import distributed as fd
        from oracle import dense_causal_attention, frozen_randn

        G, L, D, slabs = 32, 16, 8, 2
        qkv = frozen_randn(23, (3, G, L, D))
        ref = dense_causal_attention(qkv)
        # count what "direct" hands to the backend: W - 1 = 7 sends of my shard + 7 receives into the owners' slots, one batch
        seen = []
        real = dist.batch_isend_irecv

        def counting(ops):
            seen.append([(o.op.__name__, o.peer) for o in ops])
            return real(ops)

        dist.batch_isend_irecv = counting
        lo, hi = fd.shard_range(G, rank, world)
        full = fd.flash_attention_qkv_packed_sharded(qkv[:, lo:hi].contiguous(), 2, 2, mode="direct")
        dist.batch_isend_irecv = real
        ok = torch.allclose(full, ref, atol=1e-6)
        ok = ok and len(seen) == 1 and len(seen[0]) == 14
        ok = ok and sorted(p for n, p in seen[0] if n == "isend") == sorted(p for n, p in seen[0] if n == "irecv") == [r for r in range(world) if r != rank]
        ok = ok and torch.allclose(fd.flash_attention_qkv_packed_sharded(qkv[:, lo:hi].contiguous(), 2, 2), ref, atol=1e-6)
        # slab-cyclic ownership at W = 8: the ranks' groups partition G, and the gathered buffer is in global group order for both exchanges
        owners = torch.cat([fd.SlabGather(G // world, L, D, torch.float32, "cpu", slabs=slabs).owned_groups(r) for r in range(world)])
        ok = ok and sorted(owners.tolist()) == list(range(G))
        for mode in fd.FULL_GATHER_MODES:
            sg = fd.SlabGather(G // world, L, D, torch.float32, "cpu", slabs=slabs, mode=mode)
            ok = ok and torch.allclose(sg.run(qkv[:, sg.owned_groups(rank)].contiguous(), 2, 2), ref, atol=1e-6)
        # the selection: eight ranks, one answer, never "root" (timed for information when asked)
        local = full[lo:hi].contiguous()
        mode, table = fd.select_gather_mode(local, step=lambda: local.clone(), reps=2, also_time=("root",))
        ok = ok and mode in fd.FULL_GATHER_MODES and set(table) == set(fd.GATHER_MODES) and not table["root"]["eligible"]
        q.put((rank, bool(ok), mode, [round(table[m]["ms_per_step"], 9) for m in fd.GATHER_MODES]))
    finally:
        dist.destroy_process_group()


def test_sharded_forward_and_selection_at_world_size_8_gloo():
    """W = 8 (the driver's node) on gloo: `direct` = 7 send + 7 receive pairs per rank in one batch, the full O on every rank and equal to
    the oracle; SlabGather's slab-cyclic ownership covers all G; the exchange selection returns the same full-gather mode and the same
    table on all eight ranks."""
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker8, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[:2] for r in res] == [(r, True) for r in range(world)]
    assert len({r[2] for r in res}) == 1 and len({tuple(r[3]) for r in res}) == 1
