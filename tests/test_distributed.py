"""world_size-2 gloo tests of the sharded path on CPU (runs without a GPU)."""

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
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_sharded_forward_world2_gloo():
    world, G, L, D = 2, 6, 32, 8
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
    assert sorted(results) == [(0, True), (1, True)]


def test_shard_range():
    from flash_attention_amd.distributed import shard_range

    assert [shard_range(128, r, 8) for r in (0, 7)] == [(0, 16), (112, 128)]
    with pytest.raises(ValueError):
        shard_range(10, 0, 4)
