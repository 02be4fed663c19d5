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
