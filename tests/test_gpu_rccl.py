"""RCCL on the GPU box: a process group of ONE rank with the nccl backend (= RCCL on ROCm), through exactly the calls bench.py makes
for N > 1 — init_process_group("nccl", device_id=...), barrier, shard.reduce_throughput(..., "cuda") — plus the min-index reduce of a
cross-GPU relocalisation split (shard.first_success).  The collectives of a one-rank group are trivial, but librccl is loaded, the
communicator is created on the GPU and the reductions run as RCCL kernels on its stream: what a one-GPU box can exercise of §8e."""
import importlib
import os
import socket

import pytest

pytestmark = pytest.mark.gpu


def test_one_rank_nccl_group_runs_the_bench_reductions():
    import torch
    import torch.distributed as dist
    assert torch.cuda.is_available()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        shard = importlib.import_module("slam-sdvl_amd.shard")
        dist.barrier()
        torch.cuda.synchronize()
        assert shard.reduce_throughput(1234, 0.5, dist, "cuda") == (1234.0, 0.5)
        assert shard.reduce_min_max(51.25, dist, "cuda") == (51.25, 51.25)   # the per-GPU link rates of the host-fed leg
        assert shard.first_success(7, dist, "cuda") == 7
        assert shard.first_success(None, dist, "cuda") is None
        assert dist.get_backend() == "nccl"
    finally:
        dist.destroy_process_group()
