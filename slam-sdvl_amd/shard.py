"""Multi-GPU layout of the path: independent sequences shard across ranks, one process per GPU;
there is NO data-path collective — torch.distributed (RCCL on GPUs, gloo in the CPU tests) only carries the
barrier and the throughput counters.  Within a sequence the path does not shard (frame t+1 needs frame t)."""
import numpy as np

XI = np.array([0.004, 0.002, 0.001, 0.0008, -0.0012, 0.0005])  # SURVEY §8d trajectory twist per frame


def sequences_for_rank(rank, world, per_gpu):
    """global sequence ids owned by `rank` (weak scaling: every GPU gets `per_gpu` sequences)"""
    assert 0 <= rank < world and per_gpu > 0
    return list(range(rank * per_gpu, (rank + 1) * per_gpu))


def sequence_seed(gidx):
    return 20260001 + gidx


def sequence_twist(gidx):
    """per-sequence camera twist per frame: the S-A twist scaled / mirrored so that sequences differ"""
    s = 1.0 + 0.05 * (gidx % 7)
    sign = 1.0 if (gidx // 7) % 2 == 0 else -1.0
    return XI * s * np.array([sign, 1, 1, 1, sign, 1])


def reduce_throughput(tracked, elapsed, dist=None, device="cpu"):
    """(sum of tracked frames over ranks, max elapsed over ranks) — the only collective of a run"""
    import torch
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(tracked), float(elapsed)
    t_sum = torch.tensor([float(tracked)], dtype=torch.float64, device=device)
    t_max = torch.tensor([float(elapsed)], dtype=torch.float64, device=device)
    dist.all_reduce(t_sum, op=dist.ReduceOp.SUM)
    dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
    return float(t_sum[0]), float(t_max[0])
