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


def chunk_for_sequence(gidx, per_gpu, world, chunks=4):
    """S-B (BASELINE config 4: "1 vs 4 GPUs sharding independent sequence chunks"): which of the `chunks` independent
    chunks global sequence `gidx` follows.  SURVEY §8e's partition: rank g owns the chunks {c : c mod G = g}; the rank's sequences
    go round those.  One rank: all chunks, sequence j follows chunk j mod chunks; as many ranks as chunks: every rank tracks ONE
    chunk; more ranks than chunks: rank r tracks chunk r mod chunks."""
    assert per_gpu > 0 and world > 0 and chunks > 0 and gidx >= 0
    rank, j = divmod(gidx, per_gpu)
    mine = [c for c in range(chunks) if c % world == rank % world] if world < chunks else [rank % chunks]
    return mine[j % len(mine)]


def sequence_twist(gidx):
    """per-sequence camera twist per frame: the S-A twist scaled / mirrored so that sequences differ"""
    s = 1.0 + 0.05 * (gidx % 7)
    sign = 1.0 if (gidx // 7) % 2 == 0 else -1.0
    return XI * s * np.array([sign, 1, 1, 1, sign, 1])


def reduce_throughput(tracked, elapsed, dist=None, device="cpu"):
    """(sum of tracked frames over ranks, max elapsed over ranks) — the only collective of a run.  With a process group up the
    reduction always goes through it, also for a group of one rank (RCCL is exercised on a one-GPU box)."""
    import torch
    if dist is None or not dist.is_initialized():
        return float(tracked), float(elapsed)
    t_sum = torch.tensor([float(tracked)], dtype=torch.float64, device=device)
    t_max = torch.tensor([float(elapsed)], dtype=torch.float64, device=device)
    dist.all_reduce(t_sum, op=dist.ReduceOp.SUM)
    dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
    return float(t_sum[0]), float(t_max[0])


def reduce_min_max(value, dist=None, device="cpu"):
    """(min over ranks, max over ranks) of one per-rank figure — the per-GPU link rates of the host-fed leg, whose ranks share the
    host's DRAM and PCIe roots: the slowest rank shows what the shared side costs"""
    import torch
    if dist is None or not dist.is_initialized():
        return float(value), float(value)
    lo = torch.tensor([float(value)], dtype=torch.float64, device=device)
    hi = lo.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    return float(lo[0]), float(hi[0])


# ---- relocalisation over several GPUs (SDVL::Relocalize, sdvl.cc:205-238) ----------------------------------------------
# Relocalize walks the keyframes NEWEST FIRST and accepts the first one whose alignment error is < 0.001 and whose reprojection
# finds >= MinMatches points (:221, :231).  Split over G GPUs, rank r examines positions r, r + G, r + 2G, ... of that newest-first
# order (round-robin, so every rank gets recent and old keyframes alike), and the answer is the SMALLEST position any rank
# accepted: one all-reduce(MIN) of one int64.  In this build a tracker's keyframes live in the HBM of the GPU that tracked them and
# relocalisation runs there as one launch over all of them; these two functions are the partition and the reduce a
# cross-GPU split uses, tested over gloo (two ranks) and over RCCL (tests/test_gpu_rccl.py).
NO_KEYFRAME = (1 << 62)


def keyframe_positions_for_rank(n_keyframes, rank, world):
    """positions (0 = newest) of the newest-first keyframe order that `rank` examines"""
    assert 0 <= rank < world and n_keyframes >= 0
    return list(range(rank, n_keyframes, world))


def first_success(local_position, dist=None, device="cpu"):
    """`local_position`: the smallest newest-first position this rank accepted, or None.  Returns the position Relocalize would have
    stopped at had it walked all keyframes alone (the minimum over ranks), or None when no rank accepted any."""
    import torch
    mine = NO_KEYFRAME if local_position is None else int(local_position)
    assert 0 <= mine <= NO_KEYFRAME
    if dist is not None and dist.is_initialized():
        t = torch.tensor([mine], dtype=torch.int64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        mine = int(t[0])
    return None if mine >= NO_KEYFRAME else mine
