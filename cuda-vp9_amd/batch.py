"""batch.py — multi-GPU batch harness logic: independent streams are sharded one decoder per
GPU (stream i -> rank i mod world); there is no data-path collective.  The only
communication is the end-of-batch barrier + stats reduce (sum of frames / failures, max of wall
seconds) — RCCL on GPUs (torch backend "nccl"), gloo in the CPU tests."""
from typing import List, Sequence


def shard_streams(n_streams: int, rank: int, world: int) -> List[int]:
    """Indices of the streams this rank decodes."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    return list(range(rank, n_streams, world))


def reduce_stats(dist, frames: float, md5_failures: float, seconds: float, device=None):
    """(total frames, total md5 failures, slowest rank's seconds).  `dist` is
    torch.distributed (initialised) or None for a single process."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(frames), float(md5_failures), float(seconds)
    import torch
    s = torch.tensor([float(frames), float(md5_failures)], dtype=torch.float64, device=device)
    dist.all_reduce(s, op=dist.ReduceOp.SUM)
    m = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    dist.all_reduce(m, op=dist.ReduceOp.MAX)
    return float(s[0].item()), float(s[1].item()), float(m[0].item())


def throughput(frames_total: float, seconds_max: float) -> float:
    return frames_total / seconds_max if seconds_max > 0 else 0.0
