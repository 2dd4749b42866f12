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


# ---- per-rank host placement ("give each rank >= 1 dedicated core (+ tile threads), pin to the GPU's NUMA
# node") ---------------------------------------------------------------------------------------------------------------
def parse_cpulist(text: str) -> List[int]:
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11] (the format of /sys/devices/system/node/node*/cpulist)."""
    cpus: List[int] = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return cpus


def host_topology():
    """What this host says about itself: the CPUs this process may run on, the CPUs of every NUMA node, and the NUMA
    node of every GPU in PCI-bus order (the order HIP enumerates them in by default); {} entries where sysfs is silent."""
    import glob
    import os
    allowed = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    nodes = {}
    for path in glob.glob("/sys/devices/system/node/node[0-9]*/cpulist"):
        try:
            nodes[int(path.split("node")[-1].split("/")[0])] = parse_cpulist(open(path).read())
        except (OSError, ValueError):
            pass
    gpus = []
    for dev in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
        try:
            if open(os.path.join(dev, "vendor")).read().strip() != "0x1002":
                continue
            gpus.append((os.path.basename(os.path.realpath(dev)), int(open(os.path.join(dev, "numa_node")).read())))
        except (OSError, ValueError):
            pass
    gpus.sort()
    return {"allowed": allowed, "nodes": nodes, "gpu_numa": [n for _, n in gpus]}


def rank_placement(local_rank: int, world: int, topo: dict) -> dict:
    """CPUs and thread budgets of one rank of `world` on this host.

    Ranks whose GPU sits on the same NUMA node share that node's allowed CPUs in equal contiguous blocks (in rank
    order); without NUMA information all allowed CPUs are split evenly.  A block is at least one CPU; threads are
    budgeted from it instead of the fixed 8 + 8 of a single-GPU run: one CPU stays with the rank's main thread (GPU
    submission, the front-end's merge), the rest is split between the tile-column entropy threads and the packer."""
    allowed = list(topo.get("allowed") or [0])
    gpu_numa = list(topo.get("gpu_numa") or [])
    nodes = topo.get("nodes") or {}
    pool, peers = allowed, list(range(world))
    if len(gpu_numa) >= world and all(n >= 0 and n in nodes for n in gpu_numa[:world]):
        mine = gpu_numa[local_rank]
        node_cpus = [c for c in nodes[mine] if c in set(allowed)]
        if node_cpus:
            pool, peers = node_cpus, [r for r in range(world) if gpu_numa[r] == mine]
    k, n = peers.index(local_rank), len(peers)
    per = max(1, len(pool) // n)
    cpus = pool[k * per:(k + 1) * per] if (k + 1) * per <= len(pool) else [pool[(k * per) % len(pool)]]
    spare = max(0, len(cpus) - 1)
    entropy = max(1, min(8, (spare * 2 + 2) // 3))      # tile columns of a 1440p stream: 8
    pack = max(1, min(8, spare - entropy if spare > entropy else 1))
    return {"cpus": cpus, "entropy_threads": entropy, "pack_threads": pack}


def apply_placement(p: dict) -> bool:
    """Pins this process (and the threads it starts later) to the rank's CPUs; False where the host refuses."""
    import os
    try:
        os.sched_setaffinity(0, p["cpus"])
        return True
    except (AttributeError, OSError, ValueError):
        return False
