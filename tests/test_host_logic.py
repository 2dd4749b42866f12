"""Host-side packing logic (no GPU): loop-filter mask construction, intra dependency waves,
stream sharding, stats reduce over gloo with world_size 2."""
import os
import sys

import numpy as np
import pytest


def test_lf_masks_are_exclusive_and_trimmed(hip):
    import workload
    for (W, H) in [(352, 288), (200, 136), (328, 72)]:
        wl = workload.make_frame_workload(W, H, seed=W)
        aw, ah = wl["dims"][0]
        mi_rows, mi_cols = ah // 8, aw // 8
        for i, m in enumerate(wl["lfm"]):
            sr, sc = divmod(i, wl["sb_cols"])
            for side in ("left_y", "above_y", "left_uv", "above_uv"):
                a, b, c = (int(m[side][k]) for k in range(3))
                assert a & b == 0 and a & c == 0 and b & c == 0      # vp9_adjust_mask asserts (:842-857)
                assert int(m[side][3]) == 0 or side.endswith("y") or True
            rows, cols = min(8, mi_rows - sr * 8), min(8, mi_cols - sc * 8)
            inside = sum(((1 << cols) - 1) << (8 * r) for r in range(rows))
            for k in range(3):
                assert int(m["left_y"][k]) & ~inside == 0 and int(m["above_y"][k]) & ~inside == 0
            assert int(m["int_4x4_y"]) & ~inside == 0
            if sc == 0:
                assert all(int(m["left_y"][k]) & 0x0101010101010101 == 0 for k in range(3))
                assert all(int(m["left_uv"][k]) & 0x1111 == 0 for k in range(3))


def test_intra_waves_respect_dependencies(hip):
    import workload
    wl = workload.make_frame_workload(320, 192, seed=5, intra_frac=0.6)
    tasks = wl["intra_decode_order"]
    lv = workload.intra_levels(tasks, wl["dims"])
    assert lv.min() >= 1
    # every task's level exceeds that of any earlier task whose pixels it reads
    owner = [np.full((ah // 4 + 2, aw // 4 + 2), -1, np.int64) for (aw, ah) in wl["dims"]]
    for i, t in enumerate(tasks):
        m = owner[t["plane"]]
        cx, cy, n = t["x"] // 4, t["y"] // 4, 1 << int(t["tx_size"])
        deps = set()
        if t["flags"] & 2:
            deps |= set(m[cy:cy + n, cx - 1].ravel())
        if t["flags"] & 1:
            ext = 2 * n if (n == 1 and t["flags"] & 4) else n
            deps |= set(m[cy - 1, cx:cx + ext].ravel())
            if t["flags"] & 2:
                deps.add(m[cy - 1, cx - 1])
        for d in deps:
            if d >= 0:
                assert lv[d] < lv[i]
        m[cy:cy + n, cx:cx + n] = i
    ws = wl["wave_start"]
    assert ws[0] == 0 and ws[-1] == len(tasks) and (np.diff(ws) >= 0).all()


def test_shard_streams(hip):
    import cuda_vp9_amd.batch as batch
    got = sorted(sum((batch.shard_streams(19, r, 8) for r in range(8)), []))
    assert got == list(range(19))
    assert batch.shard_streams(8, 3, 8) == [3]
    with pytest.raises(ValueError):
        batch.shard_streams(4, 4, 4)


def _worker(rank, world, port, q):
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import __graft_entry__ as g
    g.load_pkg()
    import cuda_vp9_amd.batch as batch
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    mine = batch.shard_streams(5, rank, world)
    frames = 100.0 * len(mine)
    total, fails, tmax = batch.reduce_stats(dist, frames, float(rank), 1.0 + rank)
    dist.barrier()
    q.put((rank, total, fails, tmax))
    dist.destroy_process_group()


def test_stats_reduce_gloo_world2(hip):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 500)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for (_, total, fails, tmax) in res:
        assert total == 500.0 and fails == 1.0 and tmax == 2.0


def test_bench_launches_its_own_ranks_gloo_dry_run():
    """`python bench.py --gpus 2` is self-contained: the parent starts one rank per GPU (here: per CPU
    process, gloo, --dry-run = launcher + stream sharding + C packer + stats reduce, no GPU call)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run", "--backend", "gloo",
                        "--width", "352", "--height", "288", "--frames", "2"], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    line = [l for l in r.stdout.decode().splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["frames_packed_all_ranks"] == 4.0 and out["streams_of_rank0"] == [0]


def test_bench_refuses_more_gpus_than_visible_cleanly():
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "64"], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 2
    assert "GPU(s) visible" in json.loads(r.stdout.decode().strip().splitlines()[-1])["error"]


def test_rank_placement_on_a_described_host():
    """Per-rank CPU blocks and thread budgets (cuda-vp9_amd/batch.py rank_placement) on a described
    128-core, two-socket host with eight GPUs — four per NUMA node — and on hosts that say less."""
    import importlib
    import __graft_entry__ as g
    g.load_pkg()
    batch = importlib.import_module("cuda_vp9_amd.batch")
    assert batch.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    topo = {"allowed": list(range(128)), "nodes": {0: list(range(64)), 1: list(range(64, 128))},
            "gpu_numa": [0, 0, 0, 0, 1, 1, 1, 1]}
    seen = set()
    for r in range(8):
        p = batch.rank_placement(r, 8, topo)
        assert len(p["cpus"]) == 16 and p["cpus"] == list(range(p["cpus"][0], p["cpus"][0] + 16))  # a contiguous block
        assert set(p["cpus"]) <= set(topo["nodes"][topo["gpu_numa"][r]])                          # on the GPU's node
        assert not (seen & set(p["cpus"]))                                                        # nobody else's
        seen |= set(p["cpus"])
        assert 1 + p["entropy_threads"] + p["pack_threads"] <= 16 and p["entropy_threads"] == 8
    assert seen == set(range(128))
    # one GPU of eight on its own: the whole first node
    assert batch.rank_placement(0, 1, topo)["cpus"] == list(range(64))
    # a cgroup that allows 32 CPUs across both nodes: blocks come from what is allowed on the GPU's node
    topo2 = dict(topo, allowed=list(range(0, 16)) + list(range(64, 80)))
    assert batch.rank_placement(5, 8, topo2)["cpus"] == list(range(68, 72))
    # no NUMA information: an even split of the allowed CPUs
    flat = {"allowed": list(range(16)), "nodes": {}, "gpu_numa": []}
    assert [batch.rank_placement(r, 4, flat)["cpus"] for r in range(4)] == [list(range(4 * r, 4 * r + 4)) for r in range(4)]
    # more ranks than CPUs: everybody still gets one, budgets of one thread each
    tiny = {"allowed": [0, 1], "nodes": {}, "gpu_numa": []}
    p = batch.rank_placement(3, 8, tiny)
    assert len(p["cpus"]) == 1 and p["entropy_threads"] == 1 and p["pack_threads"] == 1
    # this host, whatever it is: a non-empty subset of the CPUs the process may use
    here = batch.host_topology()
    p = batch.rank_placement(0, 2, here)
    assert p["cpus"] and set(p["cpus"]) <= set(here["allowed"])
