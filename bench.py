#!/usr/bin/env python3
"""bench.py — VP9 block-reconstruction throughput on MI355X.

A "step" is one pass of the hot path (inter prediction -> inverse transform + add -> wave-ordered
intra prediction -> loop filter) over one synthetic 2560x1440 8-bit 4:2:0 frame whose packed work
lists and reference frames are already resident in HBM.  BASELINE.json's configs[1]
(Bravia.1440.ivf) cannot be used: the clip is not in the reference snapshot; the workload is the
seeded synthetic stand-in S-1440 (cuda-vp9_amd/workload.py).  One independent stream per GPU
(--gpus N, launched with torch.distributed.run): no data-path collective, RCCL only for the
barrier and the end-of-batch stats reduce.

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--width", type=int, default=2560)
    ap.add_argument("--height", type=int, default=1440)
    ap.add_argument("--bit-depth", type=int, default=8)
    ap.add_argument("--frames", type=int, default=4, help="distinct resident frames cycled through")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--no-phase-timers", action="store_true")
    ap.add_argument("--streams", type=int, default=8,
                    help="extra leg: this many independent streams in flight on the GPU (0 = skip); reported "
                         "under multi_stream, never as `value`")
    args = ap.parse_args()

    import torch
    import __graft_entry__ as g

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(local_rank)

    pkg = g.load_pkg()
    import cuda_vp9_amd.pipeline as pipeline
    import cuda_vp9_amd.workload as workload

    ctx = pkg.Context(local_rank)
    wls = [workload.make_frame_workload(args.width, args.height, seed=1440 + i, bd=args.bit_depth)
           for i in range(args.frames)]
    jobs = [pipeline.FrameJob(ctx, wl) for wl in wls]
    ab = [pipeline.algorithmic_bytes(wl) for wl in wls]
    PH = ("inter", "txb", "intra", "lf")
    KN = {"inter": "convolve", "txb": "idct_add", "intra": "intra", "lf": "loop_filter"}

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def step(i):
        jobs[i % len(jobs)].run()  # all four phases; intra and the loop filter overlap (two HIP streams)

    for i in range(args.warmup):
        step(i)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    barrier()
    elapsed = time.perf_counter() - t0

    # per-kernel-family GPU time: HIP events on the launch stream around each phase launched on its
    # own (the phases of the timed steps above overlap, so they cannot be told apart there), same
    # process, same resident frames, right after the timed region
    timers = not args.no_phase_timers
    n_timed = min(args.steps, 100) if timers else 0
    phase_ms = {ph: 0.0 for ph in PH}
    for i in range(n_timed):
        job = jobs[i % len(jobs)]
        for k, ph in enumerate(PH):
            ctx.timer_begin(4 * i + k)
            job.run(phases=(ph,))
            ctx.timer_end(4 * i + k)
    if n_timed:
        ctx.sync()
        for i in range(n_timed):
            for k, ph in enumerate(PH):
                phase_ms[ph] += ctx.timer_read(4 * i + k)
    phase_ms = {ph: (v / n_timed if n_timed else None) for ph, v in phase_ms.items()}

    # ---- extra leg: several independent streams in flight on one GPU (one context = one HIP stream
    # each).  The loop filter keeps 69 of 256 CUs busy, so frames of other streams fit beside it.
    multi = None
    if args.streams > 1:
        ctxs = [pkg.Context(local_rank) for _ in range(args.streams)]
        mjobs = [[pipeline.FrameJob(c, wl) for wl in wls[:2]] for c in ctxs]
        for w in range(3):
            for js in mjobs:
                js[w % len(js)].run()
        for c in ctxs:
            c.sync()
        barrier()
        n_rounds = max(10, args.steps // 4)
        tm0 = time.perf_counter()
        for i in range(n_rounds):
            for js in mjobs:
                js[i % len(js)].run()
        for c in ctxs:
            c.sync()
        barrier()
        tm = time.perf_counter() - tm0
        multi = {"streams": args.streams, "frames": n_rounds * args.streams,
                 "frames_per_s": round(n_rounds * args.streams / tm, 1)}
        multi_got = [js[0].download() for js in mjobs] if rank == 0 else None  # checked against the oracle below
        for js in mjobs:
            for j in js:
                j.free()
        for c in ctxs:
            c.close()

    # correctness of what was timed: frame 0 against the oracle (rank 0 only; small CPU cost)
    md5_match = None
    cpu_baseline = None
    if rank == 0:
        import frame_check
        oracle = frame_check.load_oracle()
        jobs[0].run()  # the phase-timing loop above re-ran single phases on the frame: rebuild it
        ctx.sync()
        got = jobs[0].download()
        exp, _ = frame_check.oracle_frame(oracle, wls[0])
        md5_match = frame_check.frame_md5(got, wls[0]) == frame_check.frame_md5(exp, wls[0])
        if multi is not None:
            multi["md5_match_vs_oracle"] = all(frame_check.frame_md5(g_, wls[0]) == frame_check.frame_md5(exp, wls[0])
                                               for g_ in multi_got)
        if world == 1 and not args.no_cpu_baseline:
            n, t_cpu = 0, 0.0
            while t_cpu < args.cpu_seconds or n < 2:
                _, times = frame_check.oracle_frame(oracle, wls[n % len(wls)])
                t_cpu += sum(times.values())
                n += 1
            cpu_baseline = {"value": round(n / t_cpu, 3), "unit": "frames/s", "cores": 1, "kind": "port",
                            "sample": f"{n} synthetic {args.width}x{args.height} frames (same work lists as the GPU "
                                      f"run) through oracle/ C restatement, single thread, {t_cpu:.1f} s"}

    # stats reduce: total frames (sum) and slowest rank (max) — the only collective in the harness
    import cuda_vp9_amd.batch as batch
    frames_total, _, t_max = batch.reduce_stats(dist, args.steps, 0.0, elapsed,
                                                device=torch.device("cuda", local_rank))
    if multi is not None:
        m_total, _, _ = batch.reduce_stats(dist, multi["frames_per_s"], 0.0, 0.0, device=torch.device("cuda", local_rank))
        multi["frames_per_s_all_gpus"] = round(m_total, 1)

    if rank == 0:
        kernels = {}
        if n_timed:
            for ph in PH:
                byts = sum(a[KN[ph]] for a in ab) / len(ab)
                ms = phase_ms[ph]
                gbs = byts / (ms * 1e-3) / 1e9 if ms and ms > 0 else None
                kernels[KN[ph]] = {"ms_per_frame": round(ms, 5), "algorithmic_bytes": int(byts),
                                   "GB/s": round(gbs, 1) if gbs else None,
                                   "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 5) if gbs else None}
        roofline = None
        if kernels:
            dom = max(kernels, key=lambda k: kernels[k]["ms_per_frame"])
            # HBM bytes per launch from the PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate
            # runs of the same workload, tools/pmc_collect.sh; FETCH_SIZE doubled as calibrated on gfx950 with
            # tools/fetch_calibrate.hip) — measured offline, committed under profiles/
            traffic, pmc = None, {}
            try:
                pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_current.json")))
                e = pmc.get(dom)
                if e and args.width == 2560 and args.height == 1440 and args.bit_depth == 8:
                    traffic = int(e["hbm_read_bytes_gfx950_x2"] + e["hbm_write_bytes"])
            except (OSError, KeyError, ValueError):
                pass
            for k, v in kernels.items():
                e = pmc.get(k)
                if e and traffic is not None:
                    v["hbm_traffic_bytes_pmc"] = int(e["hbm_read_bytes_gfx950_x2"] + e["hbm_write_bytes"])
                    e2 = pmc.get(k + "_residual")  # the intra phase is two kernels: residual pre-pass + walk
                    if e2:
                        v["hbm_traffic_bytes_pmc"] += int(e2["hbm_read_bytes_gfx950_x2"] + e2["hbm_write_bytes"])
                    if "lds_bank_conflict_frac" in e:
                        v["lds_bank_conflict_frac_pmc"] = e["lds_bank_conflict_frac"]
            roofline = {"kernel": dom, "bound": "hbm", "achieved": kernels[dom]["GB/s"], "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": kernels[dom]["frac_of_hbm_peak"], "traffic": traffic}
        out = {
            "metric": "decoded frames/sec (block-reconstruction path: inter+idct+intra+loop filter), 1440p VP9 8-bit",
            "value": round(frames_total / t_max, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * t_max / args.steps, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8" if args.bit_depth == 8 else "u16",
            "data": "synthetic",
            "config": {"workload": f"S-{args.height}: synthetic {args.width}x{args.height} {args.bit_depth}-bit 4:2:0 "
                                   f"inter frame ({wls[0]['n_blocks']} blocks, {len(wls[0]['inter_tasks'])} inter tasks, "
                                   f"{len(wls[0]['txb'])} coded inter tx blocks, {len(wls[0]['intra_sorted'])} intra tx "
                                   f"blocks in {wls[0]['n_waves']} waves, {wls[0]['sb_rows']}x{wls[0]['sb_cols']} SBs), "
                                   f"{args.frames} distinct frames resident in HBM, one stream per GPU",
                       "parallelism": f"streams{world}"},
            "md5_match_vs_oracle": md5_match,
            "roofline": roofline, "kernels": kernels, "cpu_baseline": cpu_baseline, "multi_stream": multi,
        }
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
