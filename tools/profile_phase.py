#!/usr/bin/env python3
"""Run one phase of the frame pipeline repeatedly (for rocprofv3 kernel-trace / --pmc runs).
    rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU ... -- python3 tools/profile_phase.py --phase inter
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
ap = argparse.ArgumentParser()
ap.add_argument("--phase", default="inter")
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--width", type=int, default=2560)
ap.add_argument("--height", type=int, default=1440)
ap.add_argument("--no-overlap", action="store_true",
                help="intra and loop filter in sequence: rocprofv3 --pmc runs one kernel at a time, and the "
                     "overlapped pair (the filter polls counters the island walk raises) needs both resident")
args = ap.parse_args()
import __graft_entry__ as g  # noqa: E402
pkg = g.load_pkg()
import cuda_vp9_amd.pipeline as pipeline  # noqa: E402
import workload  # noqa: E402
ctx = pkg.Context(0)
wl = workload.make_frame_workload(args.width, args.height, seed=1440)
job = pipeline.FrameJob(ctx, wl)
if args.no_overlap:
    job.overlap = False
job.run()            # full frame once so that every phase has realistic input
ctx.sync()
phases = ("inter", "txb", "intra", "lf") if args.phase == "all" else (args.phase,)
for i in range(args.steps):
    ctx.timer_begin(i)
    job.run(phases=phases)
    ctx.timer_end(i)
ctx.sync()
ts = sorted(ctx.timer_read(i) for i in range(args.steps))
print(f"{args.phase}: median {ts[len(ts)//2]*1e3:.1f} us, min {ts[0]*1e3:.1f} us over {args.steps} runs; "
      f"algorithmic bytes {pipeline.algorithmic_bytes(wl)}")
