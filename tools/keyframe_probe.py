#!/usr/bin/env python3
"""Time of an all-intra (key) frame through the pipeline, by phase."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g
pkg = g.load_pkg()
import cuda_vp9_amd.pipeline as pipeline
import workload
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2560, 1440)
ctx = pkg.Context(0)
wl = workload.make_frame_workload(W, H, seed=5, all_intra=True)
print("tasks", len(wl["intra_sorted"]), "waves", wl["n_waves"], "islands", len(wl["intra_islands"]), "big tasks",
      len(wl["intra_big_tasks"]), "big waves", len(wl["intra_big_wave_start"]) - 1)
job = pipeline.FrameJob(ctx, wl)
job.run(); ctx.sync()
for ph in ("intra", "lf"):
    ts = []
    for i in range(5):
        ctx.timer_begin(i); job.run(phases=(ph,)); ctx.timer_end(i)
    ctx.sync()
    ts = sorted(ctx.timer_read(i) for i in range(5))
    print(f"{ph}: {ts[2]:.3f} ms")
