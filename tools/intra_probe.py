#!/usr/bin/env python3
"""Timing probe for the intra island kernel: which part of a chunk costs the time?
Variants of the same 1440p frame: as is / prediction only (eob = 0) / all DC mode / both."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as g
pkg = g.load_pkg()
import cuda_vp9_amd.pipeline as pipeline
import workload
ctx = pkg.Context(0)
base = workload.make_frame_workload(2560, 1440, seed=1440)
isl = base["intra_islands"]
print("islands", len(isl), "tasks", len(base["intra_island_tasks"]), "max waves", int(isl["n_waves"].max()),
      "sum waves", int(isl["n_waves"].sum()))
woff = base["intra_island_wave_off"]
chunks = 0
for r in isl:
    for w in range(r["n_waves"]):
        n = woff[r["wave_off_start"] + w + 1] - woff[r["wave_off_start"] + w]
        chunks += (n + 7) // 8
deep = isl[np.argmax(isl["n_waves"])]
dchunks = sum((woff[deep["wave_off_start"] + w + 1] - woff[deep["wave_off_start"] + w] + 7) // 8 for w in range(deep["n_waves"]))
print("total chunks", chunks, "deepest island: waves", int(deep["n_waves"]), "chunks", int(dchunks))
for name, mod in (("as is", None), ("eob=0", "eob"), ("mode=DC", "mode"), ("eob=0,mode=DC", "both"), ("tx 4x4 only kept", "small")):
    wl = dict(base)
    t = base["intra_island_tasks"].copy()
    if mod in ("eob", "both"):
        t["eob"] = 0
    if mod in ("mode", "both"):
        t["mode"] = 0
    wl["intra_island_tasks"] = t
    job = pipeline.FrameJob(ctx, wl)
    job.run()
    ctx.sync()
    ts = []
    for i in range(20):
        ctx.timer_begin(i)
        job.run(phases=("intra",))
        ctx.timer_end(i)
    ctx.sync()
    ts = sorted(ctx.timer_read(i) for i in range(20))
    print(f"{name:>16}: median {ts[10]*1e3:.1f} us")
    job.free()
