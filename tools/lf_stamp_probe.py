import os, sys, ctypes
sys.path.insert(0, '/root/repo')
import numpy as np
import __graft_entry__ as g
pkg = g.load_pkg()
import cuda_vp9_amd.pipeline as pipeline, cuda_vp9_amd.workload as workload
for bd in (8, 10):
    ctx = pkg.Context(0)
    wl = workload.make_frame_workload(2560, 1440, seed=1440, bd=bd)
    job = pipeline.FrameJob(ctx, wl)
    job.run(); ctx.sync()
    out = (ctypes.c_ulonglong * 16)()
    pkg.lib().vp9hip_lfdebug_read(out); a = list(out)
    N = 20
    for i in range(N):
        job.run(phases=("lf",))
    ctx.sync()
    pkg.lib().vp9hip_lfdebug_read(out); b = list(out)
    d = [y - x for x, y in zip(a, b)]
    steps = d[8]
    names = ["loop top", "V pass", "barrier A", "H pass + strip", "barrier B"]
    print(bd, "bit, row 0 luma, cycles per step:", {n: round(d[i] / steps) for i, n in enumerate(names)}, "total", round(sum(d[:5]) / steps))
