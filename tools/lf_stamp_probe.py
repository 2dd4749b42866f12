import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_pkg()
import cuda_vp9_amd.pipeline as pipeline, cuda_vp9_amd.workload as workload
for bd in (8, 10):
    ctx = pkg.Context(0)
    wl = workload.make_frame_workload(2560, 1440, seed=1440, bd=bd)
    job = pipeline.FrameJob(ctx, wl)
    job.run(); ctx.sync()
    out = (ctypes.c_ulonglong * 24)()
    pkg.lib().vp9hip_lfdebug_read(out); a = list(out)
    for i in range(20):
        job.run(phases=("lf",))
    ctx.sync()
    pkg.lib().vp9hip_lfdebug_read(out); b = list(out)
    d = [y - x for x, y in zip(a, b)]
    steps = d[20]
    for w, role in enumerate(("filter", "data-in", "publisher", "write-back")):
        print(bd, "bit row 0 luma wave", w, role, "cycles per step: top %d | phase A work %d | barrier A wait %d | phase B work %d | barrier B wait %d" % tuple(round(d[w * 5 + i] / steps) for i in range(5)))
