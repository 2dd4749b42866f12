#!/usr/bin/env python3
"""K-conv: the convolve kernel alone on one frame worth of blocks — block-size sweep,
compound on/off, 8-bit / 10-bit, 1440p / 2160p.  Algorithmic bytes = P*bps per reference + P*bps written
+ 32 B per task; prints time per launch, GB/s and the fraction of the 8 TB/s HBM peak."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as g
pkg = g.load_pkg()
ctx = pkg.Context(0)
rng = np.random.default_rng(7)
out = []
# the last shape is eight 2160p frames stacked into one launch: the kernel without the launch floor
for (W, H) in ((2560, 1440), (3840, 2160), (3840, 2160 * 8)):
    for bd in (8, 10):
        dt = np.uint16 if bd > 8 else np.uint8
        refs = []
        for r in range(3):
            fr = pkg.DevFrame(ctx, W, H, bit_depth=bd)
            fr.upload([rng.integers(0, 1 << bd, (d[3], d[2])).astype(dt) for d in fr.dims])
            refs.append(fr)
        dst = pkg.DevFrame(ctx, W, H, bit_depth=bd)
        for bs in ((64, 32, 16, 8) if H <= 2160 else (64,)):
            for compound in ((0, 1) if H <= 2160 else (0,)):
                parts = []
                for p in range(3):
                    ss = 1 if p else 0
                    b = max(bs >> ss, 4)
                    pw, ph = W >> ss, H >> ss
                    xs, ys = np.meshgrid(np.arange(0, pw, b), np.arange(0, ph, b))
                    n = xs.size
                    t = np.zeros(n, pkg.INTER_DTYPE)
                    t["dst_x"], t["dst_y"] = xs.ravel(), ys.ravel()
                    t["w"] = b
                    t["h"] = b  # nominal block height (a VP9 shape); the kernels clip at the frame's edge
                    t["plane"] = p
                    t["flags"] = (rng.integers(0, 3, n) << 1) | compound
                    for r in range(2):
                        mv = rng.integers(-64 * 8, 64 * 8 + 1, (n, 2))          # eighth-pel luma
                        t["pos_x"][:, r] = (t["dst_x"].astype(np.int32) << 4) + mv[:, 0] * (2 >> ss)
                        t["pos_y"][:, r] = (t["dst_y"].astype(np.int32) << 4) + mv[:, 1] * (2 >> ss)
                        t["ref"][:, r] = rng.integers(0, 3, n)
                        t["step_x"][:, r] = t["step_y"][:, r] = 16
                    parts.append(t)
                # decode-like order: superblock raster of the luma grid, planes interleaved per row band
                tasks = np.concatenate(parts)
                key = (tasks["dst_y"].astype(np.int64) << np.where(tasks["plane"] > 0, 1, 0)) // 64 * 100000 + \
                      (tasks["dst_x"].astype(np.int64) << np.where(tasks["plane"] > 0, 1, 0)) // 64
                tasks = tasks[np.argsort(key, kind="stable")]
                tasks, counts = pkg.sort_inter_tasks(tasks, bd > 8)
                d_t = ctx.alloc(tasks)
                for i in range(3):
                    ctx.inter_pred_batch(d_t, counts, refs, dst)
                ctx.sync()
                N = 20
                for i in range(N):
                    ctx.timer_begin(i); ctx.inter_pred_batch(d_t, counts, refs, dst); ctx.timer_end(i)
                ctx.sync()
                ms = sorted(ctx.timer_read(i) for i in range(N))[N // 2]
                bps = 2 if bd > 8 else 1
                # samples inside the frame (a block of the bottom row overhangs it: the kernels clip)
                vis = np.minimum(tasks["h"].astype(np.int64), np.where(tasks["plane"] > 0, H >> 1, H) - tasks["dst_y"])
                px = (tasks["w"].astype(np.int64) * vis).sum()
                byts = px * bps * (2 + compound) + 32 * len(tasks)
                gbs = byts / (ms * 1e-3) / 1e9
                rec = dict(frame=f"{W}x{H}", bd=bd, block=bs, compound=compound, tasks=int(len(tasks)), ms=round(ms, 4),
                           algorithmic_MB=round(byts / 1e6, 2), GBps=round(gbs, 1), frac_hbm_peak=round(gbs / 8000, 4))
                out.append(rec)
                print(rec, flush=True)
                d_t.free()
        for fr in refs + [dst]:
            fr.free()
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "kconv.json"), "w"), indent=1)
