#!/usr/bin/env python3
"""Loop-filter chain probe: time of vp9hip_loop_filter_frame on frames 2560 wide and 1, 2, 4, 8, 23 superblock rows
tall (uniformly random content, masks of a random partition, level 32): the top row's own chain and what every
further row adds (the row-to-row lag)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as g
hip = g.load_pkg()
import synth
ctx = hip.Context(0)
rng = np.random.default_rng(17)
W = int(sys.argv[1]) if len(sys.argv) > 1 else 2560
prev = None
for rows in (1, 2, 4, 8, 16, 23):
    H = rows * 64
    frame = hip.DevFrame(ctx, W, H, bit_depth=8)
    dims = [(d[2], d[3]) for d in frame.dims]
    frame.upload([rng.integers(0, 256, (d[1], d[0])).astype(np.uint8) for d in dims])
    mi_rows, mi_cols = H // 8, W // 8
    sb_rows, sb_cols = rows, (mi_cols + 7) // 8
    lfm = synth.random_lfm(rng, sb_rows, sb_cols, mi_rows, mi_cols, hip.LFM_DTYPE)
    mblim, lim, hev = synth.lf_thresholds(0)
    th = hip.LfThresh()
    for i in range(64):
        th.mblim[i], th.lim[i], th.hev_thr[i] = int(mblim[i]), int(lim[i]), int(hev[i])
    d_lfm = ctx.alloc(lfm)
    for _ in range(3):
        ctx.loop_filter_frame(d_lfm, sb_rows, sb_cols, th, frame, 3)
    ctx.sync()
    N = 20
    for i in range(N):
        ctx.timer_begin(i); ctx.loop_filter_frame(d_lfm, sb_rows, sb_cols, th, frame, 3); ctx.timer_end(i)
    ctx.sync()
    us = sorted(ctx.timer_read(i) for i in range(N))[N // 2] * 1e3
    extra = "" if prev is None else f"  (+{(us - prev[1]) / (rows - prev[0]):.2f} us per added row)"
    print(f"{rows:2d} superblock rows: {us:7.1f} us{extra}")
    prev = (rows, us)
    d_lfm.free(); frame.free()
