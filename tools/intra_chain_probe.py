#!/usr/bin/env python3
"""Latency of ONE dependent intra block in the island walk, by transform size / mode / coded:
islands that are horizontal chains of K blocks (each needs its left neighbour)."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as g
pkg = g.load_pkg()
ctx = pkg.Context(0)
W, H = 2560, 1440
frame = pkg.DevFrame(ctx, W, H)
rng = np.random.default_rng(1)
frame.upload([rng.integers(0, 256, (d[3], d[2])).astype(np.uint8) for d in frame.dims])
K = 40
for txs in range(4):
    bs = 4 << txs
    for mode, eob in ((0, 0), (3, 0), (9, 0), (0, bs * bs), (3, bs * bs), (0, 1)):
        n_isl = 64
        tasks = np.zeros(n_isl * K, pkg.INTRA_DTYPE)
        isl = np.zeros(n_isl, pkg.ISLAND_DTYPE)
        woff = []
        coeffs = rng.integers(-50, 50, n_isl * K * bs * bs + 16).astype(np.int32)
        for i in range(n_isl):
            isl[i]["task_start"], isl[i]["wave_off_start"], isl[i]["n_waves"] = i * K, len(woff), K
            for k in range(K):
                t = tasks[i * K + k]
                t["x"], t["y"], t["plane"], t["tx_size"], t["mode"], t["eob"] = 64 + k * bs, 32 + i * 40 % 1300, 0, txs, mode, eob
                t["y"] = 32 + (i * 40) % 1300 + (i // 32) * 0
                t["coeff_off"] = (i * K + k) * bs * bs
                t["flags"] = 3
                woff.append(k)
            woff.append(K)
        # two islands per row band would overlap: spread rows
        ys = 32 + np.arange(n_isl) * 20
        for i in range(n_isl):
            tasks["y"][i * K:(i + 1) * K] = ys[i] if bs <= 16 else 32 + i * 36 % 1360
        if bs == 32:
            tasks["y"] = np.repeat(32 + (np.arange(n_isl) % 40) * 34, K)
            tasks["x"] = np.tile(64 + np.arange(K) * bs, n_isl) + np.repeat((np.arange(n_isl) // 40) * (K * bs + 64), K)
        for i in range(n_isl):
            t = tasks[i * K:(i + 1) * K]
            rlo, rhi = int(t["y"].min()) >> 6, (int(t["y"].max()) + bs - 1) >> 6
            clo, chi = int(t["x"].min()) >> 6, (int(t["x"].max()) + bs - 1) >> 6
            isl[i]["reserved"] = rlo | (rhi << 8) | (clo << 16) | (chi << 24)
        d_t, d_i, d_w, d_c = ctx.alloc(tasks), ctx.alloc(isl), ctx.alloc(np.array(woff, np.int32)), ctx.alloc(coeffs)
        for it in range(3):
            ctx.intra_pred_islands(d_t, d_i, n_isl, d_w, d_c, frame)
        ctx.sync()
        ts = []
        for it in range(10):
            ctx.timer_begin(it)
            ctx.intra_pred_islands(d_t, d_i, n_isl, d_w, d_c, frame)
            ctx.timer_end(it)
        ctx.sync()
        ts = sorted(ctx.timer_read(i) for i in range(10))
        print(f"bs {bs:2d} mode {mode} eob {eob:4d}: {ts[5]*1e3:7.1f} us per launch = {ts[5]*1e3/K:5.2f} us per dependent block")
        for b in (d_t, d_i, d_w, d_c):
            b.free()
