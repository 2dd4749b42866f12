#!/usr/bin/env python3
"""K-idct / K-lpf: the transform+add and loop-filter kernels alone on one frame worth of
blocks.  K-idct: every transform block of a 1440p / 2160p frame coded, one size per run (and the 1:1:1:1
by-area mix), dense (eob = N*N) or DC-only; algorithmic bytes = 4*N*N (or 4) + 2*N*N*bps + 16 per block.
K-lpf: uniform random frame, level 32, masks of a random partition; 2*P*bps + 160 B per superblock."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as g
pkg = g.load_pkg()
import workload
ctx = pkg.Context(0)
rng = np.random.default_rng(11)
out = []

def timeit(fn, n=20):
    for _ in range(3):
        fn()
    ctx.sync()
    for i in range(n):
        ctx.timer_begin(i); fn(); ctx.timer_end(i)
    ctx.sync()
    return sorted(ctx.timer_read(i) for i in range(n))[n // 2]

for (W, H) in ((2560, 1440), (3840, 2160)):
    for bd in (8, 10):
        bps = 2 if bd > 8 else 1
        fr = pkg.DevFrame(ctx, W, H, bit_depth=bd)
        dt = np.uint16 if bd > 8 else np.uint8
        fr.upload([rng.integers(0, 1 << bd, (d[3], d[2])).astype(dt) for d in fr.dims])
        for label, sizes in (("4x4", (0,)), ("8x8", (1,)), ("16x16", (2,)), ("32x32", (3,)), ("mix", (0, 1, 2, 3))):
            for dense in (1, 0):
                recs = []
                for p in range(3):
                    pw, ph = fr.dims[p][2], fr.dims[p][3]
                    # by-area mix: split the plane into vertical bands, one per size
                    band = pw // len(sizes) // 32 * 32
                    for k, txs in enumerate(sizes):
                        n = 4 << txs
                        x_lo, x_hi = k * band, (pw if k == len(sizes) - 1 else (k + 1) * band)
                        xs, ys = np.meshgrid(np.arange(x_lo, x_hi - n + 1, n), np.arange(0, ph - n + 1, n))
                        t = np.zeros(xs.size, pkg.TXB_DTYPE)
                        t["x"], t["y"], t["plane"], t["tx_size"] = xs.ravel(), ys.ravel(), p, txs
                        t["eob"] = n * n if dense else 1
                        recs.append(t)
                tb = np.concatenate(recs)
                nn = (4 << tb["tx_size"].astype(np.int64)) ** 2
                tb["coeff_off"] = np.concatenate([[0], np.cumsum(nn)[:-1]])
                coeffs = rng.integers(-40, 41, int(nn.sum()) + 16).astype(np.int32)
                tb, counts = pkg.sort_txb_by_size(tb)
                d_t, d_c = ctx.alloc(tb), ctx.alloc(coeffs)
                ms = timeit(lambda: ctx.idct_add_batch(d_t, counts, d_c, fr))
                byts = int((np.where(tb["eob"] > 1, nn[np.argsort(np.argsort(tb["tx_size"], kind="stable"))] * 0 + (4 << tb["tx_size"].astype(np.int64)) ** 2 * 4, 4)).sum()
                           + (2 * (4 << tb["tx_size"].astype(np.int64)) ** 2 * bps).sum() + 16 * len(tb))
                rec = dict(kernel="idct_add", frame=f"{W}x{H}", bd=bd, size=label, dense=dense, blocks=int(len(tb)), ms=round(ms, 4),
                           algorithmic_MB=round(byts / 1e6, 2), GBps=round(byts / ms / 1e6, 1), frac_hbm_peak=round(byts / ms / 1e6 / 8000, 4))
                out.append(rec); print(rec, flush=True)
                d_t.free(); d_c.free()
        # K-lpf
        wl = workload.make_frame_workload(W, H, seed=17, bd=bd, level=32)
        d_lfm = ctx.alloc(wl["lfm"])
        th = pkg.LfThresh()
        mblim, lim, hev = wl["thresholds"]
        for i in range(64):
            th.mblim[i], th.lim[i], th.hev_thr[i] = int(mblim[i]), int(lim[i]), int(hev[i])
        ms = timeit(lambda: ctx.loop_filter_frame(d_lfm, wl["sb_rows"], wl["sb_cols"], th, fr, 3), n=10)
        P = sum(d[2] * d[3] for d in fr.dims)
        byts = 2 * P * bps + 160 * wl["sb_rows"] * wl["sb_cols"]
        rec = dict(kernel="loop_filter", frame=f"{W}x{H}", bd=bd, ms=round(ms, 4), algorithmic_MB=round(byts / 1e6, 2),
                   GBps=round(byts / ms / 1e6, 1), frac_hbm_peak=round(byts / ms / 1e6 / 8000, 5))
        out.append(rec); print(rec, flush=True)
        d_lfm.free(); fr.free()
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "kmicro.json"), "w"), indent=1)
