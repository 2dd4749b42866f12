#!/usr/bin/env python3
"""Where a wave of the 8-bit register convolve (inter_reg_kernel) spends its life: per-wave stamps from the probe build
(tools/build_probe_lib.sh).  Bench frame (2560x1440 8-bit, blockgen partition), or `k64` = one 2160p frame of 64x64 blocks.
    python tools/conv_stamps.py [bench|k64]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as g
hip = g.load_pkg()
hip.LIB_PATH = os.path.join(ROOT, "tools", "build", "libvp9hip_stamps.so")
import blockgen, workload
mode = sys.argv[1] if len(sys.argv) > 1 else "bench"
SHAPES = [(4, 4), (4, 8), (8, 4), (8, 8), (8, 16), (16, 8), (16, 16), (16, 32), (32, 16), (32, 32), (32, 64), (64, 32), (64, 64)]


def wgs(counts):
    out = []
    for (w, h), n in zip(SHAPES, counts[:13]):
        L, SH = w // 4, (4 if h == 4 else 8)
        per_wg = 256 // L
        out.append((int(n) * (h // SH) + per_wg - 1) // per_wg)
    return out


ctx = hip.Context(0)
rng = np.random.default_rng(1440)
if mode == "bench":
    W, H, bd = 2560, 1440, 8
    blocks = blockgen.gen_blocks(rng, W, H, hip.BLOCK_DTYPE, intra_frac=0.08, skip_frac=0.35)
    coef, eob = blockgen.gen_coeffs(rng, blocks, W, H, bd)
    P = hip.FrameParams()
    P.width, P.height, P.ss_x, P.ss_y, P.bit_depth, P.hbd, P.log2_tile_cols, P.build_lf_masks = W, H, 1, 1, bd, 0, 2, 1
    for k in range(3):
        P.ref_width[k], P.ref_height[k] = W, H
    L = hip.Packer().pack(P, blocks, eob)
    tasks, counts = L["inter_tasks"], np.asarray(L["inter_class_count"])
else:
    W, H, bd = 3840, 2160, 8
    parts = []
    for p in range(3):
        ss = 1 if p else 0
        b = 64 >> ss
        xs, ys = np.meshgrid(np.arange(0, W >> ss, b), np.arange(0, H >> ss, b))
        n = xs.size
        t = np.zeros(n, hip.INTER_DTYPE)
        t["dst_x"], t["dst_y"], t["w"], t["h"], t["plane"] = xs.ravel(), ys.ravel(), b, b, p
        t["flags"] = rng.integers(0, 3, n) << 1
        for r in range(2):
            mv = rng.integers(-512, 513, (n, 2))
            t["pos_x"][:, r] = (t["dst_x"].astype(np.int32) << 4) + mv[:, 0] * (2 >> ss)
            t["pos_y"][:, r] = (t["dst_y"].astype(np.int32) << 4) + mv[:, 1] * (2 >> ss)
            t["ref"][:, r] = rng.integers(0, 3, n)
            t["step_x"][:, r] = t["step_y"][:, r] = 16
        parts.append(t)
    tasks, counts = hip.sort_inter_tasks(np.concatenate(parts), False)
    counts = np.asarray(counts)
refs = []
for r in range(3):
    fr = hip.DevFrame(ctx, W, H, bit_depth=bd)
    fr.upload([np.ascontiguousarray(workload.smooth_noise(rng, d[3], d[2], bd, sigma=1.5 + r).astype(np.uint8)) for d in fr.dims])
    refs.append(fr)
dst = hip.DevFrame(ctx, W, H, bit_depth=bd)
d_t = ctx.alloc(tasks)
for i in range(5):
    ctx.inter_pred_batch(d_t, counts, refs, dst)
ctx.sync()
ctx.timer_begin(0); ctx.inter_pred_batch(d_t, counts, refs, dst); ctx.timer_end(0); ctx.sync()
print(f"{mode}: {len(tasks)} tasks, launch {ctx.timer_read(0) * 1e3:.1f} us (probe build: a drain before the row pass)")
w = wgs(counts)
n_waves = min(4 * sum(w), 16384)
st = np.zeros((n_waves, 8), np.int64)
assert hip.lib().vp9hip_debug_conv_stamps(st.ctypes.data_as(ctypes.c_void_p), n_waves) == 0
ok = st[:, 7] > 0
t0 = st[ok, 0].min()
start, end = (st[:, 0] - t0) / 100.0, (st[:, 7] - t0) / 100.0  # us
print(f"{ok.sum()} of {n_waves} waves stamped; first start 0, last start {start[ok].max():.1f} us, last end {end[ok].max():.1f} us")
print("waves started by 1/2/4/8/12/16 us:", [int((start[ok] <= x).sum()) for x in (1, 2, 4, 8, 12, 16)])
print("waves ended by 4/8/12/16/20/24 us:", [int((end[ok] <= x).sum()) for x in (4, 8, 12, 16, 20, 24)])
base = 0
for (sw, sh), nw, cnt in zip(SHAPES, w, counts[:13]):
    a, b = 4 * base, min(4 * (base + nw), n_waves)
    base += nw
    if nw == 0 or a >= n_waves:
        continue
    s = st[a:b]
    s = s[(s[:, 7] > 0) & (s[:, 3] > 0)]
    if len(s) == 0:
        continue
    d = lambda i, j: float(np.mean(s[:, j] - s[:, i]))
    life = float(np.mean(s[:, 7] - s[:, 0])) / 100.0
    print(f"{sw:2d}x{sh:<2d} {int(cnt):6d} tasks {nw * 4:5d} waves: life {life:5.2f} us | cycles: task record {d(1, 2):6.0f}, window {d(2, 3):6.0f}, rows {d(3, 4):6.0f}, "
          f"columns+store {d(4, 5):6.0f}, rest {d(5, 6):6.0f}; start {np.mean(s[:, 0] - t0) / 100:5.1f} us")
