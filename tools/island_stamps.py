#!/usr/bin/env python3
"""Where the time of the fused island walk + loop filter launch goes: wall-clock stamps (100 MHz) of every workgroup's
stages, from a probe build of the library (tools/build/libvp9hip_stamps.so: lf_kernels.hip compiled with
-DVP9HIP_STAMPS).  Bench frame (2560x1440 8-bit, blockgen partition) through vp9hip_decoder_run.
    python tools/island_stamps.py [bit depth]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as g

hip = g.load_pkg()
hip.LIB_PATH = os.path.join(ROOT, "tools", "build", "libvp9hip_stamps.so")
import blockgen
import workload

W, H = 2560, 1440
bd = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rng = np.random.default_rng(1440)
dt = np.uint16 if bd > 8 else np.uint8
aw, ah = (W + 7) & ~7, (H + 7) & ~7
dims = [(aw, ah), (aw // 2, ah // 2), (aw // 2, ah // 2)]
refs = [[np.ascontiguousarray(workload.smooth_noise(rng, d[1], d[0], bd, sigma=1.5 + k).astype(dt)) for d in dims] for k in range(3)]
blocks = blockgen.gen_blocks(rng, W, H, hip.BLOCK_DTYPE, intra_frac=0.08, skip_frac=0.35)
coef, eob = blockgen.gen_coeffs(rng, blocks, W, H, bd)
P = hip.FrameParams()
P.width, P.height, P.ss_x, P.ss_y, P.bit_depth, P.hbd, P.log2_tile_cols, P.build_lf_masks = W, H, 1, 1, bd, int(bd > 8), 2, 1
for k in range(3):
    P.ref_width[k], P.ref_height[k] = W, H
th = hip.LfThresh()
hip.lib().vp9hip_lf_frame_init(32, 0, None, None, 0, 0, None, None, None, ctypes.byref(th))
pk = hip.Packer()
L = pk.pack(P, blocks, eob)
n_lds, row_pos, sb_rows, sb_cols = L["n_islands_lds"], L["island_row_pos"], L["sb_rows"], L["sb_cols"]
isl, woff = L["intra_islands"], L["intra_island_wave_off"]
print(f"{len(isl)} islands ({n_lds} in LDS), {len(L['intra_island_tasks'])} island tasks, {L['n_waves']} waves deep; row_pos {row_pos.tolist()}")
dec = hip.Decoder(0)
for k in range(3):
    dec.upload(k, refs[k], W, H, bd)
dec.alloc_slot(3, W, H, bd)
dec.begin_frame(P, blocks, eob, coef)
ALL = hip.PHASE_INTER | hip.PHASE_INTRA | hip.PHASE_LF
for _ in range(6):
    dec.run(ALL, (0, 1, 2), 3, thresh=th)
dec.sync()
n_isl = len(isl)
grid = n_isl + 3 * sb_rows
st = np.zeros((grid, 8), np.int64)
assert hip.lib().vp9hip_debug_stamps(st.ctypes.data_as(ctypes.c_void_p), grid) == 0
print(f"last run {dec.last_run_ms():.3f} ms GPU")
role = np.full(grid, 1 << 30)  # island index, or -(1 + 3 * row + plane)
for r in range(sb_rows):
    for p in range(3):
        role[row_pos[r] + 3 * r + p] = -(1 + 3 * r + p)
k = 0
for b in range(grid):
    if role[b] == 1 << 30:
        role[b] = k
        k += 1
assert k == n_isl
t0 = st[:, 0].min()
us = lambda x: (x - t0) / 100.0
is_isl = (role >= 0) & (st[:, 6] > st[:, 0])  # (islands walked through memory carry no stage stamps)
print('islands walked in LDS:', int(is_isl.sum()), 'of', n_isl)
S = st[is_isl]
names = ["tasks+boxes", "window", "residual", "waves", "write-back", "marks"]
dur = np.diff(S[:, :7], axis=1) / 100.0
nw = isl["n_waves"][role[is_isl]]
nt = np.array([woff[r["wave_off_start"] + r["n_waves"]] for r in isl])[role[is_isl]]
print("island stages, us: mean / median / max")
for j, n in enumerate(names):
    print(f"  {n:12s} {dur[:, j].mean():7.2f} {np.median(dur[:, j]):7.2f} {dur[:, j].max():7.2f}")
tot = (S[:, 6] - S[:, 0]) / 100.0
print(f"  whole island {tot.mean():7.2f} {np.median(tot):7.2f} {tot.max():7.2f};  tasks per island mean {nt.mean():.1f} max {nt.max()},  waves mean {nw.mean():.1f} max {nw.max()}")
print(f"  per wave of the walk: {(dur[:, 3] / np.maximum(nw, 1)).mean():.3f} us mean;  deepest island: {nw.max()} waves, walk {dur[nw.argmax(), 3]:.1f} us, whole {tot[nw.argmax()]:.1f} us")
print(f"island starts: first {us(S[:, 0]).min():.1f}, median {np.median(us(S[:, 0])):.1f}, last {us(S[:, 0]).max():.1f} us; last island done at {us(S[:, 6]).max():.1f} us")
R = st[role < 0]
rr = -role[role < 0] - 1
print("filter rows (luma): row: start, first superblock ready, end (us after the launch's first stamp)")
for r in range(sb_rows):
    i = np.flatnonzero(rr == 3 * r)[0]
    print(f"  row {r:2d}: {us(R[i, 0]):7.1f} {us(R[i, 1]):7.1f} {us(R[i, 7]):7.1f}")
print("filter rows (luma), the filtering wave's time per superblock step, us: vertical pass | wait at the barrier behind it | horizontal pass | "
      "wait at the barrier behind it | (wave 1's work beside the horizontal pass: gate, next interior from memory into LDS, controls)")
for r in range(sb_rows):
    i = np.flatnonzero(rr == 3 * r)[0]
    a = R[i, 2:7] / 100.0 / sb_cols
    print(f"  row {r:2d}: {a[0]:5.2f} | {a[1]:5.2f} | {a[2]:5.2f} | {a[3]:5.2f} | ({a[4]:5.2f})   sum {a[:4].sum():5.2f}")
print(f"launch span: {max(us(S[:, 6]).max(), us(R[:, 7]).max()):.1f} us")
dec.close()
pk.close()
