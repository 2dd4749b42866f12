#!/usr/bin/env python3
"""Per-task diagnosis of vp9hip_inter_pred_batch against the oracle (8-bit, unscaled): prints the
tasks whose destination block differs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as g
hip = g.load_pkg()
import vp9ref
from vp9ref import u8p
oracle = vp9ref.load_oracle(os.path.join(ROOT, "oracle", "liboracle.so"))
SIZES = [(4, 4), (4, 8), (8, 4), (8, 8), (8, 16), (16, 8), (16, 16), (16, 32), (32, 16), (32, 32), (32, 64), (64, 32), (64, 64)]
W, H = 328, 200
rng = np.random.default_rng(5)
ctx = hip.Context(0)
dst = hip.DevFrame(ctx, W, H, bit_depth=8)
refs, ref_planes = [], []
for k in range(2):
    fr = hip.DevFrame(ctx, W, H, bit_depth=8)
    planes = [rng.integers(0, 256, (d[3], d[2])).astype(np.uint8) for d in fr.dims]
    fr.upload(planes); refs.append(fr); ref_planes.append(planes)
dplanes = [rng.integers(0, 256, (d[3], d[2])).astype(np.uint8) for d in dst.dims]
dst.upload(dplanes)
expect = [p.copy() for p in dplanes]
tasks = []
SING = {}
for plane in range(3):
    aw, ah = dst.dims[plane][2], dst.dims[plane][3]
    occ = np.zeros((ah // 4 + 20, aw // 4 + 20), bool)
    for it in range(400):
        w, h = SIZES[int(rng.integers(0, len(SIZES)))]
        x = int(rng.integers(0, aw // 4)) * 4; y = int(rng.integers(0, ah // 4)) * 4
        if occ[y // 4:y // 4 + h // 4, x // 4:x // 4 + w // 4].any():
            continue
        occ[y // 4:y // 4 + h // 4, x // 4:x // 4 + w // 4] = True
        comp = it % 3 == 0
        filt = int(rng.integers(0, 4))
        t = np.zeros((), hip.INTER_DTYPE)
        t["dst_x"], t["dst_y"], t["w"], t["h"], t["plane"] = x, y, w, h, plane
        t["flags"] = (filt << 1) | int(comp)
        pad = np.zeros((ah + 64, aw + 64), np.uint8); pad[:ah, :aw] = expect[plane]
        blk = np.ascontiguousarray(pad[y:y + h, x:x + w])
        for r in range(2 if comp else 1):
            ri = int(rng.integers(0, 2))
            rw, rh = refs[ri].dims[plane][0], refs[ri].dims[plane][1]
            kind = it % 5
            if kind == 0:
                px = int(rng.integers(-200 * 16, (rw + 200) * 16)); py = int(rng.integers(-200 * 16, (rh + 200) * 16))
            elif kind == 1:
                px = int(rng.integers(-8, rw)) * 16; py = int(rng.integers(-8, rh)) * 16
            else:
                px = x * 16 + int(rng.integers(-64 * 16, 64 * 16)); py = y * 16 + int(rng.integers(-64 * 16, 64 * 16))
            if kind == 2: px &= ~15
            if kind == 3: py &= ~15
            t["pos_x"][r], t["pos_y"][r], t["ref"][r] = px, py, ri
            t["step_x"][r] = t["step_y"][r] = 16
            rp = ref_planes[ri][plane]
            oracle.vp9o_inter_predict_block(u8p(rp), rp.shape[1], rw, rh, px, py, 16, 16, filt, w, h, u8p(blk), w, r)
        pad[y:y + h, x:x + w] = blk
        expect[plane] = pad[:ah, :aw].copy()
        tasks.append(t)
        if comp:
            singles = []
            for r in range(2):
                b1 = np.zeros((h, w), np.uint8)
                rp = ref_planes[int(t["ref"][r])][plane]
                rw, rh = refs[int(t["ref"][r])].dims[plane][0], refs[int(t["ref"][r])].dims[plane][1]
                oracle.vp9o_inter_predict_block(u8p(rp), rp.shape[1], rw, rh, int(t["pos_x"][r]), int(t["pos_y"][r]), 16, 16, filt, w, h, u8p(b1), w, 0)
                singles.append(b1)
            SING[(plane, x, y)] = singles
tasks = np.array(tasks, dtype=hip.INTER_DTYPE)
tasks, counts = hip.sort_inter_tasks(tasks, False)
d_tasks = ctx.alloc(tasks)
ctx.inter_pred_batch(d_tasks, counts, refs, dst); ctx.sync()
got = dst.download()
nbad = 0
for t in tasks:
    p = int(t["plane"]); aw, ah = dst.dims[p][2], dst.dims[p][3]
    x, y, w, h = int(t["dst_x"]), int(t["dst_y"]), int(t["w"]), int(t["h"])
    a = got[p][y:min(y + h, ah), x:min(x + w, aw)]; b = expect[p][y:min(y + h, ah), x:min(x + w, aw)]
    if (a != b).any():
        nbad += 1
        if nbad <= 3 and (p, x, y) in SING:
            print("got"); print(a); print("expect"); print(b); print("p0"); print(SING[(p,x,y)][0][:a.shape[0], :a.shape[1]]); print("p1"); print(SING[(p,x,y)][1][:a.shape[0], :a.shape[1]])
        if nbad <= 12:
            bad = np.argwhere(a != b)
            print(f"plane {p} {w}x{h} at ({x},{y}) flags {int(t['flags']):#x} pos {t['pos_x']} {t['pos_y']} sub ({t['pos_x'][0]&15},{t['pos_y'][0]&15}) "
                  f"nbad {len(bad)} rows {sorted(set(bad[:,0]))[:10]} cols {sorted(set(bad[:,1]))[:10]} diff {(a.astype(int)-b)[tuple(bad[0])]}")
print("tasks", len(tasks), "bad", nbad, "counts", counts)
