"""workload.py — TEST INFRASTRUCTURE: synthetic frame workloads + a Python packing of the work lists
(predates the C packer; kept for the kernel-level tests, which want lists they control).

Produces, for one frame, exactly what libvpx's entropy stage would hand to the reconstruction
path: a partition into prediction blocks (modes, references, motion vectors, transform sizes,
skip flags, filter levels) and dequantised coefficients, and packs them into the work lists of
include/vp9hip.h (inter tasks, transform-block records, wave-ordered intra tasks, loop-filter
masks).  The packing rules restate the reference's host logic:

  * transform-block visit order and frame-edge clipping: vp9_foreach_transformed_block_in_plane
    (vp9/common/vp9_blockd.c) / decode_block (vp9/decoder/vp9_decodeframe.c:1198-1242)
  * uv transform size: uv_txsize_lookup (vp9/common/vp9_common_data.c)
  * tx_type from the intra mode: intra_mode_to_tx_type_lookup (vp9_reconintra.c:24-35)
  * have_top / have_left / have_right: vp9_predict_intra_block (vp9_reconintra.c:404-424)
  * loop-filter masks: vp9_build_mask + vp9_adjust_mask (vp9_loopfilter.c:1528-1608, 766-860)
  * thresholds: update_sharpness / vp9_loop_filter_init (vp9_loopfilter.c:212-250)

There is no pixel arithmetic here.  Real VP9 clips (Bravia.1440.ivf, FoodMarket2.ivf) are not
in the reference snapshot and no encoder is available, so the data is synthetic and seeded.
"""
import numpy as np

from cuda_vp9_amd import INTER_DTYPE, INTRA_DTYPE, LFM_DTYPE, TXB_DTYPE

# intra_mode_to_tx_type_lookup (vp9_reconintra.c:24-35): DC V H D45 D135 D117 D153 D207 D63 TM
MODE_TO_TX_TYPE = np.array([0, 1, 2, 0, 3, 1, 2, 2, 1, 3], np.uint8)


def smooth_noise(rng, h, w, bd, sigma=2.0):
    """Band-limited noise texture (box-blurred uniform noise), full range of the bit depth."""
    img = rng.random((h + 16, w + 16))
    k = int(max(1, round(sigma * 2)))
    c = np.cumsum(np.cumsum(img, 0), 1)
    blur = (c[k:, k:] - c[:-k, k:] - c[k:, :-k] + c[:-k, :-k]) / (k * k)
    blur = blur[:h, :w]
    blur = (blur - blur.min()) / max(blur.max() - blur.min(), 1e-9)
    return np.clip(blur * ((1 << bd) - 1), 0, (1 << bd) - 1).astype(np.uint16 if bd > 8 else np.uint8)


def quad_partition(rng, aw, ah, p_split=(0.85, 0.6, 0.45), min_log2=3):
    """Square blocks in VP9 decode order: 64x64 raster, recursive quad inside."""
    out = []

    def rec(x, y, lg):
        if x >= aw or y >= ah:
            return
        s = 1 << lg
        if lg > min_log2 and (rng.random() < p_split[6 - lg] or x + s > aw or y + s > ah):
            h = s >> 1
            rec(x, y, lg - 1)
            rec(x + h, y, lg - 1)
            rec(x, y + h, lg - 1)
            rec(x + h, y + h, lg - 1)
        else:
            out.append((x, y, s))

    for sy in range(0, ah, 64):
        for sx in range(0, aw, 64):
            rec(sx, sy, 6)
    return np.array(out, np.int32)


def lf_thresholds(sharpness=0):
    mblim, lim, hev = np.zeros(64, np.uint8), np.zeros(64, np.uint8), np.zeros(64, np.uint8)
    for lvl in range(64):
        bil = lvl >> ((sharpness > 0) + (sharpness > 4))
        if sharpness > 0 and bil > 9 - sharpness:
            bil = 9 - sharpness
        bil = max(bil, 1)
        lim[lvl], mblim[lvl], hev[lvl] = bil, 2 * (lvl + 2) + bil, lvl >> 4
    return mblim, lim, hev


def build_lf_masks(blocks, aw, ah):
    """LOOP_FILTER_MASK per 64x64 superblock from the block list.
    blocks: structured array with x, y, size, tx (log2 of luma tx), level, skip, inter."""
    mi_rows, mi_cols = ah // 8, aw // 8
    sb_rows, sb_cols = (mi_rows + 7) // 8, (mi_cols + 7) // 8
    n = sb_rows * sb_cols
    left_y = np.zeros((n, 4), np.uint64)
    above_y = np.zeros((n, 4), np.uint64)
    int_y = np.zeros(n, np.uint64)
    left_uv = np.zeros((n, 4), np.uint32)
    above_uv = np.zeros((n, 4), np.uint32)
    int_uv = np.zeros(n, np.uint32)
    lfl = np.zeros((n, 64), np.uint8)
    M64 = (1 << 64) - 1
    for b in blocks:
        level = int(b["level"])
        if level == 0:
            continue
        x, y, s = int(b["x"]), int(b["y"]), int(b["size"])
        mi_row, mi_col = y >> 3, x >> 3
        sb = (mi_row >> 3) * sb_cols + (mi_col >> 3)
        r0, c0 = mi_row & 7, mi_col & 7
        w8 = h8 = max(1, s >> 3)
        txy = int(b["tx"]) - 2
        # uv transform size (uv_txsize_lookup, 4:2:0): largest square <= min(tx, block/2), >= 4x4
        txuv = max(0, min(txy, int(np.log2(max(s >> 1, 4))) - 2))
        shift_y, shift_uv = c0 + (r0 << 3), (c0 >> 1) + ((r0 >> 1) << 2)
        build_uv = (r0 & 1) == 0 and (c0 & 1) == 0
        for r in range(h8):
            lfl[sb, (r0 + r) * 8 + c0:(r0 + r) * 8 + c0 + w8] = level
        size_mask = sum(((1 << w8) - 1) << (8 * r) for r in range(h8))
        wuv, huv = max(1, w8 >> 1), max(1, h8 >> 1)
        size_mask_uv = sum(((1 << wuv) - 1) << (4 * r) for r in range(huv))
        above_pred, left_pred = (1 << w8) - 1, sum(1 << (8 * r) for r in range(h8))
        above_pred_uv, left_pred_uv = (1 << wuv) - 1, sum(1 << (4 * r) for r in range(huv))
        above_y[sb, txy] |= np.uint64((above_pred << shift_y) & M64)
        left_y[sb, txy] |= np.uint64((left_pred << shift_y) & M64)
        if build_uv:
            above_uv[sb, txuv] |= (above_pred_uv << shift_uv) & 0xffff
            left_uv[sb, txuv] |= (left_pred_uv << shift_uv) & 0xffff
        if b["skip"] and b["inter"]:
            continue
        t8 = max(1, (4 << txy) >> 3)          # luma transform size in 8-px units
        rows_on = sum(0xff << (8 * r) for r in range(0, 8, t8))
        cols_on = sum(0x0101010101010101 << c for c in range(0, 8, t8))
        above_y[sb, txy] |= np.uint64(((size_mask << shift_y) & rows_on) & M64)
        left_y[sb, txy] |= np.uint64(((size_mask << shift_y) & cols_on) & M64)
        if build_uv:
            tu = max(1, (4 << txuv) >> 3)
            rows_uv = sum(0xf << (4 * r) for r in range(0, 4, tu))
            cols_uv = sum(0x1111 << c for c in range(0, 4, tu))
            above_uv[sb, txuv] |= ((size_mask_uv << shift_uv) & rows_uv) & 0xffff
            left_uv[sb, txuv] |= ((size_mask_uv << shift_uv) & cols_uv) & 0xffff
        if txy == 0:
            int_y[sb] |= np.uint64((size_mask << shift_y) & M64)
        if build_uv and txuv == 0:
            int_uv[sb] |= (size_mask_uv << shift_uv) & 0xffff
    # vp9_adjust_mask (vp9_loopfilter.c:766-860)
    out = np.zeros(n, LFM_DTYPE)
    LB, AB = 0x1111111111111111, 0x000000ff000000ff  # left_border, above_border (vp9_loopfilter.c)
    for sr in range(sb_rows):
        for sc in range(sb_cols):
            i = sr * sb_cols + sc
            ly = [int(v) for v in left_y[i]]
            ay = [int(v) for v in above_y[i]]
            luv = [int(v) for v in left_uv[i]]
            auv = [int(v) for v in above_uv[i]]
            iy, iuv = int(int_y[i]), int(int_uv[i])
            ly[2] |= ly[3]
            ay[2] |= ay[3]
            luv[2] |= luv[3]
            auv[2] |= auv[3]
            ly[1] |= ly[0] & LB
            ly[0] &= ~LB
            ay[1] |= ay[0] & AB
            ay[0] &= ~AB
            luv[1] |= luv[0] & 0x1111
            luv[0] &= ~0x1111
            auv[1] |= auv[0] & 0x000f
            auv[0] &= ~0x000f
            rows, cols = mi_rows - sr * 8, mi_cols - sc * 8
            if rows < 8:
                my = (1 << (rows << 3)) - 1
                muv = (1 << (((rows + 1) >> 1) << 2)) - 1
                for k in range(3):
                    ly[k] &= my
                    ay[k] &= my
                    luv[k] &= muv
                    auv[k] &= muv
                iy &= my
                iuv &= muv
                if rows == 1:
                    auv[1] |= auv[2]
                    auv[2] = 0
                if rows == 5:
                    auv[1] |= auv[2] & 0xff00
                    auv[2] &= ~(auv[2] & 0xff00)
            if cols < 8:
                my = ((1 << cols) - 1) * 0x0101010101010101
                muv = ((1 << ((cols + 1) >> 1)) - 1) * 0x1111
                muv_int = ((1 << (cols >> 1)) - 1) * 0x1111
                for k in range(3):
                    ly[k] &= my
                    ay[k] &= my
                    luv[k] &= muv
                    auv[k] &= muv
                iy &= my
                iuv &= muv_int
                if cols == 1:
                    luv[1] |= luv[2]
                    luv[2] = 0
                if cols == 5:
                    luv[1] |= luv[2] & 0xcccc
                    luv[2] &= ~(luv[2] & 0xcccc)
            if sc == 0:
                for k in range(3):
                    ly[k] &= 0xfefefefefefefefe
                    luv[k] &= 0xeeee
            m = out[i]
            for k in range(3):
                m["left_y"][k], m["above_y"][k] = ly[k] & M64, ay[k] & M64
                m["left_uv"][k], m["above_uv"][k] = luv[k] & 0xffff, auv[k] & 0xffff
            m["int_4x4_y"], m["int_4x4_uv"] = iy & M64, iuv & 0xffff
            m["lfl_y"] = lfl[i]
    return out, sb_rows, sb_cols


def intra_levels(tasks, dims, want_components=False):
    """Dependency level (1-based) of each intra task, tasks given in decode order.  A block
    depends on the blocks that own its left column, above row (2*bs wide only for 4x4 blocks
    with have_right) and above-left sample; already-reconstructed inter area is level 0.
    With want_components also returns the connected component (island) id of every task."""
    maps = [np.zeros((ah // 4 + 2, aw // 4 + 2), np.int32) for (aw, ah) in dims]
    owner = [np.full((ah // 4 + 2, aw // 4 + 2), -1, np.int64) for (aw, ah) in dims] if want_components else None
    parent = list(range(len(tasks)))

    def find(a):
        while parent[a] != a:
            parent[a] = parent[parent[a]]
            a = parent[a]
        return a

    lv = np.zeros(len(tasks), np.int32)
    px, py, pp = tasks["x"] // 4, tasks["y"] // 4, tasks["plane"]
    nn = 1 << tasks["tx_size"].astype(np.int32)
    fl = tasks["flags"]
    for i in range(len(tasks)):
        m = maps[pp[i]]
        aw, ah = dims[pp[i]]
        cx, cy, n = int(px[i]), int(py[i]), int(nn[i])
        xmax, ymax = aw // 4, ah // 4
        l = 0
        deps = []
        if fl[i] & 2:
            l = max(l, int(m[cy:min(cy + n, ymax), cx - 1].max()))
            if want_components:
                deps.append(owner[pp[i]][cy:min(cy + n, ymax), cx - 1])
        if fl[i] & 1:
            ext = 2 * n if (n == 1 and (fl[i] & 4)) else n
            l = max(l, int(m[cy - 1, cx:min(cx + ext, xmax)].max()))
            if want_components:
                deps.append(owner[pp[i]][cy - 1, cx:min(cx + ext, xmax)])
            if fl[i] & 2:
                l = max(l, int(m[cy - 1, cx - 1]))
                if want_components:
                    deps.append(owner[pp[i]][cy - 1, cx - 1:cx])
        l += 1
        m[cy:cy + n, cx:cx + n] = l
        lv[i] = l
        if want_components:
            for d in deps:
                for o in np.unique(d):
                    if o >= 0:
                        ra, rb = find(int(o)), find(i)
                        if ra != rb:
                            parent[rb] = ra
            owner[pp[i]][cy:cy + n, cx:cx + n] = i
    if want_components:
        comp = np.array([find(i) for i in range(len(tasks))], np.int64)
        return lv, comp
    return lv


def island_pitch(w):
    """VP9HIP_ISLAND_PITCH (include/vp9hip.h): row pitch of a plane window whose blocks span w samples."""
    p = (w + 6) & ~1
    return p if (p & 2) else p + 2


ISLAND_TILE_ELEMS, ISLAND_MAX_TASKS, ISLAND_MAX_TX32 = 20480, 768, 16


def island_fits(t):
    """VP9HIP_ISLAND_FITS for the tasks of one island: does its sample window fit the LDS of a workgroup?"""
    if len(t) > ISLAND_MAX_TASKS or int(((t["tx_size"] == 3) & (t["eob"] > 1)).sum()) > ISLAND_MAX_TX32:
        return False
    elems = 0
    bs = 4 << t["tx_size"].astype(np.int64)
    for p in range(3):
        m = t["plane"] == p
        if not m.any():
            continue
        x0, y0 = int(t["x"][m].min()), int(t["y"][m].min())
        x1, y1 = int((t["x"][m] + bs[m]).max()), int((t["y"][m] + bs[m]).max())
        elems += island_pitch(x1 - x0) * (y1 - y0 + 1)
    return elems <= ISLAND_TILE_ELEMS


def island_sb_expected(isl_tasks, islands, sb_rows, sb_cols, wave_off=None):
    """Marks, per island, the LAST task (list order = wave order) inside each luma superblock with bit 0 of
    `reserved` — when the island is done with that superblock the island kernel reports it to the loop filter
    running beside it — and returns expected[r * sb_cols + c] = number of marks in superblock (r, c)
    (vp9hip_intra_islands_lf; mirrors vp9hip_pack.c)."""
    exp = np.zeros(sb_rows * sb_cols, np.int32)
    isl_tasks["reserved"] = 0
    if wave_off is None:
        starts = np.sort(islands["task_start"].astype(np.int64)) if len(islands) else np.zeros(0, np.int64)
        end_of = dict(zip(starts.tolist(), np.r_[starts[1:], len(isl_tasks)].tolist()))
    for r in islands:
        a = int(r["task_start"])
        b = a + int(wave_off[r["wave_off_start"] + r["n_waves"]]) if wave_off is not None else end_of[a]
        t = isl_tasks[a:b]
        sc = np.where(t["plane"] > 0, 1, 0)
        sb = np.minimum((t["y"].astype(np.int64) << sc) >> 6, sb_rows - 1) * sb_cols + \
             np.minimum((t["x"].astype(np.int64) << sc) >> 6, sb_cols - 1)
        # last occurrence of each superblock in list order
        rev = sb[::-1]
        _, first_in_rev = np.unique(rev, return_index=True)
        last = len(sb) - 1 - first_in_rev
        isl_tasks["reserved"][a + last] = 1
        np.add.at(exp, sb[last], 1)
    return exp


def pack_intra_islands(tasks, levels, comp, max_island_tasks=4096, sb_rows=None):
    """Split the intra tasks into islands (one workgroup each, vp9hip_intra_pred_islands) and a
    remainder of very large components (too many blocks, or a sample window that does not fit the LDS of a
    workgroup) that keeps the per-wave launches.  Islands are ordered by group g = max(first superblock row - 1, 0)
    (the order vp9hip_intra_islands_lf wants); n_lds = all of them; returns (island tasks, islands, wave offsets, big tasks, big wave starts, n_lds, row_pos)."""
    from cuda_vp9_amd import ISLAND_DTYPE
    n = len(tasks)
    if n == 0:
        return (tasks, np.zeros(0, ISLAND_DTYPE), np.zeros(1, np.int32), tasks, np.zeros(1, np.int32), 0,
                np.zeros(sb_rows or 0, np.int32))
    ids, inv, counts = np.unique(comp, return_inverse=True, return_counts=True)
    # components that do not fit the LDS window of a workgroup (VP9HIP_ISLAND_FITS) go to the global waves too
    by_comp = np.argsort(inv, kind="stable")
    ends_c = np.cumsum(counts)
    fit_c = np.array([island_fits(tasks[by_comp[e - c:e]]) for c, e in zip(counts, ends_c)], bool)
    big = (counts[inv] > max_island_tasks) | ~fit_c[inv]
    # islands: sort by (component, level)
    idx = np.flatnonzero(~big)
    order = idx[np.lexsort((levels[idx], inv[idx]))]
    isl_tasks = tasks[order]
    islands, wave_off, fits = [], [], []
    if len(order):
        c_sorted, l_sorted = inv[order], levels[order]
        starts = np.flatnonzero(np.r_[True, c_sorted[1:] != c_sorted[:-1]])
        ends = np.r_[starts[1:], len(order)]
        for a, b in zip(starts, ends):
            lv = l_sorted[a:b]
            w = np.flatnonzero(np.r_[True, lv[1:] != lv[:-1]])
            t = isl_tasks[a:b]
            sc = np.where(t["plane"] > 0, 1, 0)
            bsz = 4 << t["tx_size"].astype(np.int64)
            rlo = int((t["y"].astype(np.int64) << sc).min()) >> 6
            rhi = (int(((t["y"].astype(np.int64) + bsz) << sc).max()) - 1) >> 6
            clo = int((t["x"].astype(np.int64) << sc).min()) >> 6
            chi = (int(((t["x"].astype(np.int64) + bsz) << sc).max()) - 1) >> 6
            islands.append((a, len(wave_off), len(w), rlo | (rhi << 8) | (clo << 16) | (chi << 24)))
            fits.append(island_fits(t))
            wave_off.extend(w.tolist())
            wave_off.append(b - a)
    islands = np.array(islands, dtype=ISLAND_DTYPE) if islands else np.zeros(0, ISLAND_DTYPE)
    fits = np.array(fits, bool)
    # by group (the order vp9hip_intra_islands_lf wants); inside a group deepest first: one workgroup walks an
    # island's waves in sequence, so the deepest island is the critical path
    grp = np.maximum((islands["reserved"] & 255).astype(np.int64) - 1, 0)
    islands = islands[np.lexsort((-islands["n_waves"].astype(np.int64), grp))]
    n_lds = int(fits.sum())
    row_pos = None
    if sb_rows is not None:
        g = np.maximum((islands["reserved"] & 255).astype(np.int64) - 1, 0)
        row_pos = np.cumsum(np.bincount(np.minimum(g, sb_rows - 1), minlength=sb_rows)).astype(np.int32)
    wave_off = np.array(wave_off if wave_off else [0], np.int32)
    # remainder: global waves
    idx = np.flatnonzero(big)
    order = idx[np.argsort(levels[idx], kind="stable")]
    big_tasks = tasks[order]
    if len(order):
        lv = levels[order]
        nw = int(lv.max())
        wave_start = np.searchsorted(lv, np.arange(1, nw + 2)).astype(np.int32)
    else:
        wave_start = np.zeros(1, np.int32)
    return isl_tasks, islands, wave_off, big_tasks, wave_start, n_lds, row_pos


def make_frame_workload(width, height, seed=1440, bd=8, intra_frac=0.08, skip_frac=0.35, compound_frac=0.15,
                        level=28, sharpness=0, n_refs=3, all_intra=False, coef_amp=1.0):
    """One synthetic frame.  Returns a dict of host arrays (see keys at the end)."""
    rng = np.random.default_rng(seed)
    hbd = bd > 8
    aw, ah = (width + 7) & ~7, (height + 7) & ~7
    dims = [(aw, ah), (aw // 2, ah // 2), (aw // 2, ah // 2)]
    crop = [(width, height), ((width + 1) // 2, (height + 1) // 2), ((width + 1) // 2, (height + 1) // 2)]
    part = quad_partition(rng, aw, ah)
    nb = len(part)
    bx, by, bs = part[:, 0], part[:, 1], part[:, 2]
    lg = np.log2(bs).astype(np.int32)
    # intra blocks cluster: whole superblocks turn intra
    sb_id = (by >> 6) * ((aw + 63) >> 6) + (bx >> 6)
    sb_intra = rng.random(sb_id.max() + 1) < (1.0 if all_intra else intra_frac * 0.7)
    inter = ~(sb_intra[sb_id] | (rng.random(nb) < (1.0 if all_intra else intra_frac * 0.3)))
    skip = rng.random(nb) < skip_frac
    tx = np.minimum(rng.integers(2, 6, nb), np.minimum(lg, 5)).astype(np.int32)   # log2 of luma tx
    blocks = np.zeros(nb, [("x", "i4"), ("y", "i4"), ("size", "i4"), ("tx", "i4"), ("level", "i4"),
                           ("skip", "?"), ("inter", "?")])
    blocks["x"], blocks["y"], blocks["size"], blocks["tx"] = bx, by, bs, tx
    blocks["skip"], blocks["inter"] = skip, inter
    lvl = np.full(nb, level, np.int32)
    lvl[rng.random(nb) < 0.1] = max(level - 12, 1)
    lvl[rng.random(nb) < 0.03] = 0
    blocks["level"] = lvl

    # ---- inter tasks ------------------------------------------------------------------------
    ib = np.flatnonzero(inter)
    ni = len(ib)
    comp = rng.random(ni) < compound_frac
    filt = rng.integers(0, 3, ni)
    # global pan + per-block jitter, 1/8-pel luma units
    mv = np.stack([rng.integers(-48, 49, (ni, 2)) + 13, rng.integers(-48, 49, (ni, 2)) - 21], -1)  # [ni, ref, (row,col)]
    mv[rng.random(ni) < 0.08] = 0                       # some zero-motion blocks
    mv[rng.random(ni) < 0.05, :, 0] &= ~7               # full-pel rows
    mv[rng.random(ni) < 0.05, :, 1] &= ~7
    refs_idx = np.stack([rng.integers(0, n_refs, ni), rng.integers(0, n_refs, ni)], -1)
    tasks = np.zeros(ni * 3, INTER_DTYPE)
    for p in range(3):
        ss = 1 if p else 0
        t = tasks[p * ni:(p + 1) * ni]
        t["dst_x"], t["dst_y"] = bx[ib] >> ss, by[ib] >> ss
        t["w"] = t["h"] = np.maximum(bs[ib] >> ss, 4)
        t["plane"] = p
        t["flags"] = (filt << 1) | comp
        for r in range(2):
            # q4 position = 16*block origin + mv * (2 >> ss)   (mv in 1/8 luma pel; 1/16 units)
            t["pos_x"][:, r] = (t["dst_x"].astype(np.int32) << 4) + mv[:, r, 1] * (2 >> ss)
            t["pos_y"][:, r] = (t["dst_y"].astype(np.int32) << 4) + mv[:, r, 0] * (2 >> ss)
            t["ref"][:, r] = refs_idx[:, r]
            t["step_x"][:, r] = t["step_y"][:, r] = 16
    # interleave planes per block (decode order: block, plane)
    order = np.arange(ni * 3).reshape(3, ni).T.ravel()
    inter_tasks = tasks[order]

    # ---- transform blocks ---------------------------------------------------------------------
    recs = {"blk": [], "plane": [], "x": [], "y": [], "txs": [], "ridx": []}
    for p in range(3):
        ss = 1 if p else 0
        ps = np.maximum(bs >> ss, 4)                    # plane block size
        tp = np.minimum(tx, np.log2(ps).astype(np.int32)) if p else tx
        tp = np.minimum(tp, 5)
        for psz in np.unique(ps):
            for t_ in np.unique(tp[ps == psz]):
                sel = np.flatnonzero((ps == psz) & (tp == t_))
                k = psz >> t_
                oy, ox = np.meshgrid(np.arange(k) << t_, np.arange(k) << t_, indexing="ij")
                X = ((bx[sel] >> ss)[:, None] + ox.ravel()[None, :]).ravel()
                Y = ((by[sel] >> ss)[:, None] + oy.ravel()[None, :]).ravel()
                B = np.repeat(sel, k * k)
                R = np.tile(np.arange(k * k), len(sel))
                keep = (X < dims[p][0]) & (Y < dims[p][1])   # blocks starting outside are not visited
                recs["blk"].append(B[keep])
                recs["plane"].append(np.full(keep.sum(), p))
                recs["x"].append(X[keep])
                recs["y"].append(Y[keep])
                recs["txs"].append(np.full(keep.sum(), t_ - 2))
                recs["ridx"].append(R[keep])
    T = {k: np.concatenate(v) for k, v in recs.items()}
    key = T["blk"].astype(np.int64) * (3 << 12) + T["plane"] * (1 << 12) + T["ridx"]
    o = np.argsort(key, kind="stable")
    T = {k: v[o] for k, v in T.items()}
    nt = len(o)
    coded = ~skip[T["blk"]] & (rng.random(nt) < 0.8)
    cls = rng.choice(3, nt, p=[0.5, 0.3, 0.2])          # 0 DC only, 1 low-frequency, 2 dense
    nn = 4 << T["txs"]
    eob = np.where(cls == 0, 1, np.where(cls == 1, np.where(nn == 4, 16, 10), nn * nn)) * coded
    csize = np.where(coded, nn * nn, 0)
    coff = np.concatenate([[0], np.cumsum(csize)[:-1]]).astype(np.int64)
    coeffs = np.zeros(int(csize.sum()) + 16, np.int32)
    amp = coef_amp * (1 << (bd - 8))
    for txs in range(4):
        n = 4 << txs
        for c in range(3):
            sel = np.flatnonzero(coded & (T["txs"] == txs) & (cls == c))
            if len(sel) == 0:
                continue
            blk = np.zeros((len(sel), n, n), np.int32)
            if c == 0:
                blk[:, 0, 0] = rng.integers(-int(300 * amp), int(300 * amp) + 1, len(sel))
            else:
                k = min(n, 4) if c == 1 else n
                fy, fx = np.meshgrid(np.arange(k), np.arange(k), indexing="ij")
                scale = (120.0 * amp) / (1.0 + fy + fx) ** 1.5
                vals = rng.standard_normal((len(sel), k, k)) * scale
                if c == 1:
                    vals *= rng.random((len(sel), k, k)) < 0.6
                blk[:, :k, :k] = np.rint(vals).astype(np.int32)
                blk[:, 0, 0] += rng.integers(-int(200 * amp), int(200 * amp) + 1, len(sel))
            idx = coff[sel][:, None] + np.arange(n * n)[None, :]
            coeffs[idx.ravel()] = blk.reshape(len(sel), -1).ravel()
    t_inter = inter[T["blk"]]
    # inter residuals -> txb records (DCT only, vp9_decodeframe.c inverse_transform_block_inter)
    sel = np.flatnonzero(t_inter & coded)
    txb = np.zeros(len(sel), TXB_DTYPE)
    txb["coeff_off"], txb["x"], txb["y"] = coff[sel], T["x"][sel], T["y"][sel]
    txb["plane"], txb["tx_size"], txb["eob"] = T["plane"][sel], T["txs"][sel], eob[sel]
    so = np.argsort(txb["tx_size"], kind="stable")
    txb_sorted = txb[so]
    txb_counts = [int((txb_sorted["tx_size"] == s).sum()) for s in range(4)]

    # ---- intra tasks (decode order) -----------------------------------------------------------
    sel = np.flatnonzero(~t_inter)
    itasks = np.zeros(len(sel), INTRA_DTYPE)
    b_of = T["blk"][sel]
    ymode = rng.integers(0, 10, nb)
    uvmode = rng.integers(0, 10, nb)
    pl = T["plane"][sel]
    mode = np.where(pl == 0, ymode[b_of], uvmode[b_of])
    sub = (pl == 0) & (bs[b_of] == 8) & (T["txs"][sel] == 0)   # 4x4 luma in an 8x8 block: own modes
    mode = np.where(sub, rng.integers(0, 10, len(sel)), mode)
    itasks["coeff_off"], itasks["x"], itasks["y"], itasks["plane"] = coff[sel], T["x"][sel], T["y"][sel], pl
    itasks["tx_size"], itasks["mode"], itasks["eob"] = T["txs"][sel], mode, eob[sel]
    itasks["tx_type"] = np.where((pl == 0) & (T["txs"][sel] < 3), MODE_TO_TX_TYPE[mode], 0)
    ss = (pl > 0).astype(np.int32)
    pbx, pby = bx[b_of] >> ss, by[b_of] >> ss
    pbs = np.maximum(bs[b_of] >> ss, 4)
    n_px = 4 << T["txs"][sel]
    have_top = ((T["y"][sel] - pby) > 0) | (by[b_of] > 0)
    have_left = ((T["x"][sel] - pbx) > 0) | (bx[b_of] > 0)
    have_right = (T["x"][sel] - pbx + n_px) < pbs
    itasks["flags"] = have_top.astype(np.uint8) | (have_left.astype(np.uint8) << 1) | (have_right.astype(np.uint8) << 2)
    levels, comp = intra_levels(itasks, dims, want_components=True)
    isl_tasks, islands, isl_wave_off, big_tasks, big_wave_start, n_lds, row_pos = pack_intra_islands(
        itasks, levels, comp, sb_rows=(ah + 63) >> 6)
    lo = np.argsort(levels, kind="stable")
    itasks_sorted = itasks[lo]
    n_waves = int(levels.max()) if len(levels) else 0
    wave_start = np.searchsorted(levels[lo], np.arange(1, n_waves + 2)).astype(np.int32)

    lfm, sb_rows, sb_cols = build_lf_masks(blocks, aw, ah)
    dt = np.uint16 if hbd else np.uint8
    refs = []
    for r in range(n_refs):
        refs.append([smooth_noise(rng, d[1], d[0], bd, sigma=1.5 + r).astype(dt) for d in dims])
    return dict(width=width, height=height, bd=bd, hbd=hbd, dims=dims, crop=crop, refs=refs, blocks=blocks,
                inter_tasks=inter_tasks, txb=txb_sorted, txb_counts=txb_counts, coeffs=coeffs,
                intra_decode_order=itasks, intra_sorted=itasks_sorted, wave_start=wave_start, n_waves=n_waves,
                intra_island_tasks=isl_tasks, intra_islands=islands, intra_island_wave_off=isl_wave_off,
                intra_big_tasks=big_tasks, intra_big_wave_start=big_wave_start,
                island_sb_expected=island_sb_expected(isl_tasks, islands, sb_rows, sb_cols, isl_wave_off),
                n_islands_lds=n_lds, island_row_pos=row_pos,
                lfm=lfm, sb_rows=sb_rows, sb_cols=sb_cols, thresholds=lf_thresholds(sharpness),
                n_blocks=nb, n_txb=nt)
