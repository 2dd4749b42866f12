"""Small synthetic-structure helpers for the GPU parity tests (test infrastructure)."""
import numpy as np


def quad_blocks(rng, W, H, min_log2=3, max_log2=6, p_split=0.6):
    """Yield (x, y, size) square blocks in VP9 decode order (64x64 raster, recursive quad)."""
    out = []

    def rec(x, y, lg):
        if x >= W or y >= H:
            return
        s = 1 << lg
        if lg > min_log2 and (rng.random() < p_split or x + s > W or y + s > H):
            h = s >> 1
            rec(x, y, lg - 1)
            rec(x + h, y, lg - 1)
            rec(x, y + h, lg - 1)
            rec(x + h, y + h, lg - 1)
        else:
            out.append((x, y, s))

    for sy in range(0, H, 1 << max_log2):
        for sx in range(0, W, 1 << max_log2):
            rec(sx, sy, max_log2)
    return out


def intra_levels(tasks, plane_dims):
    """Dependency level per task (1-based), given tasks in decode order.
    tasks: list of dicts with plane,x,y,bs,have_top,have_left,have_right."""
    maps = [np.zeros((ah // 4 + 1, aw // 4 + 1), np.int32) for (aw, ah) in plane_dims]
    levels = []
    for t in tasks:
        m = maps[t["plane"]]
        aw, ah = plane_dims[t["plane"]]
        cx, cy, n = t["x"] // 4, t["y"] // 4, t["bs"] // 4
        ymax, xmax = ah // 4 - 1, aw // 4 - 1
        lv = 0
        if t["have_left"]:
            lv = max(lv, int(m[cy:min(cy + n, ymax + 1), cx - 1].max()))
        if t["have_top"]:
            ext = 2 * n if (t["bs"] == 4 and t["have_right"]) else n
            lv = max(lv, int(m[cy - 1, cx:min(cx + ext, xmax + 1)].max()))
            if t["have_left"]:
                lv = max(lv, int(m[cy - 1, cx - 1]))
        lv += 1
        m[cy:cy + n, cx:cx + n] = lv
        levels.append(lv)
    return np.array(levels, np.int32)


def random_lfm(rng, sb_rows, sb_cols, mi_rows, mi_cols, lfm_dtype, p_edge=0.6):
    """Random but libvpx-legal LOOP_FILTER_MASK records: one filter size per edge position,
    bits outside the frame removed the way vp9_adjust_mask does
    (vp9/common/vp9_loopfilter.c:766-860)."""
    out = np.zeros(sb_rows * sb_cols, lfm_dtype)
    for sr in range(sb_rows):
        for sc in range(sb_cols):
            m = out[sr * sb_cols + sc]
            rows = min(8, mi_rows - sr * 8)
            cols = min(8, mi_cols - sc * 8)
            lv = rng.integers(1, 64, 64).astype(np.uint8)
            if rng.random() < 0.3:
                lv[:] = rng.integers(1, 64)
            m["lfl_y"] = lv

            def pick(nbits):
                kinds = rng.integers(0, 3, nbits)
                on = rng.random(nbits) < p_edge
                masks = [0, 0, 0]
                for b in range(nbits):
                    if on[b]:
                        masks[kinds[b]] |= 1 << b
                return masks

            ly, ay = pick(64), pick(64)
            luv, auv = pick(16), pick(16)
            inty = int(rng.integers(0, 1 << 63)) & int(rng.integers(0, 1 << 63))
            intuv = int(rng.integers(0, 1 << 16))
            my = sum(((1 << cols) - 1) << (8 * r) for r in range(rows))
            crows, ccols = (rows + 1) >> 1, (cols + 1) >> 1
            muv = sum(((1 << ccols) - 1) << (4 * r) for r in range(crows))
            muv_int = sum(((1 << (cols >> 1)) - 1) << (4 * r) for r in range(crows))
            for k in range(3):
                ly[k] &= my
                ay[k] &= my
                luv[k] &= muv
                auv[k] &= muv
            inty &= my
            intuv &= muv_int
            if rows & 1:   # partial last chroma row: no 16-wide filter there
                rowbits = 0xf << (4 * (crows - 1))
                auv[1] |= auv[2] & rowbits
                auv[2] &= ~rowbits
                # 8-wide needs 4 rows below: fine; keep
            if cols & 1:
                colbits = 0x1111 << (ccols - 1)
                luv[1] |= luv[2] & colbits
                luv[2] &= ~colbits
            if sc == 0:
                for k in range(3):
                    ly[k] &= 0xfefefefefefefefe
                    luv[k] &= 0xeeee
            for k in range(3):
                m["left_y"][k], m["above_y"][k] = ly[k], ay[k]
                m["left_uv"][k], m["above_uv"][k] = luv[k], auv[k]
            m["int_4x4_y"], m["int_4x4_uv"] = inty, intuv
    return out


def lf_thresholds(sharpness=0):
    """lfthr table: update_sharpness + vp9_loop_filter_init (vp9_loopfilter.c:212-250)."""
    mblim, lim, hev = np.zeros(64, np.uint8), np.zeros(64, np.uint8), np.zeros(64, np.uint8)
    for lvl in range(64):
        bil = lvl >> ((sharpness > 0) + (sharpness > 4))
        if sharpness > 0 and bil > 9 - sharpness:
            bil = 9 - sharpness
        bil = max(bil, 1)
        lim[lvl] = bil
        mblim[lvl] = 2 * (lvl + 2) + bil
        hev[lvl] = lvl >> 4
    return mblim, lim, hev
