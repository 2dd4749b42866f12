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
