"""blockgen.py — random decoded-mode information for one VP9 frame (test input only).

Generates what libvpx's entropy stage leaves in MODE_INFO, as `vp9hip_block` records in decode
order, with the structure a real stream has: 64x64 superblocks in raster order, recursive
partitioning with all four partition types and the frame-edge rules of decode_partition /
read_partition (libvpx/vp9/decoder/vp9_decodeframe.c: has_rows / has_cols), all 13 block sizes
incl. sub-8x8 with per-4x4 motion vectors / intra modes (duplicated for 4x8 / 8x4 the way
read_inter_block_mode_info / read_intra_frame_mode_info fill bmi[]), compound prediction,
transform sizes up to max_txsize_lookup, skip flags, per-block filter levels.
Also: coefficients in the reference's layout (frameBuf, vpx-master/buffers_struct.h:9-15).
"""
import numpy as np

W4 = np.array([1, 1, 2, 2, 2, 4, 4, 4, 8, 8, 8, 16, 16])
H4 = np.array([1, 2, 1, 2, 4, 2, 4, 8, 4, 8, 16, 8, 16])
# BLOCK_SIZE of a square of 2^k 8-px units, and its HORZ / VERT halves
SQUARE = {0: 3, 1: 6, 2: 9, 3: 12}
HORZ = {0: 2, 1: 5, 2: 8, 3: 11}     # 8x4, 16x8, 32x16, 64x32
VERT = {0: 1, 1: 4, 2: 7, 3: 10}     # 4x8, 8x16, 16x32, 32x64
MAX_TX = np.array([0, 0, 0, 1, 1, 1, 2, 2, 2, 3, 3, 3, 3])  # max_txsize_lookup


def gen_blocks(rng, width, height, block_dtype, *, intra_frac=0.15, compound_frac=0.2, skip_frac=0.3,
               split_p=(0.9, 0.75, 0.6, 0.5), all_intra=False, n_refs=3, mv_amp=80, far_mv_frac=0.03,
               levels=(0, 12, 28, 40, 63)):
    aw, ah = (width + 7) & ~7, (height + 7) & ~7
    mi_rows, mi_cols = ah // 8, aw // 8
    recs = []

    def emit(mi_row, mi_col, sb_type):
        if mi_row >= mi_rows or mi_col >= mi_cols:
            return
        b = np.zeros((), block_dtype)
        b["mi_row"], b["mi_col"], b["sb_type"] = mi_row, mi_col, sb_type
        b["tx_size"] = rng.integers(0, MAX_TX[sb_type] + 1)
        b["skip"] = rng.random() < skip_frac
        b["filter_level"] = levels[rng.integers(0, len(levels))]
        inter = (not all_intra) and rng.random() >= intra_frac
        if inter:
            b["ref_frame"][0] = rng.integers(1, n_refs + 1)
            b["ref_frame"][1] = rng.integers(1, n_refs + 1) if rng.random() < compound_frac else -1
            if b["ref_frame"][1] == b["ref_frame"][0]:
                b["ref_frame"][1] = -1  # a compound pair uses two different references
            b["interp_filter"] = rng.integers(0, 4)
            far = rng.random() < far_mv_frac
            amp = 3000 if far else mv_amp
            base = rng.integers(-amp, amp + 1, (2, 2))
            if rng.random() < 0.1:
                base[:] = 0
            if rng.random() < 0.1:
                base &= ~7  # full-sample
            b["mv"] = base
            sub = base[None] + rng.integers(-24, 25, (4, 2, 2))
            if sb_type == 1:      # 4x8: bmi[2] = bmi[0], bmi[3] = bmi[1]
                sub[2], sub[3] = sub[0], sub[1]
            elif sb_type == 2:    # 8x4: bmi[1] = bmi[0], bmi[3] = bmi[2]
                sub[1], sub[3] = sub[0], sub[2]
            b["sub_mv"] = np.clip(sub, -16000, 16000)
        else:
            b["ref_frame"] = (0, -1)
            b["mode"] = rng.integers(0, 10)
            b["uv_mode"] = rng.integers(0, 10)
            sm = rng.integers(0, 10, 4)
            if sb_type == 1:
                sm[2], sm[3] = sm[0], sm[1]
            elif sb_type == 2:
                sm[1], sm[3] = sm[0], sm[2]
            b["sub_mode"] = sm
        recs.append(b)

    def part(mi_row, mi_col, k):
        """k: log2 of the square's size in 8-px units (3 = 64x64)."""
        if mi_row >= mi_rows or mi_col >= mi_cols:
            return
        n8 = 1 << k
        hbs = n8 >> 1
        has_rows = (mi_row + hbs) < mi_rows if k > 0 else True
        has_cols = (mi_col + hbs) < mi_cols if k > 0 else True
        if k == 0:
            # 8x8 level: the "partition" picks the sub-8x8 shape, one MODE_INFO
            r = rng.random()
            emit(mi_row, mi_col, 3 if r < 0.55 else 2 if r < 0.7 else 1 if r < 0.85 else 0)
            return
        split = rng.random() < split_p[3 - k]
        if has_rows and has_cols:
            choice = "split" if split else ("none", "horz", "vert")[rng.integers(0, 3)]
        elif not has_rows and has_cols:
            choice = "split" if split else "horz"
        elif has_rows and not has_cols:
            choice = "split" if split else "vert"
        else:
            choice = "split"
        if choice == "none":
            emit(mi_row, mi_col, SQUARE[k])
        elif choice == "horz":
            emit(mi_row, mi_col, HORZ[k])
            if has_rows:
                emit(mi_row + hbs, mi_col, HORZ[k])
        elif choice == "vert":
            emit(mi_row, mi_col, VERT[k])
            if has_cols:
                emit(mi_row, mi_col + hbs, VERT[k])
        else:
            for dy in (0, hbs):
                for dx in (0, hbs):
                    part(mi_row + dy, mi_col + dx, k - 1)

    for sr in range(0, mi_rows, 8):
        for sc in range(0, mi_cols, 8):
            part(sr, sc, 3)
    return np.array(recs, block_dtype)


def to_ref_records(blocks):
    """The 35-int32 layout of oracle/ref_frame_driver.c."""
    n = len(blocks)
    out = np.zeros((n, 35), np.int32)
    out[:, 0], out[:, 1], out[:, 2], out[:, 3] = blocks["mi_row"], blocks["mi_col"], blocks["sb_type"], blocks["tx_size"]
    out[:, 4], out[:, 5] = blocks["skip"], blocks["interp_filter"]
    out[:, 6], out[:, 7] = blocks["ref_frame"][:, 0], blocks["ref_frame"][:, 1]
    out[:, 8], out[:, 9] = blocks["mode"], blocks["uv_mode"]
    out[:, 10:14] = blocks["sub_mode"]
    out[:, 14] = blocks["filter_level"]
    out[:, 15:19] = blocks["mv"].reshape(n, 4)
    out[:, 19:35] = blocks["sub_mv"].reshape(n, 16)
    return np.ascontiguousarray(out)


def uv_tx(sb_type, tx_size):
    if sb_type < 3:
        return 0
    m = max(1, min(W4[sb_type] >> 1, H4[sb_type] >> 1))
    return min(int(tx_size), int(np.log2(m)))


def gen_coeffs(rng, blocks, width, height, bd, eob_stride_pad=0, amp=1.0, lossless=False, zero_frac=0.2):
    """Coefficients in the reference's layout: per plane one concatenated array (a slot for every
    visited transform block of every non-skip block) + frame-strided eob planes.
    Returns (coef[3], eob[3])."""
    aw, ah = (width + 7) & ~7, (height + 7) & ~7
    mi_rows, mi_cols = ah // 8, aw // 8
    dims = [(aw, ah), (aw // 2, ah // 2), (aw // 2, ah // 2)]
    eob = [np.full((d[1], d[0] + eob_stride_pad), -12345, np.int32) for d in dims]  # garbage where nothing is written
    coef = [[], [], []]
    a = amp * (1 << (bd - 8))
    for b in blocks:
        if b["skip"]:
            continue
        sbt = int(b["sb_type"])
        bw8, bh8 = max(1, W4[sbt] >> 1), max(1, H4[sbt] >> 1)
        for p in range(3):
            ss = 1 if p else 0
            n4w, n4h = max(1, (bw8 * 2) >> ss), max(1, (bh8 * 2) >> ss)
            tx = uv_tx(sbt, b["tx_size"]) if p else int(b["tx_size"])
            vis_w = min(n4w, ((mi_cols - int(b["mi_col"])) * 8 >> ss) // 4)
            vis_h = min(n4h, ((mi_rows - int(b["mi_row"])) * 8 >> ss) // 4)
            n = 4 << tx
            for row in range(0, vis_h, 1 << tx):
                for col in range(0, vis_w, 1 << tx):
                    x = ((int(b["mi_col"]) * 8) >> ss) + 4 * col
                    y = ((int(b["mi_row"]) * 8) >> ss) + 4 * row
                    blk = np.zeros((n, n), np.int32)
                    r = rng.random()
                    if r < zero_frac:
                        r = 0.0
                    else:
                        r = 0.2 + 0.8 * (r - zero_frac) / (1.0 - zero_frac)
                    if r < 0.2:
                        e = 0
                    elif r < 0.5:
                        e = 1
                        blk[0, 0] = rng.integers(-int(300 * a), int(300 * a) + 1)
                    elif r < 0.8 or lossless:
                        k = min(n, 4) if not lossless else n
                        blk[:k, :k] = rng.integers(-int(60 * a), int(60 * a) + 1, (k, k)) * (rng.random((k, k)) < 0.5)
                        blk[0, 0] += rng.integers(-int(200 * a), int(200 * a) + 1)
                        e = 10 if n > 4 and not lossless else 16
                    else:
                        fy, fx = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
                        blk[:] = np.rint(rng.standard_normal((n, n)) * (120.0 * a) / (1.0 + fy + fx) ** 1.5)
                        e = n * n
                    eob[p][y, x] = e
                    coef[p].append(blk.ravel())
    coef = [np.concatenate(c).astype(np.int32) if c else np.zeros(0, np.int32) for c in coef]
    return coef, eob


# intra_mode_to_tx_type_lookup (libvpx/vp9/common/vp9_blockd.h / vp9_entropy: DC and D45 are DCT_DCT)
_MODE_IS_DCT = {0: True, 3: True}


def coeff_rows(eob, tx_type_is_dct, tx):
    """The reference's clearing rule (vp9_decodeframe.c:960-967): rows of the block that can be non-zero."""
    n = 4 << tx
    if eob <= 0:
        return 0
    if eob == 1:
        return 1
    if tx_type_is_dct and tx <= 2 and eob <= 10:
        return 4
    if tx == 3 and eob <= 34:
        return 8
    return n


def compact_layout(blocks, coef, eob, width, height, lossless=False, gaps=(5, 0, 3), split=True):
    """From gen_coeffs' full slots to the compact, caller-placed layout (vp9hip_coeff_layout.compact, E12):
    every transform block's slot shrinks to the rows the clearing rule leaves, eob-0 blocks take nothing, and
    each plane's slots sit in one or two regions (a gap before each, as tile columns leave).
    Returns (coef_compact[3], tile_layout dict for Packer.pack / Decoder.begin_frame, full_offsets, extents)."""
    aw, ah = (width + 7) & ~7, (height + 7) & ~7
    mi_rows, mi_cols = ah // 8, aw // 8
    nb = len(blocks)
    block_off = np.zeros((nb, 3), np.uint32)
    out = [[np.zeros(g, np.int32) + 77] for g in gaps]      # garbage in the gaps
    pos = [g for g in gaps]
    starts = [[g] for g in gaps]
    ends = [[] for _ in gaps]
    run_full = [0, 0, 0]
    for i, b in enumerate(blocks):
        if split and i == nb // 2:
            for p in range(3):
                ends[p].append(pos[p])
                out[p].append(np.zeros(7, np.int32) - 99)
                pos[p] += 7
                starts[p].append(pos[p])
        for p in range(3):
            block_off[i, p] = pos[p]
        if b["skip"]:
            continue
        sbt = int(b["sb_type"])
        inter = int(b["ref_frame"][0]) > 0
        bw8, bh8 = max(1, W4[sbt] >> 1), max(1, H4[sbt] >> 1)
        for p in range(3):
            ss = 1 if p else 0
            n4w, n4h = max(1, (bw8 * 2) >> ss), max(1, (bh8 * 2) >> ss)
            tx = uv_tx(sbt, b["tx_size"]) if p else int(b["tx_size"])
            vis_w = min(n4w, ((mi_cols - int(b["mi_col"])) * 8 >> ss) // 4)
            vis_h = min(n4h, ((mi_rows - int(b["mi_row"])) * 8 >> ss) // 4)
            n = 4 << tx
            for row in range(0, vis_h, 1 << tx):
                for col in range(0, vis_w, 1 << tx):
                    x = ((int(b["mi_col"]) * 8) >> ss) + 4 * col
                    y = ((int(b["mi_row"]) * 8) >> ss) + 4 * row
                    e = int(eob[p][y, x])
                    is_dct = True
                    if not inter and not lossless and p == 0:
                        m = int(b["sub_mode"][(row << 1) + col]) if sbt < 3 else int(b["mode"])
                        is_dct = _MODE_IS_DCT.get(m, False)
                    ext = coeff_rows(e, is_dct, tx) * n
                    full = coef[p][run_full[p]:run_full[p] + n * n]
                    assert not full[ext:].any(), "generator broke the clearing-rule invariant"
                    out[p].append(full[:ext])
                    pos[p] += ext
                    run_full[p] += n * n
    for p in range(3):
        ends[p].append(pos[p])
    coef_c = [np.concatenate(o + [np.zeros(4, np.int32)]).astype(np.int32) for o in out]
    plane_base = [0, len(coef_c[0]), len(coef_c[0]) + len(coef_c[1])]
    regions = [(p, s, e - s) for p in range(3) for s, e in zip(starts[p], ends[p])]
    layout = dict(block_off=block_off, plane_base=plane_base, total=sum(len(c) for c in coef_c), regions=regions, compact=True)
    return coef_c, layout
