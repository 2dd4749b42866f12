"""Pins the product's host packers (cuda-vp9_amd/csrc/vp9hip_pack.c, through the C-ABI of
include/vp9hip_pack.h) against the reference's own object code (oracle/_ref via
oracle/ref_frame_driver.c).  No GPU: the packed lists are executed by the oracle's block
functions, which are themselves pinned against the reference (test_oracle_vs_ref.py).

  inter    vp9_build_inter_predictors_sb on bordered frames      == packer tasks -> oracle blocks
  intra    vp9_foreach_transformed_block_in_plane + vp9_predict_intra_block (tile-aware)
                                                                 == packer tasks -> oracle blocks
  masks    vp9_build_mask + vp9_adjust_mask, any BLOCK_SIZE      == packer lfm
  levels   vp9_loop_filter_frame_init                            == vp9hip_lf_frame_init
  islands  any order the island/wave lists allow                 == decode order
"""
import ctypes

import numpy as np
import pytest

import blockgen
from frame_check import OFrame

BORDER = 192


def _params(hip, W, H, bd, refs=None, tiles=0, lossless=0, lf=1):
    P = hip.FrameParams()
    P.width, P.height, P.ss_x, P.ss_y = W, H, 1, 1
    P.bit_depth, P.hbd, P.lossless, P.log2_tile_cols, P.build_lf_masks = bd, int(bd > 8), lossless, tiles, lf
    for k, (rw, rh) in enumerate(refs or [(W, H)] * 3):
        P.ref_width[k], P.ref_height[k] = rw, rh
    return P


def _plane_dims(W, H):
    aw, ah = (W + 7) & ~7, (H + 7) & ~7
    return ([(aw, ah), (aw // 2, ah // 2), (aw // 2, ah // 2)],
            [(W, H), ((W + 1) // 2, (H + 1) // 2), ((W + 1) // 2, (H + 1) // 2)])


def _oframe(planes, dims, crop, bd):
    f = OFrame()
    for p, a in enumerate(planes):
        f.plane[p], f.stride[p] = a.ctypes.data, a.shape[1]
        f.width[p], f.height[p] = crop[p]
        f.awidth[p], f.aheight[p] = dims[p]
    f.bit_depth, f.hbd = bd, int(bd > 8)
    return f


def _bordered(core, crop_w, crop_h):
    """Plane with a BORDER replicated from the crop edge (what vpx_extend_frame_borders leaves)."""
    return np.ascontiguousarray(np.pad(core[:crop_h, :crop_w], BORDER, mode="edge"))


def _ptr(arr, r, c):
    return ctypes.c_void_p(arr.ctypes.data + (r * arr.shape[1] + c) * arr.itemsize)


@pytest.mark.parametrize("W,H,bd,refs,seed", [
    (352, 288, 8, None, 1),
    (200, 136, 8, None, 2),                                   # not a multiple of 64; right/bottom overhang
    (330, 250, 10, None, 3),                                  # crop != aligned size
    (256, 192, 8, [(512, 384), (256, 192), (128, 96)], 4),    # 2:1 down-, 1:2 up-scaled references
    (232, 168, 12, [(232, 168), (348, 252), (174, 126)], 5),  # 3:2 and 3:4
    (256, 192, 8, [(300, 200), (256, 192), (201, 151)], 6),   # odd ratios: fractional block origins
])
def test_inter_packing_matches_reference(hip, oracle, ref, W, H, bd, refs, seed):
    rng = np.random.default_rng(seed)
    dt = np.uint16 if bd > 8 else np.uint8
    blocks = blockgen.gen_blocks(rng, W, H, hip.BLOCK_DTYPE, intra_frac=0.1, compound_frac=0.3)
    ref_sizes = refs or [(W, H)] * 3
    pk = hip.Packer()
    L = pk.pack(_params(hip, W, H, bd, refs=ref_sizes, lf=0), blocks)
    dims, crop = _plane_dims(W, H)
    # reference frames: random content; ours unbordered, the reference's bordered from the crop edge
    ours_refs, theirs_refs = [], []
    for (rw, rh) in ref_sizes:
        rd, rc = _plane_dims(rw, rh)
        planes = [rng.integers(0, 1 << bd, (d[1], d[0])).astype(dt) for d in rd]
        ours_refs.append((planes, rd, rc))
        theirs_refs.append([_bordered(pl, c[0], c[1]) for pl, c in zip(planes, rc)])
    init = [rng.integers(0, 1 << bd, (d[1], d[0])).astype(dt) for d in dims]
    # ---- ours: packed tasks through the oracle's block predictor
    mine = [p.copy() for p in init]
    dst = _oframe(mine, dims, crop, bd)
    rarr = (OFrame * 3)(*[_oframe(pl, rd, rc, bd) for (pl, rd, rc) in ours_refs])
    tasks = L["inter_tasks"]
    assert sum(L["inter_class_count"]) == len(tasks)
    oracle.vp9o_recon_inter_list(tasks.ctypes.data_as(ctypes.c_void_p), len(tasks), rarr, ctypes.byref(dst))
    # ---- theirs: vp9_build_inter_predictors_sb per block
    theirs = [np.ascontiguousarray(np.pad(p, BORDER, mode="constant")) for p in init]
    cur_ptrs = (ctypes.c_void_p * 3)(*[_ptr(a, BORDER, BORDER).value for a in theirs])
    cur_strides = (ctypes.c_int * 3)(*[a.shape[1] for a in theirs])
    flat = [a for r in theirs_refs for a in r]
    ref_ptrs = (ctypes.c_void_p * 9)(*[_ptr(a, BORDER, BORDER).value for a in flat])
    ref_strides = (ctypes.c_int * 9)(*[a.shape[1] for a in flat])
    rw = (ctypes.c_int * 3)(*[s[0] for s in ref_sizes])
    rh = (ctypes.c_int * 3)(*[s[1] for s in ref_sizes])
    recs = blockgen.to_ref_records(blocks)
    rc = ref.ref_inter_frame(recs.ctypes.data_as(ctypes.c_void_p), len(recs), W, H, 1, bd, int(bd > 8), cur_ptrs,
                             cur_strides, ref_ptrs, ref_strides, rw, rh)
    assert rc == 0
    n_inter = int((blocks["ref_frame"][:, 0] > 0).sum())
    assert n_inter > 20 and len(tasks) >= 3 * n_inter
    for p, (aw, ah) in enumerate(dims):
        got, exp = mine[p], theirs[p][BORDER:BORDER + ah, BORDER:BORDER + aw]
        bad = np.argwhere(got != exp)
        assert bad.size == 0, f"plane {p}: {len(bad)} samples differ, first at {bad[:4].tolist()}"
        assert (exp != init[p]).any()
    pk.close()


@pytest.mark.parametrize("W,H,bd,tiles,seed", [(352, 288, 8, 0, 1), (520, 136, 8, 1, 2), (1032, 72, 10, 2, 3),
                                                (200, 200, 12, 0, 4), (72, 64, 8, 0, 5)])
def test_intra_packing_matches_reference(hip, oracle, ref, W, H, bd, tiles, seed):
    rng = np.random.default_rng(seed)
    dt = np.uint16 if bd > 8 else np.uint8
    blocks = blockgen.gen_blocks(rng, W, H, hip.BLOCK_DTYPE, intra_frac=0.7)
    pk = hip.Packer()
    L = pk.pack(_params(hip, W, H, bd, tiles=tiles, lf=0), blocks)
    dims, crop = _plane_dims(W, H)
    init = [rng.integers(0, 1 << bd, (d[1], d[0])).astype(dt) for d in dims]
    mine = [p.copy() for p in init]
    f = _oframe(mine, dims, crop, bd)
    tasks = L["intra_decode_order"]
    oracle.vp9o_recon_intra_list(tasks.ctypes.data_as(ctypes.c_void_p), len(tasks), None, ctypes.byref(f))
    theirs = [np.ascontiguousarray(np.pad(p, BORDER, mode="constant", constant_values=77)) for p in init]
    ptrs = (ctypes.c_void_p * 3)(*[_ptr(a, BORDER, BORDER).value for a in theirs])
    strides = (ctypes.c_int * 3)(*[a.shape[1] for a in theirs])
    recs = blockgen.to_ref_records(blocks)
    log = np.zeros((len(tasks) + 16, 5), np.int32)
    n = ref.ref_intra_frame(recs.ctypes.data_as(ctypes.c_void_p), len(recs), W, H, 1, bd, int(bd > 8), ptrs, strides,
                            tiles, log.ctypes.data_as(ctypes.c_void_p), len(log))
    # same transform blocks, same order, same transform size and mode
    assert n == len(tasks)
    assert np.array_equal(log[:n, 0], tasks["plane"])
    assert np.array_equal(log[:n, 3], tasks["tx_size"])
    assert np.array_equal(log[:n, 4], tasks["mode"])
    # same pixels (edge availability, tile columns, frame-edge handling)
    for p, (aw, ah) in enumerate(dims):
        exp = theirs[p][BORDER:BORDER + ah, BORDER:BORDER + aw]
        bad = np.argwhere(mine[p] != exp)
        assert bad.size == 0, f"plane {p}: {len(bad)} samples differ, first at {bad[:4].tolist()}"
    if tiles:
        # the tile rule must have mattered: some task at a tile's first column lost have_left
        assert ((tasks["flags"] & 2) == 0).sum() > ((tasks["x"] == 0) & ((tasks["flags"] & 2) == 0)).sum()
    pk.close()


@pytest.mark.parametrize("W,H,bd,seed", [(352, 288, 8, 1), (200, 136, 8, 2), (328, 72, 10, 3), (72, 328, 8, 4),
                                          (1000, 40, 8, 5)])
def test_lf_masks_match_reference(hip, ref, W, H, bd, seed):
    rng = np.random.default_rng(seed)
    blocks = blockgen.gen_blocks(rng, W, H, hip.BLOCK_DTYPE, intra_frac=0.3, levels=(0, 5, 20, 33, 63))
    pk = hip.Packer()
    L = pk.pack(_params(hip, W, H, bd), blocks)
    aw, ah = (W + 7) & ~7, (H + 7) & ~7
    n_sb = L["sb_rows"] * L["sb_cols"]
    lfm_ref = np.zeros(n_sb, hip.LFM_DTYPE)
    recs = blockgen.to_ref_records(blocks)
    null3 = (ctypes.c_void_p * 3)()
    z3 = (ctypes.c_int * 3)()
    sz = ref.ref_lf_frame2(recs.ctypes.data_as(ctypes.c_void_p), len(recs), aw, ah, null3, z3, bd, int(bd > 8), 0,
                           lfm_ref.ctypes.data_as(ctypes.c_void_p), 0)
    assert sz == hip.LFM_DTYPE.itemsize
    mine = L["lfm"]
    assert len(mine) == n_sb
    for f in ("left_y", "above_y", "left_uv", "above_uv"):
        assert np.array_equal(mine[f][:, :3], lfm_ref[f][:, :3]), f
    for f in ("int_4x4_y", "int_4x4_uv", "lfl_y"):
        assert np.array_equal(mine[f], lfm_ref[f]), f
    assert (mine["left_y"] != 0).any() and (mine["int_4x4_uv"] != 0).any()
    pk.close()


@pytest.mark.parametrize("lvl,sharp,seg,absd,deltas", [
    (0, 0, None, 0, None), (28, 0, None, 0, None), (40, 3, {1: -10, 5: 20}, 0, None),
    (63, 7, {0: 12, 7: 63}, 1, ((1, 0, -1, -1), (0, 0))), (33, 5, {2: -40}, 0, ((2, -3, 5, -7), (4, -6))),
    (12, 1, None, 0, ((1, 0, -1, -1), (0, 0))),
])
def test_lf_level_table_matches_reference(hip, ref, lvl, sharp, seg, absd, deltas):
    L = hip.lib()
    se = (ctypes.c_int32 * 8)()
    sd = (ctypes.c_int32 * 8)()
    for k, v in (seg or {}).items():
        se[k], sd[k] = 1, v
    rd = (ctypes.c_int8 * 4)(*(deltas[0] if deltas else (0, 0, 0, 0)))
    md = (ctypes.c_int8 * 2)(*(deltas[1] if deltas else (0, 0)))
    mine_lvl = np.zeros((8, 4, 2), np.uint8)
    mine_th = hip.LfThresh()
    L.vp9hip_lf_frame_init(lvl, sharp, se, sd, absd, int(deltas is not None), rd, md,
                           mine_lvl.ctypes.data_as(ctypes.c_void_p), ctypes.byref(mine_th))
    ref_lvl = np.zeros((8, 4, 2), np.uint8)
    ref_th = np.zeros((3, 64), np.uint8)
    ref.ref_lf_levels(lvl, sharp, se, sd, absd, int(deltas is not None), rd, md,
                      ref_lvl.ctypes.data_as(ctypes.c_void_p), ref_th.ctypes.data_as(ctypes.c_void_p))
    if deltas is not None:
        ref_lvl[:, 0, 1] = mine_lvl[:, 0, 1]  # lvl[seg][INTRA_FRAME][1] is never written nor read by libvpx
    assert np.array_equal(mine_lvl, ref_lvl)
    assert np.array_equal(np.frombuffer(bytes(mine_th), np.uint8).reshape(3, 64), ref_th)


@pytest.mark.parametrize("W,H,kw,seed", [(352, 288, dict(intra_frac=0.3), 1), (256, 256, dict(all_intra=True), 2),
                                          (640, 360, dict(intra_frac=0.08), 3)])
def test_island_and_wave_lists_preserve_dependencies(hip, oracle, W, H, kw, seed):
    """Executing the intra tasks island by island (waves in order, tasks of a wave in REVERSE order
    to shake out hidden order dependence) gives the decode-order result."""
    rng = np.random.default_rng(seed)
    bd = 8
    blocks = blockgen.gen_blocks(rng, W, H, hip.BLOCK_DTYPE, **kw)
    pk = hip.Packer()
    L = pk.pack(_params(hip, W, H, bd, lf=0), blocks)
    dims, crop = _plane_dims(W, H)
    init = [rng.integers(0, 256, (d[1], d[0])).astype(np.uint8) for d in dims]
    a = [p.copy() for p in init]
    fa = _oframe(a, dims, crop, bd)
    dec = L["intra_decode_order"]
    oracle.vp9o_recon_intra_list(dec.ctypes.data_as(ctypes.c_void_p), len(dec), None, ctypes.byref(fa))
    b = [p.copy() for p in init]
    fb = _oframe(b, dims, crop, bd)
    isl, woff, tasks = L["intra_islands"], L["intra_island_wave_off"], L["intra_island_tasks"]
    big, bws = L["intra_big_tasks"], L["intra_big_wave_start"]
    assert len(tasks) + len(big) == len(dec)
    order = []
    for r in isl[::-1]:  # islands are independent: any order
        for w in range(r["n_waves"]):
            s, e = woff[r["wave_off_start"] + w], woff[r["wave_off_start"] + w + 1]
            assert e > s
            order.extend(range(r["task_start"] + e - 1, r["task_start"] + s - 1, -1))
    assert sorted(order) == list(range(len(tasks)))
    seq = tasks[order] if len(order) else tasks
    oracle.vp9o_recon_intra_list(seq.ctypes.data_as(ctypes.c_void_p), len(seq), None, ctypes.byref(fb))
    for w in range(len(bws) - 1):
        seg = np.ascontiguousarray(big[bws[w]:bws[w + 1]][::-1])
        oracle.vp9o_recon_intra_list(seg.ctypes.data_as(ctypes.c_void_p), len(seg), None, ctypes.byref(fb))
    for p in range(3):
        assert np.array_equal(a[p], b[p]), f"plane {p}"
    if kw.get("all_intra"):
        assert len(big) > 0 or len(isl) >= 1
    pk.close()


@pytest.mark.parametrize("W,H,kw,seed", [(352, 288, dict(intra_frac=0.3), 5), (640, 360, dict(intra_frac=0.08), 6),
                                          (200, 136, dict(intra_frac=0.5), 7)])
def test_island_superblock_marks(hip, W, H, kw, seed):
    """vp9hip_intra_islands_lf hand-over data: per island exactly the LAST task (list order) inside each luma
    superblock carries bit 0 of `reserved`, island_sb_expected counts the marks per superblock, and the
    Python mirror the kernel-level tests use (tests/workload.island_sb_expected) produces the same marks and counts."""
    import workload
    rng = np.random.default_rng(seed)
    blocks = blockgen.gen_blocks(rng, W, H, hip.BLOCK_DTYPE, **kw)
    pk = hip.Packer()
    L = pk.pack(_params(hip, W, H, 8, lf=1), blocks)
    isl, woff, tasks, exp = L["intra_islands"], L["intra_island_wave_off"], L["intra_island_tasks"], L["island_sb_expected"]
    sb_rows, sb_cols = L["sb_rows"], L["sb_cols"]
    assert len(isl) > 0 and len(exp) == sb_rows * sb_cols
    count = np.zeros(sb_rows * sb_cols, np.int64)
    n_lds, row_pos = L["n_islands_lds"], L["island_row_pos"]
    assert 0 < n_lds <= len(isl) and len(row_pos) == sb_rows
    n_fit = 0
    for k, r in enumerate(isl):
        n = int(woff[r["wave_off_start"] + r["n_waves"]])
        t = tasks[r["task_start"]:r["task_start"] + n]
        sc = (t["plane"] > 0).astype(np.int64)
        sb = ((t["y"].astype(np.int64) << sc) >> 6) * sb_cols + ((t["x"].astype(np.int64) << sc) >> 6)
        assert sb.max() < sb_rows * sb_cols
        last = {int(v): i for i, v in enumerate(sb)}  # last index per superblock
        want = np.zeros(n, np.uint8)
        want[list(last.values())] = 1
        assert np.array_equal(t["reserved"] & 1, want)
        # a marked task must be in the island's last wave that touches its superblock
        wave_of = np.searchsorted(woff[r["wave_off_start"]:r["wave_off_start"] + r["n_waves"] + 1], np.arange(n), side="right") - 1
        for v, i in last.items():
            assert wave_of[i] == wave_of[sb == v].max()
            count[v] += 1
        n_fit += workload.island_fits(t)
    assert n_fit == n_lds  # VP9HIP_ISLAND_FITS and its mirror agree
    assert np.array_equal(count, exp)
    mine = tasks.copy()
    assert np.array_equal(workload.island_sb_expected(mine, isl, sb_rows, sb_cols, woff), exp)
    assert np.array_equal(mine["reserved"], tasks["reserved"])
    # grid order of the fused launch: islands by group g = max(first superblock row - 1, 0); filter row r sits
    # behind row_pos[r] islands, which are all the islands that touch superblock rows <= r + 1 — a row only ever
    # waits for workgroups in front of it
    g = np.maximum((isl["reserved"] & 255).astype(np.int64) - 1, 0)
    assert np.all(np.diff(g) >= 0)
    assert np.array_equal(row_pos, np.cumsum(np.bincount(g, minlength=sb_rows))[:sb_rows])
    first_row = (isl["reserved"] & 255).astype(np.int64)
    for r in range(sb_rows):
        assert np.all(np.flatnonzero(first_row <= r + 1) < row_pos[r])
    assert row_pos[-1] == len(isl)
    pk.close()


def test_packer_rejects_bad_input(hip):
    pk = hip.Packer()
    blocks = np.zeros(1, hip.BLOCK_DTYPE)
    blocks["sb_type"] = 12
    blocks["ref_frame"] = (1, -1)
    P = _params(hip, 64, 64, 8)
    P.ref_width[0] = 1000  # more than 2x the frame: invalid scale
    with pytest.raises(hip.Vp9HipError, match="no valid size"):
        pk.pack(P, blocks)
    P = _params(hip, 64, 64, 8)
    P.ss_y = 0
    with pytest.raises(hip.Vp9HipError, match="4:2:0"):
        pk.pack(P, blocks)
    blocks["mi_col"] = 9
    with pytest.raises(hip.Vp9HipError, match="out of range"):
        pk.pack(_params(hip, 64, 64, 8), blocks)
    pk.close()
