"""Pins the loop-filter DRIVER level against the reference's own object code
(oracle/_ref: vp9_build_mask / vp9_adjust_mask / vp9_filter_block_plane_ss00+ss11 through
oracle/ref_lf_driver.c):
  * workload.build_lf_masks (host packing logic of the product) -> identical LOOP_FILTER_MASKs
  * oracle vp9o_loop_filter_frame (sequential restatement)      -> identical pixels."""
import ctypes

import numpy as np
import pytest

from frame_check import OThresh


@pytest.mark.parametrize("W,H,bd,sharp,seed", [(352, 288, 8, 0, 1), (200, 136, 8, 3, 2), (328, 72, 10, 0, 3),
                                                (640, 360, 8, 5, 4), (136, 200, 12, 7, 5)])
def test_masks_and_filtering_match_reference_driver(hip, oracle, ref, W, H, bd, sharp, seed):
    import workload
    wl = workload.make_frame_workload(W, H, seed=seed, bd=bd, sharpness=sharp, intra_frac=0.2)
    hbd = bd > 8
    dt = np.uint16 if hbd else np.uint8
    aw, ah = wl["dims"][0]
    rng = np.random.default_rng(seed)
    # piecewise-smooth planes so that every filter branch triggers; padded like a libvpx buffer
    pads = []
    for (pw, ph) in wl["dims"]:
        base = workload.smooth_noise(rng, ph + 32, pw + 32, bd, sigma=6.0).astype(np.int64)
        noise = rng.integers(-2, 3, base.shape) << (bd - 8)
        pads.append(np.clip(base // 2 + (1 << (bd - 2)) + noise, 0, (1 << bd) - 1).astype(dt))
    b = wl["blocks"]
    flat = np.stack([b["x"], b["y"], b["size"], b["tx"], b["level"], b["skip"].astype(np.int32),
                     b["inter"].astype(np.int32)], 1).astype(np.int32)
    flat = np.ascontiguousarray(flat)
    n_sb = wl["sb_rows"] * wl["sb_cols"]
    ref_planes = [p.copy() for p in pads]
    ptrs = (ctypes.c_void_p * 3)(*[p.ctypes.data for p in ref_planes])
    strides = (ctypes.c_int * 3)(*[p.shape[1] for p in ref_planes])
    lfm_ref = np.zeros(n_sb, hip.LFM_DTYPE)
    sz = ref.ref_lf_frame(flat.ctypes.data_as(ctypes.c_void_p), len(flat), aw, ah, ptrs, strides, bd, int(hbd), sharp,
                          lfm_ref.ctypes.data_as(ctypes.c_void_p), 1)
    assert sz in (154, 160), sz  # sizeof(LOOP_FILTER_MASK): 154 bytes of fields, padded to 160
    assert sz == hip.LFM_DTYPE.itemsize
    # 1. the product's mask builder == vp9_build_mask + vp9_adjust_mask
    mine = wl["lfm"]
    for f in ("left_y", "above_y", "int_4x4_y", "left_uv", "above_uv", "int_4x4_uv", "lfl_y"):
        cmp_m = mine[f][:, :3] if f in ("left_y", "above_y", "left_uv", "above_uv") else mine[f]
        cmp_r = lfm_ref[f][:, :3] if f in ("left_y", "above_y", "left_uv", "above_uv") else lfm_ref[f]
        assert np.array_equal(cmp_m, cmp_r), f
    # 2. the oracle's driver == vp9_filter_block_plane_ss00/ss11 in raster order
    mine_planes = [p.copy() for p in pads]
    th = OThresh()
    mblim, lim, hev = wl["thresholds"]
    for i in range(64):
        th.mblim[i], th.lim[i], th.hev_thr[i] = int(mblim[i]), int(lim[i]), int(hev[i])
    p2 = (ctypes.c_void_p * 3)(*[p.ctypes.data for p in mine_planes])
    oracle.vp9o_loop_filter_frame(lfm_ref.ctypes.data_as(ctypes.c_void_p), wl["sb_rows"], wl["sb_cols"],
                                  ctypes.byref(th), p2, strides, ah // 8, bd, int(hbd), 3)
    changed = 0
    for p, (pw, ph) in enumerate(wl["dims"]):
        assert np.array_equal(mine_planes[p][:ph, :pw], ref_planes[p][:ph, :pw]), f"plane {p}"
        changed += int((ref_planes[p][:ph, :pw] != pads[p][:ph, :pw]).sum())
    assert changed > (1000 if sharp < 6 else 0)
