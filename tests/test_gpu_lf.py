"""GPU parity: vp9hip_loop_filter_frame (anti-diagonal superblock wavefront) vs the oracle's
sequential raster-order driver."""
import ctypes

import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu


class OThresh(ctypes.Structure):
    _fields_ = [("mblim", ctypes.c_uint8 * 64), ("lim", ctypes.c_uint8 * 64), ("hev_thr", ctypes.c_uint8 * 64)]


def smooth_planes(rng, dims, bd, dt):
    """piecewise-smooth content so that flat / flat2 / hev branches all trigger"""
    out = []
    for (aw, ah) in dims:
        base = rng.integers(0, 1 << bd, (ah // 8 + 1, aw // 8 + 1))
        img = np.kron(base, np.ones((8, 8), np.int64))[:ah, :aw]
        amp = rng.choice([0, 1, 2, 4, 24], (ah // 16 + 1, aw // 16 + 1)) << (bd - 8)
        ampf = np.kron(amp, np.ones((16, 16), np.int64))[:ah, :aw]
        noise = (rng.random((ah, aw)) * 2 - 1) * ampf
        # make neighbouring 8x8 means close in places
        img = (img // 3 + (img.mean() * 2 // 3)).astype(np.int64)
        out.append(np.clip(img + noise, 0, (1 << bd) - 1).astype(dt))
    return out


@pytest.mark.parametrize("W,H,bd,hbd,sharp", [(256, 192, 8, False, 0), (200, 136, 8, False, 3),
                                                (328, 200, 10, True, 0), (136, 72, 12, True, 6),
                                                # 100 superblock rows x 3 planes = 300 workgroups > 256 CUs: rows
                                                # wait for rows that must already be resident (dispatch order)
                                                (136, 6400, 8, False, 0)])
def test_loop_filter_frame_matches_oracle(hip, oracle, W, H, bd, hbd, sharp):
    rng = np.random.default_rng(400 + W + bd)
    dt = np.uint16 if hbd else np.uint8
    ctx = hip.Context(0)
    frame = hip.DevFrame(ctx, W, H, bit_depth=bd, hbd=hbd)
    dims = [(d[2], d[3]) for d in frame.dims]
    planes = smooth_planes(rng, dims, bd, dt)
    frame.upload(planes)
    aw, ah = dims[0]
    mi_rows, mi_cols = ah // 8, aw // 8
    sb_rows, sb_cols = (mi_rows + 7) // 8, (mi_cols + 7) // 8
    lfm = synth.random_lfm(rng, sb_rows, sb_cols, mi_rows, mi_cols, hip.LFM_DTYPE)
    mblim, lim, hev = synth.lf_thresholds(sharp)
    th = hip.LfThresh()
    oth = OThresh()
    for i in range(64):
        th.mblim[i] = oth.mblim[i] = int(mblim[i])
        th.lim[i] = oth.lim[i] = int(lim[i])
        th.hev_thr[i] = oth.hev_thr[i] = int(hev[i])
    # oracle, sequential
    # libvpx filters whole 8-sample segments even where only half of one lies inside the
    # (aligned) chroma plane; its frame buffers have a border for that.  Give the oracle one.
    padded = []
    for pl in planes:
        buf = np.zeros((pl.shape[0] + 16, pl.shape[1] + 16), dt)
        buf[:pl.shape[0], :pl.shape[1]] = pl
        padded.append(buf)
    ptrs = (ctypes.c_void_p * 3)(*[e.ctypes.data for e in padded])
    strides = (ctypes.c_int * 3)(*[e.shape[1] for e in padded])
    oracle.vp9o_loop_filter_frame(lfm.ctypes.data_as(ctypes.c_void_p), sb_rows, sb_cols, ctypes.byref(oth), ptrs,
                                  strides, mi_rows, bd, int(hbd), 3)
    expect = [b[:pl.shape[0], :pl.shape[1]] for b, pl in zip(padded, planes)]
    d_lfm = ctx.alloc(lfm)
    ctx.loop_filter_frame(d_lfm, sb_rows, sb_cols, th, frame, 3)
    ctx.sync()
    got = frame.download()
    changed = 0
    for p in range(3):
        bad = np.argwhere(got[p] != expect[p])
        assert bad.size == 0, f"plane {p}: {len(bad)} px differ, first {bad[:6]}"
        changed += int((expect[p] != planes[p]).sum())
    assert changed > 500  # the filters actually did something
    ctx.close()
