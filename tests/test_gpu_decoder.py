"""GPU parity of the frame-level driver (include/vp9hip_decoder.h) on frames with a REAL VP9 block
structure (all 13 block sizes, sub-8x8, compound, tiles; tests/blockgen.py): host frames and the
reference's coefficient layout in, all three phases — inter + residual, intra, LOOP FILTER with the
C packer's masks and vp9hip_lf_frame_init's thresholds — reconstructed frame out, against the
oracle's sequential reconstruction of the same packed lists.  Also: references kept resident in the
frame pool across frames give the same result as re-uploading them."""
import ctypes

import numpy as np
import pytest

import blockgen
from frame_check import OFrame, OThresh

pytestmark = pytest.mark.gpu


def _dims(W, H):
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


def _params(hip, W, H, bd, tiles):
    P = hip.FrameParams()
    P.width, P.height, P.ss_x, P.ss_y, P.bit_depth, P.hbd = W, H, 1, 1, bd, int(bd > 8)
    P.log2_tile_cols, P.build_lf_masks = tiles, 1
    for k in range(3):
        P.ref_width[k], P.ref_height[k] = W, H
    return P


def _thresholds(hip, sharpness):
    th = hip.LfThresh()
    hip.lib().vp9hip_lf_frame_init(32, sharpness, None, None, 0, 0, None, None, None, ctypes.byref(th))
    return th


def _oracle_frame(hip, oracle, P, blocks, coef, eob, refs, W, H, bd, th):
    """inter + residual + intra + loop filter through the oracle, from the packer's lists."""
    pk = hip.Packer()
    L = pk.pack(P, blocks, eob)
    pk.close()
    dims, crop = _dims(W, H)
    dt = np.uint16 if bd > 8 else np.uint8
    PAD = 16
    bufs = [np.zeros((d[1] + PAD, d[0] + PAD), dt) for d in dims]
    dst = _oframe(bufs, dims, crop, bd)
    rarr = (OFrame * 3)(*[_oframe(r, dims, crop, bd) for r in refs])
    allc = np.concatenate(coef + [np.zeros(16, np.int32)])
    cp = allc.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
    for key, fn in (("inter_tasks", "vp9o_recon_inter_list"), ("txb", "vp9o_recon_txb_list"),
                    ("intra_decode_order", "vp9o_recon_intra_list")):
        t = L[key]
        if not len(t):
            continue
        if key == "inter_tasks":
            oracle.vp9o_recon_inter_list(t.ctypes.data_as(ctypes.c_void_p), len(t), rarr, ctypes.byref(dst))
        else:
            getattr(oracle, fn)(t.ctypes.data_as(ctypes.c_void_p), len(t), cp, ctypes.byref(dst))
    oth = OThresh.from_buffer_copy(bytes(th))
    ptrs = (ctypes.c_void_p * 3)(*[b.ctypes.data for b in bufs])
    strides = (ctypes.c_int * 3)(*[b.shape[1] for b in bufs])
    oracle.vp9o_loop_filter_frame(L["lfm"].ctypes.data_as(ctypes.c_void_p), L["sb_rows"], L["sb_cols"], ctypes.byref(oth),
                                  ptrs, strides, dims[0][1] // 8, bd, int(bd > 8), 3)
    return [b[:d[1], :d[0]].copy() for b, d in zip(bufs, dims)]


@pytest.mark.parametrize("W,H,bd,tiles,sharp,kw", [
    (352, 288, 8, 0, 0, {}),
    (330, 250, 10, 1, 3, dict(intra_frac=0.4)),
    (200, 136, 12, 0, 6, dict(compound_frac=0.5)),
    (1280, 720, 8, 2, 0, dict(levels=(0, 8, 30, 63))),
    (256, 256, 8, 0, 0, dict(all_intra=True)),
    (2560, 1440, 8, 3, 0, dict(intra_frac=0.08)),   # BASELINE.json's frame size, 8 tile columns
    (8, 8, 8, 0, 0, dict(intra_frac=0.5)),          # one 8x8 block: every edge is a frame edge
    (24, 40, 10, 0, 2, dict(intra_frac=0.3)),
    (72, 16, 8, 0, 0, {}),
])
def test_decoder_three_phases_match_oracle(hip, oracle, W, H, bd, tiles, sharp, kw):
    import workload
    rng = np.random.default_rng(W * 7 + H + bd)
    dt = np.uint16 if bd > 8 else np.uint8
    dims, crop = _dims(W, H)
    blocks = blockgen.gen_blocks(rng, W, H, hip.BLOCK_DTYPE, **kw)
    coef, eob = blockgen.gen_coeffs(rng, blocks, W, H, bd)
    refs = [[np.ascontiguousarray(workload.smooth_noise(rng, d[1], d[0], bd, sigma=1.5 + k).astype(dt)) for d in dims]
            for k in range(3)]
    P = _params(hip, W, H, bd, tiles)
    th = _thresholds(hip, sharp)
    expect = _oracle_frame(hip, oracle, P, blocks, coef, eob, refs, W, H, bd, th)

    dec = hip.Decoder(0)
    for k in range(3):
        dec.upload(k, refs[k], W, H, bd)
    dec.alloc_slot(3, W, H, bd)
    dec.begin_frame(P, blocks, eob, coef)
    dec.run(hip.PHASE_INTER | hip.PHASE_INTRA | hip.PHASE_LF, (0, 1, 2), 3, thresh=th)
    dec.sync()
    assert dec.last_run_ms() > 0
    got = [np.zeros((d[1], d[0]), dt) for d in dims]
    dec.download(3, got, W, H, bd)
    for p in range(3):
        bad = np.argwhere(got[p] != expect[p])
        assert bad.size == 0, f"plane {p}: {len(bad)} samples differ, first at {bad[:5].tolist()}"

    # second frame: the frame just decoded stays in the pool and serves as LAST without an upload
    blocks2 = blockgen.gen_blocks(rng, W, H, hip.BLOCK_DTYPE, intra_frac=0.1)
    blocks2["ref_frame"][:, 1] = -1
    coef2, eob2 = blockgen.gen_coeffs(rng, blocks2, W, H, bd)
    refs2 = [expect, refs[1], refs[2]]
    expect2 = _oracle_frame(hip, oracle, P, blocks2, coef2, eob2, refs2, W, H, bd, th)
    dec.alloc_slot(4, W, H, bd)
    dec.begin_frame(P, blocks2, eob2, coef2)
    dec.run(hip.PHASE_INTER | hip.PHASE_INTRA | hip.PHASE_LF, (3, 1, 2), 4, thresh=th)
    dec.sync()
    got2 = [np.zeros((d[1], d[0]), dt) for d in dims]
    dec.download(4, got2, W, H, bd)
    for p in range(3):
        assert np.array_equal(got2[p], expect2[p]), f"second frame, plane {p}"
    dec.close()


def test_decoder_rejects_inconsistent_calls(hip):
    dec = hip.Decoder(0)
    P = _params(hip, 64, 64, 8, 0)
    blocks = np.zeros(1, hip.BLOCK_DTYPE)
    blocks["sb_type"], blocks["ref_frame"] = 12, (1, -1)
    with pytest.raises(hip.Vp9HipError, match="no frame begun"):
        dec.run(hip.PHASE_INTER, (-1, -1, -1), 3)
    dec.begin_frame(P, blocks)
    dec.alloc_slot(3, 64, 64, 8)
    with pytest.raises(hip.Vp9HipError, match="reference 0 is used"):
        dec.run(hip.PHASE_INTER, (-1, -1, -1), 3)
    dec.alloc_slot(0, 32, 32, 8)
    with pytest.raises(hip.Vp9HipError, match="in the pool"):
        dec.run(hip.PHASE_INTER, (0, -1, -1), 3)
    with pytest.raises(hip.Vp9HipError, match="threshold"):
        dec.run(hip.PHASE_LF, (-1, -1, -1), 3)
    dec.close()
