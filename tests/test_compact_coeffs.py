"""The compact coefficient layout (vp9hip_coeff_layout.compact, oracle/patch_decodeframe.py E12): a transform
block's slot holds only the rows the reference's clearing rule (vp9_decodeframe.c:960-967) leaves non-zero.
CPU: the extent rule against its Python restatement and the packer's offsets into caller-placed compact slots.
GPU: the frame driver fed compact slots gives the frame it gives from the reference's full slots (= the oracle's)."""
import numpy as np
import pytest

import blockgen


def _params(hip, W, H, bd, lossless=0):
    P = hip.FrameParams()
    P.width, P.height, P.ss_x, P.ss_y, P.bit_depth, P.hbd = W, H, 1, 1, bd, int(bd > 8)
    P.build_lf_masks, P.lossless = 1, lossless
    for k in range(3):
        P.ref_width[k], P.ref_height[k] = W, H
    return P


def test_extent_rule_matches_restatement(hip):
    lib = hip.lib()
    for tx in range(4):
        n = 4 << tx
        for tt in range(4):
            for eob in list(range(0, 40)) + [135, 136, n * n]:
                if eob > n * n:
                    continue
                rows = blockgen.coeff_rows(eob, tt == 0, tx)
                assert lib.vp9hip_coeff_rows(eob, tt, tx) == rows
                assert lib.vp9hip_coeff_extent(eob, tt, tx) == rows * n
    # the reference's three clears, literally: 1 coefficient; 4 * (4 << tx_size); 256; 16 << (tx_size << 1)
    assert lib.vp9hip_coeff_extent(10, 0, 2) == 4 * (4 << 2) and lib.vp9hip_coeff_extent(34, 0, 3) == 256
    assert lib.vp9hip_coeff_extent(11, 0, 2) == 16 << (2 << 1) and lib.vp9hip_coeff_extent(10, 1, 1) == 64


@pytest.mark.parametrize("W,H,kw,lossless", [(352, 288, dict(intra_frac=0.3), False), (200, 136, dict(all_intra=True), False),
                                             (136, 72, dict(intra_frac=0.2), True)])
def test_packer_offsets_into_compact_slots(hip, W, H, kw, lossless):
    rng = np.random.default_rng(W + H)
    blocks = blockgen.gen_blocks(rng, W, H, hip.BLOCK_DTYPE, **kw)
    if lossless:
        blocks["tx_size"] = 0
    coef, eob = blockgen.gen_coeffs(rng, blocks, W, H, 8, lossless=lossless)
    coef_c, layout = blockgen.compact_layout(blocks, coef, eob, W, H, lossless=lossless)
    assert sum(len(c) for c in coef_c) < 0.8 * sum(len(c) for c in coef)
    P = _params(hip, W, H, 8, int(lossless))
    pk = hip.Packer()
    full = pk.pack(P, blocks, eob)
    comp = pk.pack(P, blocks, eob, tile_layout=layout)
    allf = np.concatenate(coef)
    allc = np.concatenate(coef_c)
    assert comp["coeff_total"] == len(allc)
    for key in ("txb", "intra_decode_order"):
        a, b = full[key], comp[key]
        assert len(a) == len(b)
        for f in a.dtype.names:
            if f != "coeff_off":
                assert np.array_equal(a[f], b[f]), (key, f)
        for ra, rb in zip(a, b):
            if ra["eob"] == 0:
                continue
            tx = int(ra["tx_size"])
            ext = hip.lib().vp9hip_coeff_extent(int(ra["eob"]), 0 if lossless else int(ra["tx_type"]) & 3, tx)
            fo, co = int(ra["coeff_off"]), int(rb["coeff_off"])
            assert np.array_equal(allf[fo:fo + ext], allc[co:co + ext])
    # a slot that would end past the buffer is refused, not read
    bad = dict(layout, total=layout["total"] // 2)
    with pytest.raises(hip.Vp9HipError, match="past the buffer"):
        pk.pack(P, blocks, eob, tile_layout=bad)
    pk.close()


@pytest.mark.gpu
@pytest.mark.parametrize("W,H,bd,kw", [(352, 288, 8, dict(intra_frac=0.25)), (330, 250, 10, dict(intra_frac=0.5)),
                                        (1280, 720, 8, dict(intra_frac=0.1))])
def test_decoder_compact_slots_same_frame(hip, W, H, bd, kw):
    import ctypes
    import workload
    rng = np.random.default_rng(W * 3 + bd)
    dt = np.uint16 if bd > 8 else np.uint8
    aw, ah = (W + 7) & ~7, (H + 7) & ~7
    dims = [(aw, ah), (aw // 2, ah // 2), (aw // 2, ah // 2)]
    blocks = blockgen.gen_blocks(rng, W, H, hip.BLOCK_DTYPE, **kw)
    coef, eob = blockgen.gen_coeffs(rng, blocks, W, H, bd)
    coef_c, layout = blockgen.compact_layout(blocks, coef, eob, W, H)
    refs = [[np.ascontiguousarray(workload.smooth_noise(rng, d[1], d[0], bd, sigma=1.5 + k).astype(dt)) for d in dims]
            for k in range(3)]
    P = _params(hip, W, H, bd)
    th = hip.LfThresh()
    hip.lib().vp9hip_lf_frame_init(32, 0, None, None, 0, 0, None, None, None, ctypes.byref(th))
    dec = hip.Decoder(0)
    for k in range(3):
        dec.upload(k, refs[k], W, H, bd)
    outs = []
    for slot, (c, tl) in enumerate(((coef, None), (coef_c, layout))):
        dec.alloc_slot(3 + slot, W, H, bd)
        dec.begin_frame(P, blocks, eob, c, tile_layout=tl)
        dec.run(hip.PHASE_INTER | hip.PHASE_INTRA | hip.PHASE_LF, (0, 1, 2), 3 + slot, thresh=th)
        dec.sync()
        got = [np.zeros((d[1], d[0]), dt) for d in dims]
        dec.download(3 + slot, got, W, H, bd)
        outs.append(got)
    dec.close()
    for p in range(3):
        assert np.array_equal(outs[0][p], outs[1][p]), f"plane {p}: compact slots give another frame"
