"""Whole frames on the CPU, two independent ways:
  (a) the REFERENCE's own C functions walking the decoded blocks in decode order (ref_recon_frame,
      oracle/ref_frame_driver.c: vp9_build_inter_predictors_sb, vp9_predict_intra_block, the reference's
      inverse transforms, vp9_build_mask / vp9_adjust_mask / vp9_filter_block_plane_*);
  (b) the PRODUCT's C packer (vp9hip_pack_frame) -> work lists -> the oracle's block functions in list order.
Equal frames pin packer + oracle together against the reference at frame level, loop filter included
(with libvpx's `eobtotal == 0 -> skip`), without a GPU.  The GPU frame tests and bench.py compare the
HIP output with (a)."""
import numpy as np
import pytest

import blockgen
import refframe
from test_gpu_decoder import _oracle_frame, _params, _thresholds


@pytest.mark.parametrize("W,H,bd,tiles,sharp,kw", [
    (352, 288, 8, 0, 0, {}),
    (330, 250, 10, 1, 3, dict(intra_frac=0.4)),
    (200, 136, 12, 0, 6, dict(compound_frac=0.5)),
    (640, 360, 8, 2, 0, dict(levels=(0, 8, 30, 63), skip_frac=0.1)),
    (256, 256, 8, 0, 0, dict(all_intra=True)),
    (72, 40, 8, 0, 0, {}),
    # most transform blocks uncoded: many inter blocks >= 8x8 end up with eobtotal == 0 and lose their inner edges
    (640, 384, 8, 1, 0, dict(skip_frac=0.0, intra_frac=0.05, zero=0.85)),
])
def test_reference_walk_equals_packed_lists(hip, oracle, ref, W, H, bd, tiles, sharp, kw):
    import workload
    rng = np.random.default_rng(W * 3 + H + bd)
    dt = np.uint16 if bd > 8 else np.uint8
    dims, _ = refframe.plane_dims(W, H)
    kw = dict(kw)
    zero = kw.pop("zero", 0.2)
    blocks = blockgen.gen_blocks(rng, W, H, hip.BLOCK_DTYPE, **kw)
    coef, eob = blockgen.gen_coeffs(rng, blocks, W, H, bd, zero_frac=zero)
    refs = [[np.ascontiguousarray(workload.smooth_noise(rng, d[1], d[0], bd, sigma=1.5 + k).astype(dt)) for d in dims]
            for k in range(3)]
    P = _params(hip, W, H, bd, tiles)
    th = _thresholds(hip, sharp)
    packed = _oracle_frame(hip, oracle, P, blocks, coef, eob, refs, W, H, bd, th)
    rf = refframe.RefFrame(refframe.load_ref(), blocks, W, H, bd, refs, [(W, H)] * 3, coef, eob, tiles=tiles, sharp=sharp)
    rf.run()
    walked = rf.planes()
    for p in range(3):
        bad = np.argwhere(packed[p] != walked[p])
        assert bad.size == 0, f"plane {p}: {len(bad)} samples differ, first at {bad[:5].tolist()}"
