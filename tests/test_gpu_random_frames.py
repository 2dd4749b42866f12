"""Randomised whole-frame parity through the frame-level driver: random frame sizes, bit depths, tile
counts, intra fractions, filter levels and sharpness, every frame with all three phases (the island walk
and the loop filter side by side when the frame allows it) against the oracle.  The inter-workgroup
hand-offs (loop-filter rows, island -> filter gates) are timing dependent; many different shapes and
island layouts are the way to shake them.  VP9HIP_RANDOM_FRAMES=<n> runs more of them."""
import os

import numpy as np
import pytest

import blockgen
from test_gpu_decoder import _dims, _oracle_frame, _params, _thresholds

pytestmark = pytest.mark.gpu


def test_random_frames_match_oracle(hip, oracle):
    import workload
    n = int(os.environ.get("VP9HIP_RANDOM_FRAMES", "24"))
    rng = np.random.default_rng(20261004)
    dec = hip.Decoder(0)
    for it in range(n):
        W, H = int(rng.integers(64, 1400)), int(rng.integers(64, 800))
        bd = int(rng.choice([8, 8, 10, 12]))
        tiles = int(rng.integers(0, 3)) if W >= 512 * 2 else (1 if W >= 512 else 0)
        sharp = int(rng.integers(0, 8))
        kw = dict(intra_frac=float(rng.choice([0.0, 0.05, 0.15, 0.5])), compound_frac=float(rng.choice([0.0, 0.3])),
                  skip_frac=float(rng.choice([0.1, 0.4, 0.8])), levels=tuple(int(v) for v in rng.choice(64, 4)))
        if rng.random() < 0.1:
            kw = dict(all_intra=True)
            W, H = min(W, 320), min(H, 256)  # key frames take one launch per wave
        dt = np.uint16 if bd > 8 else np.uint8
        dims, crop = _dims(W, H)
        blocks = blockgen.gen_blocks(rng, W, H, hip.BLOCK_DTYPE, **kw)
        coef, eob = blockgen.gen_coeffs(rng, blocks, W, H, bd)
        refs = [[np.ascontiguousarray(workload.smooth_noise(rng, d[1], d[0], bd, sigma=1.5 + k).astype(dt)) for d in dims]
                for k in range(3)]
        P = _params(hip, W, H, bd, tiles)
        th = _thresholds(hip, sharp)
        expect = _oracle_frame(hip, oracle, P, blocks, coef, eob, refs, W, H, bd, th)
        for k in range(3):
            dec.upload(k, refs[k], W, H, bd)
        dec.alloc_slot(3, W, H, bd)
        dec.begin_frame(P, blocks, eob, coef)
        dec.run(hip.PHASE_INTER | hip.PHASE_INTRA | hip.PHASE_LF, (0, 1, 2), 3, thresh=th)
        dec.sync()
        got = [np.zeros((d[1], d[0]), dt) for d in dims]
        dec.download(3, got, W, H, bd)
        for p in range(3):
            bad = np.argwhere(got[p] != expect[p])
            assert bad.size == 0, (f"frame {it} ({W}x{H} bd {bd} tiles {tiles} sharp {sharp} {kw}): plane {p}, {len(bad)} samples "
                                   f"differ, first at {bad[:5].tolist()}")
    dec.close()
