"""Known-answer pin of the oracle's intra predictors against the MD5 signatures libvpx's own
test holds (test/test_intra_pred_speed.cc), reproduced with its ACMRandom input recipe.
The same vectors are pushed through the HIP rtcd twins in the gpu-marked test."""
import hashlib
import json
import os

import numpy as np
import pytest

from vp9ref import ACMRandom, ptr_at, u8p, u16p

HERE = os.path.dirname(os.path.abspath(__file__))
KAT = json.load(open(os.path.join(HERE, "golden", "intra_pred_kat.json")))
# kVp9IntraPredNames order -> oracle mode ids (vp9_oracle.h)
MODE_IDS = [0, 11, 12, 10, 1, 2, 3, 4, 5, 6, 7, 8, 9]
KBPS = 32


def kat_inputs(bs, bd, dtype):
    """IntraPredTestMem::Init (test_intra_pred_speed.cc:43-58)"""
    rnd = ACMRandom(0xbaba)
    mask = (1 << bd) - 1
    ref_src = np.array([rnd.rand16() & mask for _ in range(KBPS * KBPS)], dtype).reshape(KBPS, KBPS)
    left = np.array([rnd.rand16() & mask for _ in range(KBPS)], dtype)
    above_mem = np.zeros((1, 2 * KBPS + 16), dtype)
    for i in range(-1, KBPS):
        above_mem[0, 16 + i] = rnd.rand16() & mask
    for i in range(bs, 2 * KBPS):
        above_mem[0, 16 + i] = above_mem[0, 16 + bs - 1]
    return ref_src, left, above_mem


@pytest.mark.parametrize("bs", [4, 8, 16, 32])
def test_oracle_lowbd_predictors_match_libvpx_md5(oracle, bs):
    ref_src, left, above_mem = kat_inputs(bs, 8, np.uint8)
    for k, mode in enumerate(MODE_IDS):
        src = ref_src.copy()
        oracle.vp9o_intra_predictor(mode, bs, u8p(src), KBPS, ptr_at(above_mem, 0, 16), u8p(left))
        assert hashlib.md5(src.tobytes()).hexdigest() == KAT["lowbd"][str(bs)][k], (bs, KAT["modes"][k])


@pytest.mark.parametrize("bs", [4, 8, 16, 32])
def test_oracle_highbd_predictors_match_libvpx_md5(oracle, bs):
    ref_src, left, above_mem = kat_inputs(bs, 12, np.uint16)
    for k, mode in enumerate(MODE_IDS):
        src = ref_src.copy()
        oracle.vp9o_highbd_intra_predictor(mode, bs, u16p(src), KBPS, ptr_at(above_mem, 0, 16), u16p(left), 12)
        assert hashlib.md5(src.tobytes()).hexdigest() == KAT["highbd12"][str(bs)][k], (bs, KAT["modes"][k])
