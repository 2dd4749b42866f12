#!/usr/bin/env python3
"""Generates tests/golden/intra_e.npz: the four 4x4 predictors of the dispatch table that VP9 itself never
selects (vpx_d45e / d63e / he / ve_predictor_4x4, libvpx/vpx_dsp/vpx_dsp_rtcd_defs.pl:46, 51, 57, 70) run
through the REFERENCE's own object code (oracle/_ref/libvpxref.so).  Build container only; the arrays are
committed.     python tests/golden/make_intra_e.py"""
import ctypes
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from vp9ref import load_ref, ptr_at, u8p  # noqa: E402

ref = load_ref()
rng = np.random.default_rng(46515770)
N = 24
above = rng.integers(0, 256, (N, 1, 32)).astype(np.uint8)  # above[-1] at column 15, above[0..7] at 16..23
left = rng.integers(0, 256, (N, 4)).astype(np.uint8)
out = {}
for nm in ("d45e", "d63e", "he", "ve"):
    res = np.zeros((N, 4, 4), np.uint8)
    for i in range(N):
        dst = np.zeros((4, 4), np.uint8)
        getattr(ref, f"vpx_{nm}_predictor_4x4_c")(u8p(dst), ctypes.c_ssize_t(4), ptr_at(above[i], 0, 16), u8p(left[i]))
        res[i] = dst
    out[nm] = res
np.savez_compressed(os.path.join(HERE, "intra_e.npz"), above=above, left=left, **out)
print("wrote intra_e.npz")
