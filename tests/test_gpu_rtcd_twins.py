"""GPU: the block-level `_hip` twins (include/vp9hip_rtcd.h) against
 (a) libvpx's own MD5 known answers for the intra predictors (test_intra_pred_speed.cc),
 (b) the committed golden vectors generated from the reference's object code."""
import ctypes
import hashlib
import os

import numpy as np
import pytest

import refcases
from test_intra_kat import KAT, KBPS, kat_inputs
from vp9ref import i32p, ptr_at, u8p, u16p

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KAT_NAMES = ["dc", "dc_left", "dc_top", "dc_128", "v", "h", "d45", "d135", "d117", "d153", "d207", "d63", "tm"]


def _err(lib):
    lib.vp9hip_rtcd_last_error.restype = ctypes.c_char_p
    return lib.vp9hip_rtcd_last_error().decode()


@pytest.mark.parametrize("bs", [4, 8, 16, 32])
def test_intra_twins_match_libvpx_md5(hip, bs):
    lib = hip.lib()
    ref_src, left, above_mem = kat_inputs(bs, 8, np.uint8)
    for k, nm in enumerate(KAT_NAMES):
        src = ref_src.copy()
        getattr(lib, f"vpx_{nm}_predictor_{bs}x{bs}_hip")(u8p(src), ctypes.c_ssize_t(KBPS), ptr_at(above_mem, 0, 16), u8p(left))
        assert _err(lib) == ""
        assert hashlib.md5(src.tobytes()).hexdigest() == KAT["lowbd"][str(bs)][k], (bs, nm)
    ref_src, left, above_mem = kat_inputs(bs, 12, np.uint16)
    for k, nm in enumerate(KAT_NAMES):
        src = ref_src.copy()
        getattr(lib, f"vpx_highbd_{nm}_predictor_{bs}x{bs}_hip")(u16p(src), ctypes.c_ssize_t(KBPS), ptr_at(above_mem, 0, 16),
                                                              u16p(left), 12)
        assert _err(lib) == ""
        assert hashlib.md5(src.tobytes()).hexdigest() == KAT["highbd12"][str(bs)][k], (bs, nm)


def test_extra_4x4_twins_match_golden(hip):
    """vpx_d45e / d63e / he / ve_predictor_4x4 (never selected by VP9, but prototypes of the dispatch table):
    expected blocks from the reference's object code (tests/golden/make_intra_e.py)."""
    lib = hip.lib()
    g = np.load(os.path.join(G, "intra_e.npz"))
    for nm in ("d45e", "d63e", "he", "ve"):
        for i in range(len(g["above"])):
            dst = np.zeros((4, 4), np.uint8)
            above = np.ascontiguousarray(g["above"][i])
            left = np.ascontiguousarray(g["left"][i])
            getattr(lib, f"vpx_{nm}_predictor_4x4_hip")(u8p(dst), ctypes.c_ssize_t(4), ptr_at(above, 0, 16), u8p(left))
            assert _err(lib) == ""
            assert np.array_equal(dst, g[nm][i]), (nm, i)


def test_txfm_twins_match_golden(hip):
    lib = hip.lib()
    z = np.load(os.path.join(G, "txfm.npz"))
    for k in range(int(z["count"][0])):
        n, tx_type, lossless, eob, bd, hbd = [int(v) for v in z["meta%d" % k]]
        c = np.ascontiguousarray(z["c%d" % k])
        d = np.ascontiguousarray(z["in%d" % k]).copy()
        if lossless:
            name = f"iwht4x4_{eob}_add"
        elif tx_type:
            name = f"iht{n}x{n}_{n * n}_add"
        else:
            name = f"idct{n}x{n}_{eob}_add"
        if tx_type and not lossless:
            if hbd:
                getattr(lib, f"vp9_highbd_{name}_hip")(i32p(c), u16p(d), n, tx_type, bd)
            else:
                getattr(lib, f"vp9_{name}_hip")(i32p(c), u8p(d), n, tx_type)
        else:
            if hbd:
                getattr(lib, f"vpx_highbd_{name}_hip")(i32p(c), u16p(d), n, bd)
            else:
                getattr(lib, f"vpx_{name}_hip")(i32p(c), u8p(d), n)
        assert _err(lib) == ""
        assert np.array_equal(d, z["out%d" % k]), (k, name, bd)


def test_convolve_twins_match_golden(hip, oracle):
    lib = hip.lib()
    z = np.load(os.path.join(G, "convolve.npz"))
    for k in range(int(z["count"][0])):
        mode, scaled, filt, x0, xs, y0, ys, w, h, bd, hbd = [int(v) for v in z["meta%d" % k]]
        src = np.ascontiguousarray(z["src%d" % k])
        d = np.ascontiguousarray(z["in%d" % k]).copy()
        kp = ctypes.c_void_p(oracle.vp9o_filter_kernels(filt))  # same table contents as vp9_filter_kernels[filt]
        sp = ptr_at(src, 8, 8)
        name = refcases.CONV_NAMES[mode]
        if scaled and not hbd and name.startswith("convolve8"):
            name = {"convolve8": "scaled_2d", "convolve8_horiz": "scaled_horiz", "convolve8_vert": "scaled_vert",
                    "convolve8_avg": "scaled_avg_2d", "convolve8_avg_horiz": "scaled_avg_horiz",
                    "convolve8_avg_vert": "scaled_avg_vert"}[name]
        S = ctypes.c_ssize_t
        if hbd:
            getattr(lib, f"vpx_highbd_{name}_hip")(sp, S(src.shape[1]), u16p(d), S(80), kp, x0, xs, y0, ys, w, h, bd)
        else:
            getattr(lib, f"vpx_{name}_hip")(sp, S(src.shape[1]), u8p(d), S(80), kp, x0, xs, y0, ys, w, h)
        assert _err(lib) == "", _err(lib)
        assert np.array_equal(d, z["out%d" % k]), (k, name, w, h, xs, ys, bd)


def test_lpf_twins_match_golden(hip):
    lib = hip.lib()
    z = np.load(os.path.join(G, "lpf.npz"))
    for k in range(int(z["count"][0])):
        vertical, kind, dual, bd, hbd = [int(v) for v in z["meta%d" % k]]
        img = np.ascontiguousarray(z["in%d" % k]).copy()
        th = [np.array([v], np.uint8) for v in z["th%d" % k]]
        args = [u8p(t) for t in th]
        rargs = args[:3] if (kind == 16 or not dual) else args
        name = "vpx_%slpf_%s_%d%s_hip" % ("highbd_" if hbd else "", "vertical" if vertical else "horizontal", kind,
                                          "_dual" if dual else "")
        if hbd:
            getattr(lib, name)(ptr_at(img, 12, 12), 40, *rargs, bd)
        else:
            getattr(lib, name)(ptr_at(img, 12, 12), 40, *rargs)
        assert _err(lib) == ""
        assert np.array_equal(img, z["out%d" % k]), (k, name, bd)
