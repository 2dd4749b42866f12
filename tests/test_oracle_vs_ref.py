"""Oracle (our C restatement) vs the reference's own object code (oracle/_ref/libvpxref.so,
built from /root/reference sources by oracle/Makefile).  Skipped when _ref is not built."""
import ctypes

import numpy as np
import pytest

import refcases
from vp9ref import c_i16p, i32p, i64p, ptr_at, u8p, u16p


def test_idct_iadst_1d(oracle, ref):
    rng = np.random.default_rng(1)
    for n, names in [(4, ("idct4_c", "iadst4_c")), (8, ("idct8_c", "iadst8_c")), (16, ("idct16_c", "iadst16_c")),
                     (32, ("idct32_c", None))]:
        for it in range(1500):
            kind = it % 4
            if kind == 0:
                x = rng.integers(-32768, 32768, n).astype(np.int32)
            elif kind == 1:
                x = rng.integers(-2000, 2000, n).astype(np.int32)
            elif kind == 2:
                x = np.zeros(n, np.int32)
                x[rng.integers(0, n)] = rng.choice([-32768, 32767])
            else:
                x = (rng.integers(-32768, 32768, n) * (rng.random(n) < 0.2)).astype(np.int32)
            for j, nm in enumerate(names):
                if nm is None:
                    continue
                o, r = np.zeros(n, np.int32), np.zeros(n, np.int32)
                getattr(ref, nm)(i32p(x), i32p(r))
                (oracle.vp9o_idct1d if j == 0 else oracle.vp9o_iadst1d)(n, i32p(x), i32p(o), 0)
                assert np.array_equal(o, r), (nm, x)
            if n < 32:
                bd = [8, 10, 12][it % 3]
                lim = 1 << (bd + 8)
                x = (rng.integers(-lim, lim, n) if it % 4 else rng.integers(-(1 << 26), 1 << 26, n)).astype(np.int32)
                for j, nm in enumerate(("vpx_highbd_idct%d_c" % n, "vpx_highbd_iadst%d_c" % n)):
                    o, r = np.zeros(n, np.int32), np.zeros(n, np.int32)
                    getattr(ref, nm)(i32p(x), i32p(r), bd)
                    (oracle.vp9o_idct1d if j == 0 else oracle.vp9o_iadst1d)(n, i32p(x), i32p(o), 1)
                    assert np.array_equal(o, r), (nm, x)


def test_inverse_transforms_2d(oracle, ref):
    """every vpx_idctNxN_*_add_c, vp9_iht*_add_c, vpx_iwht4x4_*_add_c and highbd twin.
    The fork's full highbd _add_c functions store the residual into tran_high_t instead of
    adding (inv_txfm.c:1450-1471, 1638-1659 ...): we add and clip their output ourselves.
    Its edited vpx_highbd_iwht4x4_16_add_c writes overlapping rows (:1346-1352, a fork bug), so
    that one function is not used as a pin."""
    rng = np.random.default_rng(2)
    for n in (4, 8, 16, 32):
        for tag in refcases.TXFM_VARIANTS[n]:
            for it in range(120):
                c = refcases.txfm_coeffs(rng, n, tag, it % 3, 32768)
                d0 = rng.integers(0, 256, (n, n + 5)).astype(np.uint8)
                dr, do = d0.copy(), d0.copy()
                getattr(ref, "vpx_idct%dx%d_%d_add_c" % (n, n, tag))(i32p(c), u8p(dr), n + 5)
                oracle.vp9o_inv_txfm_add(n, 0, 0, i32p(c), u8p(do), n + 5, tag)
                assert np.array_equal(dr, do), (n, tag, it)
                bd = [8, 10, 12][it % 3]
                c = refcases.txfm_coeffs(rng, n, tag, it % 3, 1 << (bd + 8))
                h0 = rng.integers(0, 1 << bd, (n, n + 3)).astype(np.uint16)
                ho = h0.copy()
                oracle.vp9o_highbd_inv_txfm_add(n, 0, 0, i32p(c), u16p(ho), n + 3, tag, bd)
                if tag == n * n:
                    res = np.zeros((n, n + 3), np.int64)
                    getattr(ref, "vpx_highbd_idct%dx%d_%d_add_c" % (n, n, tag))(i32p(c), i64p(res), n + 3, bd)
                    hr = np.clip(h0.astype(np.int64) + res, 0, (1 << bd) - 1).astype(np.uint16)
                else:
                    hr = h0.copy()
                    getattr(ref, "vpx_highbd_idct%dx%d_%d_add_c" % (n, n, tag))(i32p(c), u16p(hr), n + 3, bd)
                assert np.array_equal(hr, ho), (n, tag, it, bd)
        if n < 32:
            for tx in range(4):
                for it in range(120):
                    c = refcases.txfm_coeffs(rng, n, n * n, it % 3, 32768 if it % 2 else 4096)
                    d0 = rng.integers(0, 256, (n, n + 5)).astype(np.uint8)
                    dr, do = d0.copy(), d0.copy()
                    getattr(ref, "vp9_iht%dx%d_%d_add_c" % (n, n, n * n))(i32p(c), u8p(dr), n + 5, tx)
                    oracle.vp9o_inv_txfm_add(n, tx, 0, i32p(c), u8p(do), n + 5, n * n)
                    assert np.array_equal(dr, do), (n, tx, it)
                    bd = [8, 10, 12][it % 3]
                    c = refcases.txfm_coeffs(rng, n, n * n, it % 3, 1 << (bd + 8))
                    h0 = rng.integers(0, 1 << bd, (n, n + 3)).astype(np.uint16)
                    ho = h0.copy()
                    oracle.vp9o_highbd_inv_txfm_add(n, tx, 0, i32p(c), u16p(ho), n + 3, n * n, bd)
                    res = np.zeros((n, n + 3), np.int64)
                    getattr(ref, "vp9_highbd_iht%dx%d_%d_add_c" % (n, n, n * n))(i32p(c), i64p(res), n + 3, tx, bd)
                    hr = np.clip(h0.astype(np.int64) + res, 0, (1 << bd) - 1).astype(np.uint16)
                    assert np.array_equal(hr, ho), (n, tx, it)
    for it in range(300):
        c = rng.integers(-32768, 32768, (4, 4)).astype(np.int32)
        for eob, nm in ((16, "vpx_iwht4x4_16_add_c"), (1, "vpx_iwht4x4_1_add_c")):
            d0 = rng.integers(0, 256, (4, 9)).astype(np.uint8)
            dr, do = d0.copy(), d0.copy()
            getattr(ref, nm)(i32p(c), u8p(dr), 9)
            oracle.vp9o_inv_txfm_add(4, 0, 1, i32p(c), u8p(do), 9, eob)
            assert np.array_equal(dr, do)
        h0 = rng.integers(0, 1024, (4, 9)).astype(np.uint16)
        hr, ho = h0.copy(), h0.copy()
        ref.vpx_highbd_iwht4x4_1_add_c(i32p(c), u16p(hr), 9, 10)
        oracle.vp9o_highbd_inv_txfm_add(4, 0, 1, i32p(c), u16p(ho), 9, 1, 10)
        assert np.array_equal(hr, ho)


def test_convolve_family(oracle, ref):
    rng = np.random.default_rng(3)
    kern_tab = (ctypes.c_void_p * 5).in_dll(ref, "vp9_filter_kernels")
    for f in range(5):
        a = np.ctypeslib.as_array(ctypes.cast(kern_tab[f], c_i16p), (128,))
        b = np.ctypeslib.as_array(ctypes.cast(oracle.vp9o_filter_kernels(f), c_i16p), (128,))
        assert np.array_equal(a, b)
    for it in range(1500):
        c = refcases.conv_case(rng, it)
        dr, do = c["dst"].copy(), c["dst"].copy()
        sp = ptr_at(c["src"], 8, 8)
        W = c["src"].shape[1]
        kp, ko = ctypes.c_void_p(kern_tab[c["filt"]]), ctypes.c_void_p(oracle.vp9o_filter_kernels(c["filt"]))
        name = refcases.CONV_NAMES[c["mode"]]
        if c["hbd"]:
            getattr(ref, "vpx_highbd_" + name + "_c")(sp, W, u16p(dr), 80, kp, c["x0"], c["xs"], c["y0"], c["ys"],
                                                     c["w"], c["h"], c["bd"])
            oracle.vp9o_highbd_convolve(c["mode"], c["scaled"], sp, W, u16p(do), 80, ko, c["x0"], c["xs"], c["y0"],
                                        c["ys"], c["w"], c["h"], c["bd"])
        else:
            getattr(ref, "vpx_" + name + "_c")(sp, W, u8p(dr), 80, kp, c["x0"], c["xs"], c["y0"], c["ys"], c["w"], c["h"])
            oracle.vp9o_convolve(c["mode"], c["scaled"], sp, W, u8p(do), 80, ko, c["x0"], c["xs"], c["y0"], c["ys"],
                                 c["w"], c["h"])
        assert np.array_equal(dr, do), {k: v for k, v in c.items() if k not in ("src", "dst")}


def test_intra_predictors(oracle, ref):
    rng = np.random.default_rng(4)
    for it in range(150):
        for bs in (4, 8, 16, 32):
            for nm, m in refcases.INTRA_NAMES.items():
                hbd = it % 2
                bd = [8, 10, 12][it % 3] if hbd else 8
                dt = np.uint16 if hbd else np.uint8
                above = rng.integers(0, 1 << bd, (1, 16 + 64)).astype(dt)
                left = rng.integers(0, 1 << bd, 32).astype(dt)
                dr, do = np.zeros((32, 40), dt), np.zeros((32, 40), dt)
                ap = ptr_at(above, 0, 16)
                if hbd:
                    getattr(ref, "vpx_highbd_%s_predictor_%dx%d_c" % (nm, bs, bs))(u16p(dr), 40, ap, u16p(left), bd)
                    oracle.vp9o_highbd_intra_predictor(m, bs, u16p(do), 40, ap, u16p(left), bd)
                else:
                    getattr(ref, "vpx_%s_predictor_%dx%d_c" % (nm, bs, bs))(u8p(dr), 40, ap, u8p(left))
                    oracle.vp9o_intra_predictor(m, bs, u8p(do), 40, ap, u8p(left))
                assert np.array_equal(dr, do), (nm, bs, hbd)


def test_loop_filter_kernels(oracle, ref):
    rng = np.random.default_rng(5)
    changed = 0
    for it in range(2400):
        c = refcases.lpf_case(rng, it)
        ir, io = c["img"].copy(), c["img"].copy()
        th = [np.array([v], np.uint8) for v in c["th"]]
        args = [u8p(t) for t in th]
        rargs = args[:3] if (c["kind"] == 16 or not c["dual"]) else args
        pr, po = ptr_at(ir, 12, 12), ptr_at(io, 12, 12)
        if c["hbd"]:
            getattr(ref, refcases.lpf_name(c))(pr, 40, *rargs, c["bd"])
            oracle.vp9o_highbd_lpf(c["vertical"], c["kind"], c["dual"], po, 40, *args, c["bd"])
        else:
            getattr(ref, refcases.lpf_name(c))(pr, 40, *rargs)
            oracle.vp9o_lpf(c["vertical"], c["kind"], c["dual"], po, 40, *args)
        assert np.array_equal(ir, io), refcases.lpf_name(c)
        changed += int((ir != c["img"]).any())
    assert changed > 400
