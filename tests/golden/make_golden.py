#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the REFERENCE's own object code (oracle/_ref/libvpxref.so,
built by `make -C oracle ref` from /root/reference sources).  Run in the build container only;
the .npz files (inputs + expected outputs, plain arrays) are committed so that the oracle stays
pinned where neither /root/reference nor oracle/_ref exist.

    python tests/golden/make_golden.py
"""
import ctypes
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import refcases  # noqa: E402
from vp9ref import c_i16p, i32p, i64p, load_ref, ptr_at, u8p, u16p  # noqa: E402


def main():
    ref = load_ref()
    rng = np.random.default_rng(20261004)
    # ---- inverse transforms: every 8-bit variant + highbd (residual-store forms added here)
    tx = {}
    k = 0
    for n in (4, 8, 16, 32):
        for tag in refcases.TXFM_VARIANTS[n]:
            for it in range(6):
                c = refcases.txfm_coeffs(rng, n, tag, it % 3, 32768)
                d0 = rng.integers(0, 256, (n, n)).astype(np.uint8)
                d = d0.copy()
                getattr(ref, "vpx_idct%dx%d_%d_add_c" % (n, n, tag))(i32p(c), u8p(d), n)
                tx["c%d" % k], tx["in%d" % k], tx["out%d" % k] = c, d0, d
                tx["meta%d" % k] = np.array([n, 0, 0, tag, 8, 0], np.int32)  # n, tx_type, lossless, eob, bd, hbd
                k += 1
        if n < 32:
            for t in range(1, 4):
                for it in range(4):
                    c = refcases.txfm_coeffs(rng, n, n * n, it % 3, 32768 if it % 2 else 4096)
                    d0 = rng.integers(0, 256, (n, n)).astype(np.uint8)
                    d = d0.copy()
                    getattr(ref, "vp9_iht%dx%d_%d_add_c" % (n, n, n * n))(i32p(c), u8p(d), n, t)
                    tx["c%d" % k], tx["in%d" % k], tx["out%d" % k] = c, d0, d
                    tx["meta%d" % k] = np.array([n, t, 0, n * n, 8, 0], np.int32)
                    k += 1
                    bd = [10, 12][it % 2]
                    c = refcases.txfm_coeffs(rng, n, n * n, it % 3, 1 << (bd + 8))
                    h0 = rng.integers(0, 1 << bd, (n, n)).astype(np.uint16)
                    res = np.zeros((n, n), np.int64)
                    getattr(ref, "vp9_highbd_iht%dx%d_%d_add_c" % (n, n, n * n))(i32p(c), i64p(res), n, t, bd)
                    tx["c%d" % k], tx["in%d" % k] = c, h0
                    tx["out%d" % k] = np.clip(h0.astype(np.int64) + res, 0, (1 << bd) - 1).astype(np.uint16)
                    tx["meta%d" % k] = np.array([n, t, 0, n * n, bd, 1], np.int32)
                    k += 1
    for it in range(6):
        c = rng.integers(-32768, 32768, (4, 4)).astype(np.int32)
        for eob, nm in ((16, "vpx_iwht4x4_16_add_c"), (1, "vpx_iwht4x4_1_add_c")):
            d0 = rng.integers(0, 256, (4, 4)).astype(np.uint8)
            d = d0.copy()
            getattr(ref, nm)(i32p(c), u8p(d), 4)
            tx["c%d" % k], tx["in%d" % k], tx["out%d" % k] = c, d0, d
            tx["meta%d" % k] = np.array([4, 0, 1, eob, 8, 0], np.int32)
            k += 1
    tx["count"] = np.array([k], np.int32)
    np.savez_compressed(os.path.join(HERE, "txfm.npz"), **tx)
    # ---- convolve
    kern_tab = (ctypes.c_void_p * 5).in_dll(ref, "vp9_filter_kernels")
    cv, k = {}, 0
    for it in range(60):
        c = refcases.conv_case(rng, it)
        d = c["dst"].copy()
        sp = ptr_at(c["src"], 8, 8)
        W = c["src"].shape[1]
        kp = ctypes.c_void_p(kern_tab[c["filt"]])
        name = refcases.CONV_NAMES[c["mode"]]
        if c["hbd"]:
            getattr(ref, "vpx_highbd_" + name + "_c")(sp, W, u16p(d), 80, kp, c["x0"], c["xs"], c["y0"], c["ys"],
                                                     c["w"], c["h"], c["bd"])
        else:
            getattr(ref, "vpx_" + name + "_c")(sp, W, u8p(d), 80, kp, c["x0"], c["xs"], c["y0"], c["ys"], c["w"], c["h"])
        cv["src%d" % k], cv["in%d" % k], cv["out%d" % k] = c["src"], c["dst"], d
        cv["meta%d" % k] = np.array([c["mode"], c["scaled"], c["filt"], c["x0"], c["xs"], c["y0"], c["ys"], c["w"],
                                     c["h"], c["bd"], c["hbd"]], np.int32)
        k += 1
    cv["count"] = np.array([k], np.int32)
    np.savez_compressed(os.path.join(HERE, "convolve.npz"), **cv)
    # ---- loop filter kernels
    lf, k = {}, 0
    for it in range(96):
        c = refcases.lpf_case(rng, it)
        img = c["img"].copy()
        th = [np.array([v], np.uint8) for v in c["th"]]
        args = [u8p(t) for t in th]
        rargs = args[:3] if (c["kind"] == 16 or not c["dual"]) else args
        if c["hbd"]:
            getattr(ref, refcases.lpf_name(c))(ptr_at(img, 12, 12), 40, *rargs, c["bd"])
        else:
            getattr(ref, refcases.lpf_name(c))(ptr_at(img, 12, 12), 40, *rargs)
        lf["in%d" % k], lf["out%d" % k], lf["th%d" % k] = c["img"], img, c["th"]
        lf["meta%d" % k] = np.array([c["vertical"], c["kind"], c["dual"], c["bd"], c["hbd"]], np.int32)
        k += 1
    lf["count"] = np.array([k], np.int32)
    np.savez_compressed(os.path.join(HERE, "lpf.npz"), **lf)
    print("wrote txfm.npz convolve.npz lpf.npz")


if __name__ == "__main__":
    main()
