"""Oracle vs committed golden vectors (tests/golden/*.npz, produced by make_golden.py from the
reference's own object code).  Runs anywhere: needs neither /root/reference nor oracle/_ref."""
import ctypes
import os

import numpy as np

from vp9ref import i32p, ptr_at, u8p, u16p

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_txfm_golden(oracle):
    z = np.load(os.path.join(G, "txfm.npz"))
    for k in range(int(z["count"][0])):
        n, tx_type, lossless, eob, bd, hbd = [int(v) for v in z["meta%d" % k]]
        c = np.ascontiguousarray(z["c%d" % k])
        d = np.ascontiguousarray(z["in%d" % k]).copy()
        if hbd:
            oracle.vp9o_highbd_inv_txfm_add(n, tx_type, lossless, i32p(c), u16p(d), n, eob, bd)
        else:
            oracle.vp9o_inv_txfm_add(n, tx_type, lossless, i32p(c), u8p(d), n, eob)
        assert np.array_equal(d, z["out%d" % k]), (k, n, tx_type, eob, bd)
    assert k > 100


def test_convolve_golden(oracle):
    z = np.load(os.path.join(G, "convolve.npz"))
    for k in range(int(z["count"][0])):
        mode, scaled, filt, x0, xs, y0, ys, w, h, bd, hbd = [int(v) for v in z["meta%d" % k]]
        src = np.ascontiguousarray(z["src%d" % k])
        d = np.ascontiguousarray(z["in%d" % k]).copy()
        ko = ctypes.c_void_p(oracle.vp9o_filter_kernels(filt))
        sp = ptr_at(src, 8, 8)
        if hbd:
            oracle.vp9o_highbd_convolve(mode, scaled, sp, src.shape[1], u16p(d), 80, ko, x0, xs, y0, ys, w, h, bd)
        else:
            oracle.vp9o_convolve(mode, scaled, sp, src.shape[1], u8p(d), 80, ko, x0, xs, y0, ys, w, h)
        assert np.array_equal(d, z["out%d" % k]), (k, mode, w, h)


def test_lpf_golden(oracle):
    z = np.load(os.path.join(G, "lpf.npz"))
    for k in range(int(z["count"][0])):
        vertical, kind, dual, bd, hbd = [int(v) for v in z["meta%d" % k]]
        img = np.ascontiguousarray(z["in%d" % k]).copy()
        th = [np.array([v], np.uint8) for v in z["th%d" % k]]
        args = [u8p(t) for t in th]
        if hbd:
            oracle.vp9o_highbd_lpf(vertical, kind, dual, ptr_at(img, 12, 12), 40, *args, bd)
        else:
            oracle.vp9o_lpf(vertical, kind, dual, ptr_at(img, 12, 12), 40, *args)
        assert np.array_equal(img, z["out%d" % k]), (k, vertical, kind, dual, bd)
