"""Whole-frame checker: runs a synthetic workload through the oracle (sequential CPU) — used
by the GPU parity tests, smoke() and bench.py's cpu_baseline.  Test infrastructure."""
import ctypes
import hashlib
import os
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OFrame(ctypes.Structure):
    _fields_ = [("plane", ctypes.c_void_p * 3), ("stride", ctypes.c_int32 * 3), ("width", ctypes.c_int32 * 3),
                ("height", ctypes.c_int32 * 3), ("awidth", ctypes.c_int32 * 3), ("aheight", ctypes.c_int32 * 3),
                ("bit_depth", ctypes.c_int32), ("hbd", ctypes.c_int32)]


class OThresh(ctypes.Structure):
    _fields_ = [("mblim", ctypes.c_uint8 * 64), ("lim", ctypes.c_uint8 * 64), ("hev_thr", ctypes.c_uint8 * 64)]


def load_oracle():
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
    return lib


def _oframe(planes, wl, pad):
    f = OFrame()
    for p, a in enumerate(planes):
        f.plane[p] = a.ctypes.data
        f.stride[p] = a.shape[1]
        f.width[p], f.height[p] = wl["crop"][p]
        f.awidth[p], f.aheight[p] = wl["dims"][p]
    f.bit_depth, f.hbd = wl["bd"], int(wl["hbd"])
    return f


def oracle_frame(oracle, wl, phases=("inter", "txb", "intra", "lf")):
    """Sequential CPU reconstruction; returns (planes, seconds per phase)."""
    dt = np.uint16 if wl["hbd"] else np.uint8
    PAD = 16  # the loop filter touches half-outside chroma segments (libvpx has a border there)
    bufs = [np.zeros((ah + PAD, aw + PAD), dt) for (aw, ah) in wl["dims"]]
    dst = _oframe(bufs, wl, PAD)
    refs_arr = (OFrame * len(wl["refs"]))(*[_oframe([np.ascontiguousarray(p) for p in r], wl, 0) for r in wl["refs"]])
    keep = [[np.ascontiguousarray(p) for p in r] for r in wl["refs"]]
    for i, r in enumerate(keep):
        for p, a in enumerate(r):
            refs_arr[i].plane[p] = a.ctypes.data
            refs_arr[i].stride[p] = a.shape[1]
    coeffs = wl["coeffs"]
    cp = coeffs.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
    times = {}
    t0 = time.perf_counter()
    if "inter" in phases and len(wl["inter_tasks"]):
        oracle.vp9o_recon_inter_list(wl["inter_tasks"].ctypes.data_as(ctypes.c_void_p), len(wl["inter_tasks"]),
                                     refs_arr, ctypes.byref(dst))
    times["inter"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    if "txb" in phases and len(wl["txb"]):
        oracle.vp9o_recon_txb_list(wl["txb"].ctypes.data_as(ctypes.c_void_p), len(wl["txb"]), cp, ctypes.byref(dst))
    times["txb"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    if "intra" in phases and len(wl["intra_decode_order"]):
        oracle.vp9o_recon_intra_list(wl["intra_decode_order"].ctypes.data_as(ctypes.c_void_p),
                                     len(wl["intra_decode_order"]), cp, ctypes.byref(dst))
    times["intra"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    if "lf" in phases:
        th = OThresh()
        mblim, lim, hev = wl["thresholds"]
        for i in range(64):
            th.mblim[i], th.lim[i], th.hev_thr[i] = int(mblim[i]), int(lim[i]), int(hev[i])
        ptrs = (ctypes.c_void_p * 3)(*[b.ctypes.data for b in bufs])
        strides = (ctypes.c_int * 3)(*[b.shape[1] for b in bufs])
        oracle.vp9o_loop_filter_frame(wl["lfm"].ctypes.data_as(ctypes.c_void_p), wl["sb_rows"], wl["sb_cols"],
                                      ctypes.byref(th), ptrs, strides, wl["dims"][0][1] // 8, wl["bd"], int(wl["hbd"]), 3)
    times["lf"] = time.perf_counter() - t0
    planes = [b[:ah, :aw].copy() for b, (aw, ah) in zip(bufs, wl["dims"])]
    return planes, times


def frame_md5(planes, wl):
    """MD5 over the Y, U, V rows of the crop rectangle — vpxdec's --md5 of an i420 frame
    (libvpx/vpxdec.c:285-302: d_w x bytes-per-sample per row, planes in order)."""
    h = hashlib.md5()
    for p, a in enumerate(planes):
        w, hh = wl["crop"][p]
        h.update(np.ascontiguousarray(a[:hh, :w]).tobytes())
    return h.hexdigest()
