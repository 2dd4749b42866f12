"""refframe.py — one whole frame through the REFERENCE's own C functions (oracle/_ref/libvpxref.so,
ref_recon_frame in oracle/ref_frame_driver.c): the expected frame of the GPU frame tests / bench.py and
bench.py's cpu_baseline (kind "reference").  Independent of the product's packers: the input is the list
of decoded blocks + the coefficients in the reference's layout, exactly what the product is given.
Test infrastructure."""
import ctypes
import hashlib
import os
import time

import numpy as np

import blockgen

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BORDER = 192  # >= libvpx's VP9_ENC_BORDER_IN_PIXELS (160) + the 8-tap footprint


def load_ref():
    path = os.path.join(ROOT, "oracle", "_ref", "libvpxref.so")
    if not os.path.exists(path):
        raise RuntimeError(f"{path} missing: run `make -C oracle` where /root/reference is present")
    lib = ctypes.CDLL(path)
    lib.ref_recon_frame.restype = ctypes.c_int
    return lib


def plane_dims(W, H):
    aw, ah = (W + 7) & ~7, (H + 7) & ~7
    return ([(aw, ah), (aw // 2, ah // 2), (aw // 2, ah // 2)],
            [(W, H), ((W + 1) // 2, (H + 1) // 2), ((W + 1) // 2, (H + 1) // 2)])


def _ptr(arr, r, c):
    return arr.ctypes.data + (r * arr.shape[1] + c) * arr.itemsize


class RefFrame:
    """Inputs of one frame prepared once (bordered references, records), run() any number of times."""

    def __init__(self, ref, blocks, W, H, bd, refs, ref_sizes, coef, eob, tiles=0, lossless=0, filter=True, sharp=0):
        self.ref, self.W, self.H, self.bd = ref, W, H, bd
        self.dims, self.crop = plane_dims(W, H)
        dt = np.uint16 if bd > 8 else np.uint8
        self.recs = blockgen.to_ref_records(blocks)
        self.bref = []
        for planes, (rw, rh) in zip(refs, ref_sizes):
            _, rc = plane_dims(rw, rh)
            self.bref += [np.ascontiguousarray(np.pad(np.asarray(pl, dt)[:c[1], :c[0]], BORDER, mode="edge"))
                          for pl, c in zip(planes, rc)]
        self.cur = [np.zeros((d[1] + 2 * BORDER, d[0] + 2 * BORDER), dt) for d in self.dims]
        self.coef = [np.ascontiguousarray(c, np.int32) for c in coef]
        self.eob = [np.ascontiguousarray(e, np.int32) for e in eob]
        self.args = (
            self.recs.ctypes.data_as(ctypes.c_void_p), len(self.recs), W, H, 1, bd, int(bd > 8),
            (ctypes.c_void_p * 3)(*[_ptr(a, BORDER, BORDER) for a in self.cur]),
            (ctypes.c_int * 3)(*[a.shape[1] for a in self.cur]),
            (ctypes.c_void_p * 9)(*[_ptr(a, BORDER, BORDER) for a in self.bref]),
            (ctypes.c_int * 9)(*[a.shape[1] for a in self.bref]),
            (ctypes.c_int * 3)(*[s[0] for s in ref_sizes]), (ctypes.c_int * 3)(*[s[1] for s in ref_sizes]),
            (ctypes.c_void_p * 3)(*[c.ctypes.data if len(c) else None for c in self.coef]),
            (ctypes.c_void_p * 3)(*[e.ctypes.data for e in self.eob]),
            (ctypes.c_int * 3)(*[e.shape[1] for e in self.eob]),
            tiles, int(lossless), int(filter), sharp)

    def run(self):
        """Reconstructs the frame; returns seconds."""
        for a in self.cur:
            a[:] = 0
        t0 = time.perf_counter()
        rc = self.ref.ref_recon_frame(*self.args)
        dt = time.perf_counter() - t0
        if rc != 0:
            raise RuntimeError(f"ref_recon_frame failed ({rc})")
        return dt

    def planes(self):
        return [a[BORDER:BORDER + d[1], BORDER:BORDER + d[0]].copy() for a, d in zip(self.cur, self.dims)]


def frame_md5(planes, W, H):
    """vpxdec --md5 of an i420 frame (libvpx/vpxdec.c:285-302): rows of the crop rectangle, Y U V."""
    _, crop = plane_dims(W, H)
    h = hashlib.md5()
    for a, (w, hh) in zip(planes, crop):
        h.update(np.ascontiguousarray(a[:hh, :w]).tobytes())
    return h.hexdigest()
