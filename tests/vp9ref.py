"""ctypes helpers shared by the tests: loaders for the oracle (checker), the reference
build (oracle/_ref, optional) and the product library (libvp9hip.so), plus small numpy
conveniences.  Test infrastructure only."""
import ctypes
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = ctypes.POINTER
c_i32p, c_u8p, c_u16p, c_i16p, c_i64p = (P(ctypes.c_int32), P(ctypes.c_uint8), P(ctypes.c_uint16),
                                         P(ctypes.c_int16), P(ctypes.c_int64))


def i32p(a): return a.ctypes.data_as(c_i32p)
def u8p(a): return a.ctypes.data_as(c_u8p)
def u16p(a): return a.ctypes.data_as(c_u16p)
def i16p(a): return a.ctypes.data_as(c_i16p)
def i64p(a): return a.ctypes.data_as(c_i64p)


def ptr_at(arr, r, c):
    """pointer to element (r, c) of a 2-D array (so negative offsets are legal)."""
    addr = arr.ctypes.data + (r * arr.shape[1] + c) * arr.itemsize
    return ctypes.cast(addr, {1: c_u8p, 2: c_u16p}[arr.itemsize])


def load_oracle(path=None):
    lib = ctypes.CDLL(path or os.path.join(ROOT, "oracle", "liboracle.so"))
    lib.vp9o_filter_kernels.restype = ctypes.c_void_p
    return lib


def load_ref(path=None):
    lib = ctypes.CDLL(path or os.path.join(ROOT, "oracle", "_ref", "libvpxref.so"))
    return lib


def load_hip():
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "cuda_vp9_amd", os.path.join(ROOT, "cuda-vp9_amd", "__init__.py"),
        submodule_search_locations=[os.path.join(ROOT, "cuda-vp9_amd")])
    import sys
    if "cuda_vp9_amd" in sys.modules:
        return sys.modules["cuda_vp9_amd"]
    mod = importlib.util.module_from_spec(spec)
    sys.modules["cuda_vp9_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


class ACMRandom:
    """libvpx's test RNG (test/acm_random.h:24-85 over gtest's LCG,
    third_party/googletest/src/src/gtest.cc:340): state = (1103515245*state + 12345) mod 2^31."""
    M = 1 << 31

    def __init__(self, seed=0xbaba):
        self.state = seed % self.M

    def generate(self, rng):
        self.state = (1103515245 * self.state + 12345) % self.M
        return self.state % rng

    def rand16(self):
        return (self.generate(self.M) >> 15) & 0xffff

    def rand8(self):
        return (self.generate(self.M) >> 23) & 0xff
