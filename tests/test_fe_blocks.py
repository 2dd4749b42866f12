"""The product's own VP9 bitstream front-end (include/vp9hip_fe.h, CPU only) against the reference's parse of the
same streams: every block's mode information — size, transform size, skip flag, segment, references, prediction
modes, motion vectors, interpolation filter, loop-filter level — and a checksum over every block's eobs and
dequantised coefficients, block by block in decode order (tests/fe_compare.py).  The reference side is
oracle/_ref/vpx/vpxdec_c, the reference's vpxdec compiled from its own sources (test infrastructure)."""
import os

import pytest

import fe_compare

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = os.path.join(ROOT, "tests", "golden", "streams")
BIG = os.path.join(ROOT, "tests", "streams_big")
REFDEC = os.path.join(ROOT, "oracle", "_ref", "vpx", "vpxdec_c")

pytestmark = pytest.mark.skipif(not os.path.exists(REFDEC), reason="oracle/_ref/vpx/vpxdec_c not built (oracle/build_refvpx.sh)")


@pytest.mark.parametrize("name", ["s704_8", "s350_8", "s352_arf", "s704_10", "s352_444", "s352_aq1", "s352_aq3", "s352_er", "s352_fp", "s352_ll",
                                  "s352_12", "s352_tr", "s352_svc2", "s704_svc3", "s352_svc2_10", "s352_444_10", "s16x16", "s6x10"])
def test_front_end_parses_what_the_reference_parses(hip, name):
    path = os.path.join(SMALL, name + ".ivf")
    mine = fe_compare.parse_stream(hip, path)
    assert fe_compare.compare(mine, fe_compare.reference_blocks(path)) is None


def test_front_end_same_lists_for_every_thread_count(hip):
    path = os.path.join(SMALL, "s704_8.ivf")  # two tile columns
    one = fe_compare.parse_stream(hip, path, threads=1)
    two = fe_compare.parse_stream(hip, path, threads=2)
    assert len(one) == len(two) and all((a == b).all() for a, b in zip(one, two))


@pytest.mark.parametrize("name", ["S-1440", "S-2176", "S-1080-10", "S-1440-10", "S-1080-8", "S-704-resize"])
def test_front_end_on_baseline_sized_streams(hip, name):
    path = os.path.join(BIG, name + ".ivf")
    if not os.path.exists(path):
        pytest.skip("tests/streams_big not generated (make_streams.py --big)")
    mine = fe_compare.parse_stream(hip, path, threads=0)  # one thread per tile column
    assert fe_compare.compare(mine, fe_compare.reference_blocks(path)) is None


def test_front_end_rejects_garbage(hip):
    import ctypes
    lib = hip.lib()
    fe = ctypes.c_void_p()
    lib.vp9hip_fe_create.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    lib.vp9hip_fe_parse.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p]
    lib.vp9hip_fe_destroy.argtypes = [ctypes.c_void_p]
    assert lib.vp9hip_fe_create(ctypes.byref(fe), None, None, None, 1) == 0
    out = ctypes.create_string_buffer(4096)
    for junk in (b"\x00" * 40, b"\xff" * 64, bytes(range(200)), b"\x82\x49\x83\x42\x00\x00"):
        assert lib.vp9hip_fe_parse(fe, junk, len(junk), out) != 0
    # a truncated key frame of a real stream
    pkt = next(fe_compare.ivf_frames(os.path.join(SMALL, "s704_8.ivf")))
    for cut in (3, 10, 40, len(pkt) // 2):
        lib.vp9hip_fe_parse(fe, pkt[:cut], cut, out)  # must not crash; may succeed on a long prefix only by chance
    lib.vp9hip_fe_destroy(fe)
