"""Stream-level parity: the reference's own vpxdec (its libvpx compiled where it lies, decode_tiles the
caller) linked against libvp9hip_shim.so decodes VP9 bitstreams on the GPU; every frame's MD5
(vpxdec --md5, libvpx/vpxdec.c:285-302, 490-495 — the format of the reference's Sony.md5 / netflix.md5)
must equal the list the same vpxdec produced with the CPU wrap_cuda_* bodies of oracle/ref_stream_wraps.c
(pinned by `vpxenc --test-decode=fatal`, tests/golden/streams/make_streams.py).

  vpxdec_hip   patched frame driver (oracle/patch_decodeframe.py = INTEGRATION.md mode C): inverse
               transforms, prediction and loop filter on the GPU, references resident in HBM
  vpxdec_hip_mt  the patched driver with the entropy stage run by one thread per tile column
               (patch_decodeframe.py --mt, E10: private list segments / coefficient regions per tile column,
               merged into the canonical lists)
  vpxdec_hipA  UNCHANGED reference frame driver (mode A: its CPU transforms left int64 residual planes,
               its CPU loop filter runs afterwards) — high-bitdepth streams only, like the reference

The binaries are built in the development container (oracle/build_refvpx.sh, needs /root/reference)
and travel to the GPU box as built artefacts; nothing here reads /root/reference."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = os.path.join(ROOT, "tests", "golden", "streams")
BIG = os.path.join(ROOT, "tests", "streams_big")
HIP = os.path.join(ROOT, "shim", "build", "vpxdec_hip")
HIP_A = os.path.join(ROOT, "shim", "build", "vpxdec_hipA")
HIP_MT = os.path.join(ROOT, "shim", "build", "vpxdec_hip_mt")

pytestmark = pytest.mark.gpu


def md5_lines(decoder, ivf, env=None):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([decoder, "--rawvideo", "--md5", "-o", "img-%wx%h-%4.i420", ivf], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, env=e, timeout=900)
    out = r.stdout.decode(errors="replace")
    assert r.returncode == 0, out[-2000:]
    return [l for l in out.splitlines() if re.match(r"^[0-9a-f]{32}  img-", l)]


def golden(path):
    with open(path) as f:
        return [l.rstrip("\n") for l in f if l.strip()]


def check(decoder, directory, name):
    if not os.path.exists(decoder):
        pytest.fail(f"{decoder} missing: run oracle/build_refvpx.sh in the development container")
    want = golden(os.path.join(directory, name + ".md5"))
    got = md5_lines(decoder, os.path.join(directory, name + ".ivf"))
    assert len(got) == len(want), f"{name}: {len(got)} frames decoded, {len(want)} expected"
    bad = [i for i, (a, b) in enumerate(zip(got, want)) if a != b]
    assert not bad, f"{name}: frames {bad[:8]} differ, e.g. {got[bad[0]]} != {want[bad[0]]}"


@pytest.mark.parametrize("name", ["s704_8", "s350_8", "s352_arf", "s704_10", "s352_444", "s352_aq1", "s352_aq3", "s352_er", "s352_fp", "s352_ll", "s352_12",
                                  "s352_tr", "s352_svc2", "s704_svc3", "s352_svc2_10", "s352_444_10", "s16x16", "s6x10"])
def test_stream_md5_patched_driver(name):
    check(HIP, SMALL, name)


@pytest.mark.parametrize("name", ["s704_8", "s350_8", "s352_arf", "s704_10", "s352_444", "s352_aq1", "s352_aq3", "s352_er", "s352_fp", "s352_ll", "s352_12",
                                  "s352_tr", "s352_svc2", "s704_svc3", "s352_svc2_10", "s352_444_10", "s16x16", "s6x10"])
def test_stream_md5_tile_parallel_entropy_stage(name):
    # every stream goes through the tile-column threads (one column: one thread) with compact coefficient slots
    check(HIP_MT, SMALL, name)


def test_tile_parallel_result_does_not_depend_on_the_thread_count():
    want = golden(os.path.join(SMALL, "s704_8.md5"))
    for threads in ("1", "2", "5"):
        assert md5_lines(HIP_MT, os.path.join(SMALL, "s704_8.ivf"), env={"VP9HIP_SHIM_THREADS": threads}) == want


def test_stream_md5_unchanged_reference_driver():
    # profile 2, 704x576: what the reference's decode_tiles can run as it is
    check(HIP_A, SMALL, "s704_10")


@pytest.mark.parametrize("name", ["S-1440", "S-1440-q44", "S-2160", "S-2176", "S-1080-10", "S-1440-10", "S-1080-8", "S-704-resize"])
def test_baseline_sized_stream_md5(name):
    if not os.path.exists(os.path.join(BIG, name + ".ivf")):
        pytest.skip(f"tests/streams_big/{name}.ivf not generated (make_streams.py --big)")
    check(HIP, BIG, name)
    check(HIP_MT, BIG, name)  # 8 / 16 / 16 / 4 tile columns


RTCD = os.path.join(ROOT, "shim", "build", "vpxdec_rtcd")


@pytest.mark.parametrize("name", ["s352_arf", "s350_8"])
def test_stream_md5_rtcd_pointer_dispatch(name):
    """The block-level integration: the reference's run-time dispatch POINTERS assigned to the `_hip` twins
    (shim/vp9hip_rtcd_install.c, called from initialize_dec — oracle/patch_decodeframe.py E13 — where the reference's own
    setup_rtcd_internal assigns SIMD variants, vpx-master/vpx_dsp_rtcd.h:2074).  vpxdec_rtcd is the reference's vpxdec
    with its CPU reconstruction; its directional intra predictors and 16-wide loop filters run on the GPU one block
    at a time (bring-up mode: slow), and the stream's MD5s must still be the golden ones."""
    if not os.path.exists(RTCD):
        pytest.fail(f"{RTCD} missing: run oracle/build_refvpx.sh in the development container")
    want = golden(os.path.join(SMALL, name + ".md5"))
    e = dict(os.environ, VP9HIP_RTCD_TRACE="1")
    r = subprocess.run([RTCD, "--rawvideo", "--md5", "-o", "img-%wx%h-%4.i420", os.path.join(SMALL, name + ".ivf")],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=e, timeout=900)
    out = r.stdout.decode(errors="replace")
    assert r.returncode == 0, out[-2000:]
    got = [l for l in out.splitlines() if re.match(r"^[0-9a-f]{32}  img-", l)]
    assert got == want
    m = re.search(r"vp9hip: (\d+) rtcd pointers were assigned to _hip twins; last twin error: \"(.*)\"", out)
    assert m and int(m.group(1)) >= 15 and m.group(2) == "", out[-600:]
