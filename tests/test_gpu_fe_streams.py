"""The stand-alone decoder built only from this repository (cuda-vp9_amd/vp9hip_dec: IVF -> vp9hip_fe on the CPU ->
vp9hip_decoder on the GPU; no libvpx on either side) against the per-frame MD5 lists of the reference's CPU path
(tests/golden/streams/*.md5, vpxdec --md5 format): every shown frame of every stream, pipelined and serial, one
and several entropy threads."""
import os
import re
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEC = os.path.join(ROOT, "cuda-vp9_amd", "vp9hip_dec")
SMALL = os.path.join(ROOT, "tests", "golden", "streams")
BIG = os.path.join(ROOT, "tests", "streams_big")


def md5_lines(ivf, *opts):
    r = subprocess.run([DEC, "--md5", "-o", "img-%wx%h-%4.i420", *opts, ivf], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode(errors="replace")[-800:]
    return [l for l in r.stdout.decode().splitlines() if re.match(r"^[0-9a-f]{32}  img-", l)]


def golden(path):
    return [l.rstrip("\n") for l in open(path) if l.strip()]


def check(d, name, *opts):
    want = golden(os.path.join(d, name + ".md5"))
    got = md5_lines(os.path.join(d, name + ".ivf"), *opts)
    assert len(got) == len(want), f"{name}: {len(got)} frames, golden {len(want)}"
    bad = [i for i, (a, b) in enumerate(zip(got, want)) if a != b]
    assert not bad, f"{name}: frames {bad[:8]} differ, first: {got[bad[0]]} != {want[bad[0]]}"


@pytest.mark.parametrize("name", ["s704_8", "s350_8", "s352_arf", "s704_10", "s352_444", "s352_aq1", "s352_aq3", "s352_er", "s352_fp", "s352_ll",
                                  "s352_12", "s352_tr", "s352_svc2", "s704_svc3", "s352_svc2_10", "s352_444_10", "s16x16", "s6x10"])
def test_standalone_decoder_md5(name):
    assert os.path.exists(DEC), "cuda-vp9_amd/vp9hip_dec not built (make -C cuda-vp9_amd)"
    check(SMALL, name)  # (int16 coefficient slots wherever a frame's coefficients fit: the default)
    check(SMALL, name, "--serial", "--threads=1")
    check(SMALL, name, "--wide-slots")  # int32 slots throughout, the reference's width


@pytest.mark.parametrize("name", ["S-1440", "S-1440-q44", "S-2160", "S-2176", "S-1080-10", "S-1440-10", "S-1080-8", "S-704-resize"])
def test_standalone_decoder_md5_baseline_sized(name):
    if not os.path.exists(os.path.join(BIG, name + ".ivf")):
        pytest.skip("tests/streams_big not generated (make_streams.py --big)")
    check(BIG, name)


def test_standalone_decoder_writes_the_frames_it_hashes(tmp_path):
    """-o without --md5: one raw file per shown frame (vpxdec's multi-file mode); the files' MD5s are the golden lines."""
    import hashlib
    ivf = os.path.join(SMALL, "s352_arf.ivf")  # hidden frames and a show-existing frame: the numbering follows shown frames
    pattern = str(tmp_path / "img-%wx%h-%4.i420")
    r = subprocess.run([DEC, "-o", pattern, ivf], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr.decode(errors="replace")[-500:]
    want = golden(os.path.join(SMALL, "s352_arf.md5"))
    for line in want:
        digest, name = line.split()
        data = open(tmp_path / name, "rb").read()
        assert len(data) == 352 * 288 * 3 // 2
        assert hashlib.md5(data).hexdigest() == digest, name
    assert len(list(tmp_path.iterdir())) == len(want)


def test_standalone_decoder_refuses_a_damaged_stream(tmp_path):
    """A stream cut in the middle of a frame ends with an error message and a non-zero exit code, not with a fault."""
    data = open(os.path.join(SMALL, "s704_8.ivf"), "rb").read()
    # keep the IVF framing intact but overwrite the second half of the third packet's payload
    pos, k = 32, 0
    while k < 2:
        pos += 12 + int.from_bytes(data[pos:pos + 4], "little")
        k += 1
    n = int.from_bytes(data[pos:pos + 4], "little")
    bad = bytearray(data)
    for i in range(pos + 12 + n // 2, pos + 12 + n):
        bad[i] = (i * 37) & 0xff
    p = tmp_path / "bad.ivf"
    p.write_bytes(bytes(bad))
    r = subprocess.run([DEC, "--md5", "-o", "img-%wx%h-%4.i420", str(p)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    lines = [l for l in r.stdout.decode().splitlines() if re.match(r"^[0-9a-f]{32}  img-", l)]
    want = golden(os.path.join(SMALL, "s704_8.md5"))
    assert lines[:2] == want[:2]  # the frames before the damage are right
    # the damaged frame either fails to parse (error exit) or decodes to something else; it never takes the process down
    assert r.returncode in (0, 1) and (r.returncode == 1 or lines[2:3] != want[2:3])


@pytest.mark.parametrize("order", ["0", "1"])
def test_both_grid_orders_of_the_fused_walk_and_filter_launch(order):
    """VP9HIP_ISLANDS_FIRST: the filter's rows first (one context in the process, the default there) or the islands first
    (what a process with several contexts gets, so that a row never waits for a workgroup dispatched after it)."""
    ivf = os.path.join(SMALL, "s704_8.ivf")
    r = subprocess.run([DEC, "--md5", "-o", "img-%wx%h-%4.i420", ivf], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300,
                       env=dict(os.environ, VP9HIP_ISLANDS_FIRST=order))
    assert r.returncode == 0, r.stderr.decode(errors="replace")[-500:]
    got = [l for l in r.stdout.decode().splitlines() if re.match(r"^[0-9a-f]{32}  img-", l)]
    assert got == golden(os.path.join(SMALL, "s704_8.md5"))
