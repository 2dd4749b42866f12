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
                                  "s352_12", "s352_tr"])
def test_standalone_decoder_md5(name):
    assert os.path.exists(DEC), "cuda-vp9_amd/vp9hip_dec not built (make -C cuda-vp9_amd)"
    check(SMALL, name)
    check(SMALL, name, "--serial", "--threads=1")


@pytest.mark.parametrize("name", ["S-1440", "S-2160", "S-2176", "S-1080-10"])
def test_standalone_decoder_md5_baseline_sized(name):
    if not os.path.exists(os.path.join(BIG, name + ".ivf")):
        pytest.skip("tests/streams_big not generated (make_streams.py --big)")
    check(BIG, name)
