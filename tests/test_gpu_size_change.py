"""A frame size (and bit depth) change in mid-stream through all three GPU decoders: two golden streams back to back in
one IVF file (tests/ivf_tools.concat_ivf — each starts with a key frame, so the expected MD5 list is the two golden
lists one after the other; the reference's own CPU decoder oracle/_ref/vpx/vpxdec_c gives exactly that for these files,
checked when the test was written).  What it reaches that no single stream does: the front-end's arrays growing while
earlier frames are still in flight (vp9hip_dec's pipelined mode), the device frame pool and the work-list ring
re-sized under a running decoder, the loop filter's island counters for another geometry, libvpx's own buffer
re-allocation around the shim's resident references."""
import os
import re
import subprocess

import pytest

import ivf_tools

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = os.path.join(ROOT, "tests", "golden", "streams")
DECODERS = {
    "vp9hip_dec": [os.path.join(ROOT, "cuda-vp9_amd", "vp9hip_dec"), "--md5", "-o", "img-%wx%h-%4.i420"],
    "vp9hip_dec_serial": [os.path.join(ROOT, "cuda-vp9_amd", "vp9hip_dec"), "--md5", "-o", "img-%wx%h-%4.i420", "--serial", "--threads=1"],
    "vpxdec_hip": [os.path.join(ROOT, "shim", "build", "vpxdec_hip"), "--rawvideo", "--md5", "-o", "img-%wx%h-%4.i420"],
    "vpxdec_hip_mt": [os.path.join(ROOT, "shim", "build", "vpxdec_hip_mt"), "--rawvideo", "--md5", "-o", "img-%wx%h-%4.i420"],
}


def hashes(path):
    return [l.split()[0] for l in open(path) if l.strip()]


@pytest.mark.parametrize("decoder", sorted(DECODERS))
@pytest.mark.parametrize("parts", [("s352_arf", "s704_8"), ("s704_8", "s352_arf"), ("s704_10", "s352_12", "s704_10"),
                                    ("s350_8", "s704_8", "s352_er", "s350_8")])
def test_size_change_in_mid_stream(tmp_path, decoder, parts):
    cmd = DECODERS[decoder]
    assert os.path.exists(cmd[0]), f"{cmd[0]} not built"
    ivf = str(tmp_path / "cat.ivf")
    ivf_tools.concat_ivf([os.path.join(SMALL, p + ".ivf") for p in parts], ivf)
    want = [h for p in parts for h in hashes(os.path.join(SMALL, p + ".md5"))]
    r = subprocess.run(cmd + [ivf], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert r.returncode == 0, r.stdout.decode(errors="replace")[-1200:]
    got = [l.split()[0] for l in r.stdout.decode().splitlines() if re.match(r"^[0-9a-f]{32}  img-", l)]
    assert len(got) == len(want), f"{len(got)} frames, expected {len(want)}"
    bad = [i for i, (a, b) in enumerate(zip(got, want)) if a != b]
    assert not bad, f"frames {bad[:8]} differ"
