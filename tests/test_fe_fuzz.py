"""Damaged bitstreams through the front-end and the packer under AddressSanitizer + UBSan (CPU build of
cuda-vp9_amd/csrc/fe/vp9fe.c and csrc/vp9hip_pack.c with tests/native/fe_fuzz.c): bit flips, truncations, damaged
headers, garbage tails.  Nothing may read or write out of bounds whatever the bytes say, and what reaches the GPU
side has passed the packer's checks.  (GPU sanitizers are not available on the pool; the host side is where a
hostile stream arrives.)"""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def fuzz_binary(tmp_path_factory):
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    out = str(tmp_path_factory.mktemp("fe_fuzz") / "fe_fuzz")
    cmd = ["gcc", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-std=gnu99", "-Wno-missing-braces",
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "cuda-vp9_amd", "csrc", "fe"),
           os.path.join(ROOT, "tests", "native", "fe_fuzz.c"), os.path.join(ROOT, "cuda-vp9_amd", "csrc", "fe", "vp9fe.c"),
           os.path.join(ROOT, "cuda-vp9_amd", "csrc", "vp9hip_pack.c"), "-o", out, "-lpthread"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode:
        pytest.skip("sanitizer build not available here: " + r.stdout.decode(errors="replace")[-300:])
    return out


@pytest.mark.parametrize("name,seed", [("s704_8", 11), ("s352_arf", 23), ("s704_10", 5), ("s352_444", 41), ("s350_8", 77)])
def test_damaged_streams_stay_in_bounds(fuzz_binary, name, seed):
    ivf = os.path.join(ROOT, "tests", "golden", "streams", name + ".ivf")
    r = subprocess.run([fuzz_binary, ivf, "60", str(seed)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1"))
    err = r.stderr.decode(errors="replace")
    assert r.returncode == 0 and "AddressSanitizer" not in err and "runtime error" not in err, err[-1500:]
    assert b"frames parsed" in r.stdout
