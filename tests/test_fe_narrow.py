"""int16 coefficient slots of the bitstream front-end ("int16 coefficient packing", with the exact fall-back
the reference's 32-bit coefficients ask for): CPU, AddressSanitizer + UBSan, tests/native/fe_narrow.c.  Every golden
stream with the real int16 range, and with a test limit that sends ordinary frames through the fall-back (the tile
columns are parsed a second time into int32 slots; lists, coefficients and the stream state must come out as if the frame
had been parsed once)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STREAMS = os.path.join(ROOT, "tests", "golden", "streams")


@pytest.fixture(scope="module")
def narrow_binary(tmp_path_factory):
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    out = str(tmp_path_factory.mktemp("fe_narrow") / "fe_narrow")
    cmd = ["gcc", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-std=gnu99", "-Wno-missing-braces",
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "cuda-vp9_amd", "csrc", "fe"),
           os.path.join(ROOT, "tests", "native", "fe_narrow.c"), os.path.join(ROOT, "cuda-vp9_amd", "csrc", "fe", "vp9fe.c"),
           os.path.join(ROOT, "cuda-vp9_amd", "csrc", "vp9hip_pack.c"), "-o", out, "-lpthread"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode:
        pytest.skip("sanitizer build not available here: " + r.stdout.decode(errors="replace")[-300:])
    return out


def run(binary, name, threads, limit):
    r = subprocess.run([binary, os.path.join(STREAMS, name + ".ivf"), str(threads), str(limit)], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=600, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1"))
    err = r.stderr.decode(errors="replace")
    assert r.returncode == 0 and "AddressSanitizer" not in err and "runtime error" not in err, err[-1500:]
    m = re.search(rb"fe_narrow: (\d+) frames, (\d+) with int16 slots, (\d+) parsed again", r.stdout)
    assert m, r.stdout
    return tuple(int(v) for v in m.groups())


@pytest.mark.parametrize("name", ["s704_8", "s350_8", "s352_arf", "s704_10", "s352_444", "s352_aq3", "s352_ll", "s352_12", "s352_tr", "s352_svc2"])
def test_int16_slots_hold_the_same_coefficients(narrow_binary, name):
    frames, narrow, again = run(narrow_binary, name, 2, 1)
    assert frames > 0 and narrow + again > 0


@pytest.mark.parametrize("name,limit", [("s704_8", 300), ("s352_arf", 64), ("s704_10", 1000), ("s352_svc2", 200)])
def test_fall_back_to_int32_slots(narrow_binary, name, limit):
    frames, narrow, again = run(narrow_binary, name, 1, limit)
    assert again > 0, "the limit was meant to send some frames through the fall-back"
    frames2, narrow2, again2 = run(narrow_binary, name, 2, limit)
    assert (frames2, narrow2, again2) == (frames, narrow, again)
