#!/usr/bin/env python3
"""A/B of library builds on tests/test_gpu_multi_decoder.py's first-launch test and its 8-decoder test (tools only):
    VP9HIP_TOOLS_LIB=tools/build/libvp9hip_nullfill.so python tools/first_launch_probe.py [repetitions]
prints per repetition whether a row gave up / a frame differed."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import vp9ref
hip = vp9ref.load_hip()
if os.environ.get("VP9HIP_TOOLS_LIB"):
    hip.LIB_PATH = os.path.join(ROOT, os.environ["VP9HIP_TOOLS_LIB"])
import test_gpu_multi_decoder as t
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for k in range(reps):
    for name, fn in (("first launch beside busy decoders", lambda: t.test_first_launch_of_a_context_on_a_busy_gpu(hip)),
                     ("4 decoders 720p 10-bit", lambda: t.test_decoders_in_one_process(hip, 1280, 720, 10, 4, 60))):
        try:
            fn()
            print(k, name, "ok", flush=True)
        except AssertionError as e:
            print(k, name, "FAILED:", str(e)[:300], flush=True)
