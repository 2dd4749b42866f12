#!/usr/bin/env python3
"""Run the phases of the bench frame through the product path (C packer + vp9hip_decoder_run) for rocprofv3
kernel-trace / --pmc runs.
    rocprofv3 --pmc FETCH_SIZE -- python3 tools/profile_phase.py --separate --steps 6
--separate: every launch in its own vp9hip_decoder_run, synchronised (rocprofv3 --pmc executes one kernel at a time):
convolve, transforms, the fused island walk + loop filter (walk_lf_kernel: one launch, fine under --pmc), then the two
halves as launches of their own; default: whole frames as in bench.py."""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--width", type=int, default=2560)
ap.add_argument("--height", type=int, default=1440)
ap.add_argument("--bit-depth", type=int, default=8)
ap.add_argument("--separate", action="store_true")
args = ap.parse_args()
import __graft_entry__ as g  # noqa: E402
hip = g.load_pkg()
import bench  # noqa: E402
W, H, bd = args.width, args.height, args.bit_depth
refs, frames = bench.make_frames(hip, W, H, bd, 0, 1)
P = bench.frame_params(hip, W, H, bd)
th = hip.LfThresh()
hip.lib().vp9hip_lf_frame_init(32, 0, None, None, 0, 0, None, None, None, ctypes.byref(th))
dec = hip.Decoder(0)
for k in range(3):
    dec.upload(k, refs[k], W, H, bd)
dec.alloc_slot(3, W, H, bd)
dec.begin_frame(P, frames[0][0], frames[0][2], frames[0][1])
ALL = hip.PHASE_INTER | hip.PHASE_INTRA | hip.PHASE_LF
dec.run(ALL, (0, 1, 2), 3, thresh=th)
dec.sync()
for i in range(args.steps):
    if args.separate:
        for bits in (hip.PHASE_INTER_PRED, hip.PHASE_INTER_RESID, hip.PHASE_INTRA | hip.PHASE_LF, hip.PHASE_INTRA, hip.PHASE_LF):
            dec.run(bits, (0, 1, 2), 3, thresh=th)
            dec.sync()
    else:
        dec.run(ALL, (0, 1, 2), 3, thresh=th)
dec.sync()
print(f"{args.steps} frames, last run {dec.last_run_ms() * 1e3:.1f} us")
dec.close()
