#!/usr/bin/env python3
"""HIP-event time of one phase of the bench frame, for A/B of library builds (median of 40 launches):
    VP9HIP_TOOLS_LIB=tools/build/libX.so python tools/phase_ab.py [inter_pred|inter_resid|intra_lf] [bit depth]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as g
hip = g.load_pkg()
if os.environ.get("VP9HIP_TOOLS_LIB"):
    hip.LIB_PATH = os.path.join(ROOT, os.environ["VP9HIP_TOOLS_LIB"])
import bench
which = sys.argv[1] if len(sys.argv) > 1 else "inter_resid"
bd = int(sys.argv[2]) if len(sys.argv) > 2 else 8
bits = {"inter_pred": hip.PHASE_INTER_PRED, "inter_resid": hip.PHASE_INTER_RESID, "intra_lf": hip.PHASE_INTRA | hip.PHASE_LF,
        "all": hip.PHASE_INTER | hip.PHASE_INTRA | hip.PHASE_LF}[which]
W, H = 2560, 1440
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
dec.run(ALL, (0, 1, 2), 3, thresh=th); dec.sync()
ts = []
for i in range(40):
    dec.run(hip.PHASE_INTER_PRED, (0, 1, 2), 3, thresh=th)  # (the prediction the residual is added to)
    dec.sync()
    dec.run(bits, (0, 1, 2), 3, thresh=th)
    dec.sync()
    ts.append(dec.last_run_ms() * 1e3)
ts.sort()
print(f"{which} {bd}-bit: median {ts[len(ts) // 2]:.1f} us, min {ts[0]:.1f} us")
