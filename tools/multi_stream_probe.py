#!/usr/bin/env python3
"""Which phase limits several decoders in flight on one GPU: frames/s of N decoders (one host thread each)
running only the given phases of the bench frame."""
import ctypes, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as g
hip = g.load_pkg()
import bench
W, H, bd = 2560, 1440, 8
refs, frames = bench.make_frames(hip, W, H, bd, 0, 1)
P = bench.frame_params(hip, W, H, bd)
th = hip.LfThresh()
hip.lib().vp9hip_lf_frame_init(32, 0, None, None, 0, 0, None, None, None, ctypes.byref(th))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
decs = []
for s in range(N):
    d = hip.Decoder(0)
    for k in range(3):
        d.upload(k, refs[k], W, H, bd)
    d.alloc_slot(3, W, H, bd)
    d.begin_frame(P, frames[0][0], frames[0][2], frames[0][1])
    decs.append(d)
for name, bits in (("inter_pred", hip.PHASE_INTER_PRED), ("inter_resid", hip.PHASE_INTER_RESID), ("inter", hip.PHASE_INTER),
                   ("intra", hip.PHASE_INTRA), ("lf", hip.PHASE_LF), ("intra+lf", hip.PHASE_INTRA | hip.PHASE_LF),
                   ("all", hip.PHASE_INTER | hip.PHASE_INTRA | hip.PHASE_LF)):
    for n_act in sorted({1, 2, N}):
        R = 40
        def drive(d):
            for _ in range(R):
                d.run(bits, (0, 1, 2), 3, thresh=th)
            d.sync()
        for d in decs[:n_act]:
            d.run(bits, (0, 1, 2), 3, thresh=th); d.sync()
        thr = [threading.Thread(target=drive, args=(d,)) for d in decs[:n_act]]
        t0 = time.perf_counter()
        for x in thr: x.start()
        for x in thr: x.join()
        dt = time.perf_counter() - t0
        print(f"{name:12s} streams {n_act:2d}: {R * n_act / dt:8.0f} frames/s  ({dt / R * 1e6:7.1f} us per round)")
