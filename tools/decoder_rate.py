#!/usr/bin/env python3
"""Reconstruction rate on a frame with a REAL VP9 block structure (tests/blockgen.py: all 13 block sizes,
sub-8x8, compound, tiles) packed by the C packer, everything resident in HBM: vp9hip_decoder_begin_frame
once, then vp9hip_decoder_run (inter + residual, islands || loop filter) repeatedly."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes
import numpy as np
import __graft_entry__ as g
hip = g.load_pkg()
if os.environ.get("VP9HIP_TOOLS_LIB"):  # A/B of library builds (tools only)
    hip.LIB_PATH = os.environ["VP9HIP_TOOLS_LIB"]
import blockgen
import workload
W, H = 2560, 1440
bd = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rng = np.random.default_rng(3)
dt = np.uint16 if bd > 8 else np.uint8
aw, ah = (W + 7) & ~7, (H + 7) & ~7
dims = [(aw, ah), (aw // 2, ah // 2), (aw // 2, ah // 2)]
for name, kw in (("blockgen default (45 % of the 8x8 blocks are sub-8x8)", dict(intra_frac=0.08, skip_frac=0.35)),
                 ("larger blocks (split 0.6/0.45/0.3/0.2)", dict(intra_frac=0.08, skip_frac=0.35, split_p=(0.6, 0.45, 0.3, 0.2)))):
    blocks = blockgen.gen_blocks(rng, W, H, hip.BLOCK_DTYPE, **kw)
    coef, eob = blockgen.gen_coeffs(rng, blocks, W, H, bd)
    refs = [[np.ascontiguousarray(workload.smooth_noise(rng, d[1], d[0], bd, sigma=1.5 + k).astype(dt)) for d in dims] for k in range(3)]
    P = hip.FrameParams()
    P.width, P.height, P.ss_x, P.ss_y, P.bit_depth, P.hbd, P.log2_tile_cols, P.build_lf_masks = W, H, 1, 1, bd, int(bd > 8), 2, 1
    for k in range(3):
        P.ref_width[k], P.ref_height[k] = W, H
    th = hip.LfThresh()
    hip.lib().vp9hip_lf_frame_init(32, 0, None, None, 0, 0, None, None, None, ctypes.byref(th))
    dec = hip.Decoder(0)
    for k in range(3):
        dec.upload(k, refs[k], W, H, bd)
    dec.alloc_slot(3, W, H, bd)
    t0 = time.perf_counter(); dec.begin_frame(P, blocks, eob, coef); t_begin = time.perf_counter() - t0
    ALL = hip.PHASE_INTER | hip.PHASE_INTRA | hip.PHASE_LF
    for _ in range(5):
        dec.run(ALL, (0, 1, 2), 3, thresh=th)
    dec.sync()
    N = 100
    t0 = time.perf_counter()
    for _ in range(N):
        dec.run(ALL, (0, 1, 2), 3, thresh=th)
    dec.sync()
    dt_ = time.perf_counter() - t0
    inter = int((blocks["ref_frame"][:, 0] > 0).sum())
    print(f"{name}: {len(blocks)} blocks ({inter} inter), begin_frame {t_begin*1e3:.1f} ms; {N/dt_:.0f} frames/s ({dt_/N*1e3:.3f} ms per frame, "
          f"last run {dec.last_run_ms():.3f} ms GPU)")
    dec.close()
