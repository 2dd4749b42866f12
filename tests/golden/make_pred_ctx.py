#!/usr/bin/env python3
"""Generates tests/golden/pred_ctx.npz: the reference's neighbour-context functions (vp9_pred_common.h / .c) over
every combination of absent / intra / single / compound neighbours under every sign-bias pattern, run through
the REFERENCE's own object code (oracle/dump_pred_contexts.c linked against oracle/_ref/vpx/libvpxfull.a).
Build container only; the table is committed.     python tests/golden/make_pred_ctx.py"""
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("REF", "/root/reference")
VPX = os.path.join(ROOT, "oracle", "_ref", "vpx")

import triton  # noqa: E402  (only for NVIDIA's cuda_runtime.h, which the reference's headers include)
cuda_inc = os.path.join(os.path.dirname(triton.__file__), "backends", "nvidia", "include")
exe = os.path.join(VPX, "tools", "dump_pred_contexts")
subprocess.check_call(["gcc", "-std=gnu99", "-O1", "-w", f"-I{REF}/vpx-master", f"-I{REF}/libvpx", f"-I{cuda_inc}", "-include",
                       os.path.join(VPX, "simd_to_c.h"), os.path.join(ROOT, "oracle", "dump_pred_contexts.c"),
                       os.path.join(VPX, "libvpxfull.a"), "-lm", "-lpthread", "-o", exe])
raw = np.frombuffer(subprocess.check_output([exe]), "<i4").reshape(-1, 24)
np.savez_compressed(os.path.join(HERE, "pred_ctx.npz"), above=raw[:, 0:6], left=raw[:, 6:12], sign_bias=raw[:, 12:15], max_tx=raw[:, 15],
                    ctx=raw[:, 16:24])
print("wrote pred_ctx.npz:", raw.shape[0], "cases")
