"""smoke: one small frame (352x288, 8-bit 4:2:0) through every HIP kernel family on cuda:0,
compared bit-for-bit with the sequential CPU oracle."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def run(verbose=False):
    import __graft_entry__ as g
    import frame_check
    pkg = g.load_pkg()
    import cuda_vp9_amd.pipeline as pipeline
    import workload
    wl = workload.make_frame_workload(352, 288, seed=7, bd=8, intra_frac=0.2)
    ctx = pkg.Context(0)
    job = pipeline.FrameJob(ctx, wl)
    job.clear_dst()
    job.run()
    ctx.sync()
    got = job.download()
    exp, _ = frame_check.oracle_frame(frame_check.load_oracle(), wl)
    for p in range(3):
        if not np.array_equal(got[p], exp[p]):
            raise AssertionError(f"smoke: plane {p} differs from the oracle in {(got[p] != exp[p]).sum()} samples")
    md5 = frame_check.frame_md5(got, wl)
    if verbose:
        print(f"smoke ok: 352x288 frame, {wl['n_blocks']} blocks, {len(wl['inter_tasks'])} inter tasks, "
              f"{len(wl['txb'])} tx blocks, {len(wl['intra_sorted'])} intra blocks in {wl['n_waves']} waves; "
              f"md5 {md5} == oracle")
    job.free()
    ctx.close()
    # and one small real bitstream through the reference's vpxdec linked against the shim (when the built
    # decoder travelled with the tree): every frame's MD5 against the CPU path's list
    import re
    import subprocess
    dec = os.path.join(ROOT, "shim", "build", "vpxdec_hip_mt")
    ivf = os.path.join(ROOT, "tests", "golden", "streams", "s704_8.ivf")
    if os.path.exists(dec):
        out = subprocess.run([dec, "--rawvideo", "--md5", "-o", "img-%wx%h-%4.i420", ivf], stdout=subprocess.PIPE,
                             stderr=subprocess.STDOUT, timeout=300)
        got_md5 = [l for l in out.stdout.decode(errors="replace").splitlines() if re.match(r"^[0-9a-f]{32}  img-", l)]
        want = [l.rstrip("\n") for l in open(ivf[:-4] + ".md5") if l.strip()]
        if out.returncode or got_md5 != want:
            raise AssertionError("smoke: vpxdec_hip_mt on s704_8.ivf: per-frame MD5s differ from the CPU path's")
        if verbose:
            print(f"smoke ok: s704_8.ivf through vpxdec -> decode_tiles -> wrap_cuda_* -> HIP, {len(want)} frames MD5-equal")
    # and the same stream through the decoder built only from this repository (own front-end + GPU reconstruction)
    own = os.path.join(ROOT, "cuda-vp9_amd", "vp9hip_dec")
    if os.path.exists(own):
        out = subprocess.run([own, "--md5", "-o", "img-%wx%h-%4.i420", ivf], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        got_md5 = [l for l in out.stdout.decode(errors="replace").splitlines() if re.match(r"^[0-9a-f]{32}  img-", l)]
        want = [l.rstrip("\n") for l in open(ivf[:-4] + ".md5") if l.strip()]
        if out.returncode or got_md5 != want:
            raise AssertionError("smoke: vp9hip_dec on s704_8.ivf: per-frame MD5s differ from the reference CPU path's")
        if verbose:
            print(f"smoke ok: s704_8.ivf through vp9hip_dec (vp9hip_fe -> vp9hip_decoder), {len(want)} frames MD5-equal")


if __name__ == "__main__":
    run(verbose=True)
