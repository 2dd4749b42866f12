#!/usr/bin/env python3
"""tests/golden/streams/make_streams.py — synthesizes the VP9 test streams and their golden MD5 lists.

Runs in the development container only (needs oracle/_ref/vpx/{vpxenc_c,vpxdec_c,vpxdec_cA}, i.e.
/root/reference + oracle/build_refvpx.sh).  For every stream:
  1. a seeded synthetic source (band-limited noise translating by (dx, dy) samples per frame, fresh
     noise patches in some frames to force intra blocks) is written to a scratch file;
  2. the reference's encoder, linked with the CPU stream oracle as its decoder, encodes it with
     `--test-decode=fatal`: every frame the oracle decodes is compared with the encoder's own
     reconstruction, so a stream only gets here if the oracle decodes it exactly;
  3. the reference's vpxdec (CPU wrap_cuda_* bodies) writes the per-frame MD5 list in vpxdec's
     `--md5` format (libvpx/vpxdec.c:285-302, 490-495) — the format of Sony.md5 / netflix.md5.
Small streams (<= ~200 KB) land in tests/golden/streams/ and are committed; the BASELINE.json-sized
ones (S-1440, S-2160, S-2176, S-1080-10, S-1440-10, S-1080-8) land in tests/streams_big/ (git-ignored, travels to the GPU box).

    python3 tests/golden/streams/make_streams.py [--big] [name ...]
"""
import hashlib
import os
import re
import subprocess
import sys
import tempfile

import numpy as np
from scipy.ndimage import gaussian_filter

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", "..", ".."))
VPX = os.path.join(ROOT, "oracle", "_ref", "vpx")
BIG = os.path.join(ROOT, "tests", "streams_big")

COMMON = ["--codec=vp9", "--ivf", "--end-usage=q", "--kf-max-dist=9999", "--test-decode=fatal", "--quiet"]

# name: (width, height, frames, seed, dx, dy, bits, chroma, patch_prob, jitter, encoder args, big)
STREAMS = {
    # 8-bit 4:2:0, two tile columns, one pass, no lag
    "s704_8": (704, 576, 8, 704, 5, 3, 8, "420", 0.4, 0, ["--cpu-used=2", "--cq-level=30", "--tile-columns=1", "--lag-in-frames=0", "--passes=1"], False),
    # odd size (not a multiple of 8), tile rows
    "s350_8": (350, 286, 8, 350, -3, 2, 8, "420", 0.4, 0, ["--cpu-used=1", "--cq-level=28", "--tile-rows=1", "--lag-in-frames=0", "--passes=1"], False),
    # two-pass with alt-ref: compound prediction, hidden frames, show_existing_frame
    "s352_arf": (352, 288, 16, 352, 4, -2, 8, "420", 0.2, 0, ["--good", "--cpu-used=1", "--cq-level=32", "--passes=2", "--auto-alt-ref=1", "--lag-in-frames=12"], False),
    # 10-bit profile 2, width and height multiples of 64 (the UNCHANGED reference driver reads
    # size_for_mb out of bounds otherwise, vp9_decodeframe.c:2489-2534)
    "s704_10": (704, 576, 6, 710, 6, 2, 10, "420", 0.4, 0, ["--profile=2", "--bit-depth=10", "--input-bit-depth=10", "--cpu-used=2", "--cq-level=30", "--tile-columns=1", "--lag-in-frames=0", "--passes=1"], False),
    # 4:4:4 (profile 1): chroma planes as large as luma, loop-filtered with the luma masks (LF_PATH_444)
    "s352_444": (352, 288, 6, 444, 3, 2, 8, "444", 0.4, 0, ["--profile=1", "--cpu-used=2", "--cq-level=30", "--lag-in-frames=0", "--passes=1"], False),
    # frames smaller than a block: 16x16, and 6x10 (chroma planes 3 samples wide: narrower than the dword the register
    # convolve replicates edges from — every task goes through the generic kernel)
    "s16x16": (16, 16, 5, 1616, 1, 1, 8, "420", 0.0, 0, ["--cpu-used=2", "--cq-level=20", "--lag-in-frames=0", "--passes=1"], False),
    "s6x10": (6, 10, 5, 610, 1, 0, 8, "420", 0.0, 0, ["--cpu-used=2", "--cq-level=20", "--lag-in-frames=0", "--passes=1"], False),
    # profile 3: 4:4:4 with 16-bit samples (the chroma planes on the luma path of the loop filter, full-size chroma convolve)
    "s352_444_10": (352, 288, 4, 4410, -2, 3, 10, "444", 0.4, 0, ["--profile=3", "--bit-depth=10", "--input-bit-depth=10", "--cpu-used=2", "--cq-level=28", "--lag-in-frames=0", "--passes=1"], False),
    # syntax the streams above do not reach (all 352x288, a few frames):
    # segmentation: variance AQ (per-segment quantiser, segment map coded in every frame)
    "s352_aq1": (352, 288, 6, 3521, 3, 2, 8, "420", 0.4, 0, ["--cpu-used=2", "--cq-level=30", "--aq-mode=1", "--lag-in-frames=0", "--passes=1"], False),
    # segmentation with temporal prediction of the map: cyclic refresh (real-time, CBR)
    "s352_aq3": (352, 288, 8, 3523, 2, 1, 8, "420", 0.3, 0, ["--rt", "--cpu-used=5", "--end-usage=cbr", "--target-bitrate=300", "--aq-mode=3", "--lag-in-frames=0", "--passes=1", "--error-resilient=0"], False),
    # error-resilient: every frame resets its contexts, no backward adaptation, no previous-frame vectors
    "s352_er": (352, 288, 6, 3524, 4, 2, 8, "420", 0.4, 0, ["--cpu-used=2", "--cq-level=30", "--error-resilient=1", "--lag-in-frames=0", "--passes=1"], False),
    # frame-parallel mode: forward probability updates only
    "s352_fp": (352, 288, 6, 3525, 4, 2, 8, "420", 0.4, 0, ["--cpu-used=2", "--cq-level=30", "--frame-parallel=1", "--lag-in-frames=0", "--passes=1"], False),
    # lossless: Walsh-Hadamard 4x4 only, base_qindex 0
    "s352_ll": (352, 288, 3, 3526, 2, 1, 8, "420", 0.0, 0, ["--cpu-used=2", "--lossless=1", "--lag-in-frames=0", "--passes=1"], False),
    # 12-bit profile 2: 18-bit coefficient category
    "s352_12": (320, 256, 4, 3527, 3, 2, 12, "420", 0.4, 0, ["--profile=2", "--bit-depth=12", "--input-bit-depth=12", "--cpu-used=2", "--cq-level=24", "--lag-in-frames=0", "--passes=1"], False),
    # four tile rows and two tile columns
    "s352_tr": (352, 288, 5, 3528, 3, 2, 8, "420", 0.4, 0, ["--cpu-used=2", "--cq-level=30", "--tile-columns=1", "--tile-rows=2", "--lag-in-frames=0", "--passes=1"], False),
    # BASELINE.json-sized streams (SURVEY §8d / BASELINE.md §2)
    "S-1440": (2560, 1440, 60, 1440, 5, 3, 8, "420", 0.1, 0, ["--cpu-used=2", "--cq-level=24", "--tile-columns=3", "--lag-in-frames=0", "--passes=1"], True),
    "S-2160": (3840, 2160, 30, 2160, 23, -17, 8, "420", 0.0, 8, ["--cpu-used=4", "--cq-level=32", "--tile-columns=4", "--lag-in-frames=0", "--passes=1"], True),
    # BASELINE.json config 1's exact geometry (FoodMarket2: 3840x2176 = 60x34 whole superblocks), two-pass with alt-ref
    "S-2176": (3840, 2176, 12, 2176, 9, -5, 8, "420", 0.1, 0, ["--good", "--cpu-used=4", "--cq-level=30", "--tile-columns=4", "--passes=2", "--auto-alt-ref=1", "--lag-in-frames=8"], True),
    # the same motion as S-1440 at a lower rate (cq 44: a fifth of the coefficients) — what a stream of ordinary
    # quality asks of the entropy stage
    "S-1440-q44": (2560, 1440, 60, 1440, 5, 3, 8, "420", 0.1, 0, ["--cpu-used=2", "--cq-level=44", "--tile-columns=3", "--lag-in-frames=0", "--passes=1"], True),
    "S-1080-10": (1920, 1080, 30, 1080, 7, 4, 10, "420", 0.1, 0, ["--profile=2", "--bit-depth=10", "--input-bit-depth=10", "--cpu-used=2", "--cq-level=28", "--tile-columns=2", "--lag-in-frames=0", "--passes=1"], True),
    # the headline geometry at 10 bits (Bravia.1440.ivf is almost certainly profile 2 — the reference's kernels only
    # handle uint16 frames) and the north star's other size at 8 bits
    "S-1440-10": (2560, 1440, 30, 14410, 5, 3, 10, "420", 0.1, 0, ["--profile=2", "--bit-depth=10", "--input-bit-depth=10", "--cpu-used=3", "--cq-level=26", "--tile-columns=3", "--lag-in-frames=0", "--passes=1"], True),
    "S-1080-8": (1920, 1080, 60, 10808, 6, -3, 8, "420", 0.1, 0, ["--cpu-used=2", "--cq-level=26", "--tile-columns=2", "--lag-in-frames=0", "--passes=1"], True),
    # the encoder's dynamic resize (one-pass CBR far under the content's rate): NON-key frames of another size in
    # mid-stream — 704x576, from frame 60 on 528x432, from frame 90 on 352x288 — predicted from the larger frames
    # through scale factors 4/3, 2 and 3/2 (vp9_decodeframe.c:1781 setup_frame_size_with_refs, :3232 scale factors; the
    # 64-phase scaled convolve with steps other than 32, vpx_convolve.c:22-535).  (The encoder does not resize frames
    # under 426x240, vp9_ratectrl.c, hence the size.)
    "S-704-resize": (704, 576, 120, 777, 5, 3, 8, "420", 0.9, 8,
                     ["--rt", "--cpu-used=7", "--end-usage=cbr", "--target-bitrate=80", "--resize-allowed=1", "--lag-in-frames=0", "--passes=1",
                      "--min-q=2", "--max-q=52", "--buf-sz=1000", "--buf-initial-sz=500", "--buf-optimal-sz=600", "--undershoot-pct=50",
                      "--overshoot-pct=50", "--drop-frame=0"], True),
}


# Streams the command-line encoder cannot make (oracle/ref_svc_encode.c drives the reference's encoder API): spatial
# layers — every superframe holds a half-size (quarter-size) frame and the frames predicted from it through scale
# factors, so a frame's references have ANOTHER size (vp9_decodeframe.c:1781, 3232-3237) — and an intra-only frame in
# mid-stream (:3182-3213).  name: (width, height, frames, seed, dx, dy, patch_prob, layers, intra_only_at, kbps, speed[, bits])
# s352_svc2_10: the same in profile 2 — scaled prediction with 16-bit samples at stream level
SVC_STREAMS = {
    "s352_svc2": (352, 288, 8, 35202, 3, 2, 0.3, 2, 4, 700, 6),
    "s704_svc3": (704, 576, 6, 70403, 4, -2, 0.3, 3, 3, 1800, 7),
    "s352_svc2_10": (352, 288, 6, 35210, -3, 2, 0.3, 2, 3, 900, 6, 10),
}


def source(path, w, h, n, seed, dx, dy, bits, chroma, patch_prob, jitter):
    rng = np.random.default_rng(seed)
    ss = 1 if chroma == "420" else 0
    pad = 64 + jitter
    W, H = w + abs(dx) * n + 2 * pad, h + abs(dy) * n + 2 * pad
    planes = []
    for _ in range(3):
        p = gaussian_filter(rng.uniform(0.0, 1.0, (H, W)).astype(np.float32), 2.0)
        planes.append((p - p.min()) / (p.max() - p.min()))
    mx = (1 << bits) - 1
    dt = np.uint8 if bits == 8 else "<u2"
    with open(path, "wb") as f:
        for i in range(n):
            ox = pad + (dx * i if dx >= 0 else -dx * (n - 1 - i))
            oy = pad + (dy * i if dy >= 0 else -dy * (n - 1 - i))
            fr = [planes[0][oy:oy + h, ox:ox + w].copy(),
                  planes[1][oy:oy + h, ox:ox + w][::1 + ss, ::1 + ss].copy(),
                  planes[2][oy:oy + h, ox:ox + w][::1 + ss, ::1 + ss].copy()]
            if jitter and i > 0:  # per-64x64-tile random displacement: incoherent sub-pel motion
                for ty in range(0, h, 64):
                    for tx in range(0, w, 64):
                        jx, jy = rng.integers(-jitter, jitter + 1, 2)
                        fr[0][ty:ty + 64, tx:tx + 64] = planes[0][oy + jy + ty:oy + jy + ty + 64, ox + jx + tx:ox + jx + tx + 64][:h - ty, :w - tx]
            if i > 0 and rng.uniform() < patch_prob:
                ps = min(256, h // 3)
                py, px = int(rng.integers(0, h - ps)), int(rng.integers(0, w - ps))
                fr[0][py:py + ps, px:px + ps] = rng.uniform(0.0, 1.0, (ps, ps))
            for p in fr:
                f.write(np.clip(np.round(p * mx), 0, mx).astype(dt).tobytes())


def run(cmd, **kw):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, **kw)
    if r.returncode:
        sys.exit(f"FAILED ({r.returncode}): {' '.join(cmd)}\n{r.stdout.decode(errors='replace')[-2000:]}")
    return r.stdout.decode(errors="replace")


def md5_list(decoder, ivf):
    out = run([decoder, "--i420" if False else "--rawvideo", "--md5", "-o", "img-%wx%h-%4.i420", ivf])
    return [l for l in out.splitlines() if re.match(r"^[0-9a-f]{32}  img-", l)]


def make(name):
    w, h, n, seed, dx, dy, bits, chroma, patch, jitter, enc, big = STREAMS[name]
    outdir = BIG if big else HERE
    os.makedirs(outdir, exist_ok=True)
    ivf = os.path.join(outdir, name + ".ivf")
    with tempfile.TemporaryDirectory() as tmp:
        yuv = os.path.join(tmp, "src.yuv")
        source(yuv, w, h, n, seed, dx, dy, bits, chroma, patch, jitter)
        fmt = ["--i420"] if chroma == "420" else ["--i444"]
        run([os.path.join(VPX, "vpxenc_c")] + COMMON + fmt + enc + ["-w", str(w), "-h", str(h), f"--limit={n}",
             f"--fpf={tmp}/fpf", "-o", ivf, yuv])
    lines = md5_list(os.path.join(VPX, "vpxdec_c"), ivf)
    if len(lines) < n - 1:
        sys.exit(f"{name}: only {len(lines)} frames decoded")
    if bits > 8 and w % 64 == 0 and h % 64 == 0:  # the unchanged reference driver agrees with the patched one
        if md5_list(os.path.join(VPX, "vpxdec_cA"), ivf) != lines:
            sys.exit(f"{name}: unchanged and patched reference drivers disagree")
    with open(os.path.join(outdir, name + ".md5"), "w") as f:
        f.write("\n".join(lines) + "\n")
    print(f"{name}: {os.path.getsize(ivf)} bytes, {len(lines)} frames, list md5 {hashlib.md5(''.join(lines).encode()).hexdigest()[:8]}")


def make_svc(name):
    w, h, n, seed, dx, dy, patch, layers, intra_at, kbps, speed = SVC_STREAMS[name][:11]
    bits = SVC_STREAMS[name][11] if len(SVC_STREAMS[name]) > 11 else 8
    ivf = os.path.join(HERE, name + ".ivf")
    with tempfile.TemporaryDirectory() as tmp:
        yuv = os.path.join(tmp, "src.yuv")
        source(yuv, w, h, n, seed, dx, dy, bits, "420", patch, 0)
        # (the driver decodes every superframe with the stream oracle and compares the reference buffers with the
        # encoder's own: its `--test-decode=fatal`)
        out = run([os.path.join(VPX, "ref_svc_encode"), yuv, str(w), str(h), str(n), ivf, str(layers), str(intra_at), str(kbps), str(speed), str(bits)])
        if "decoded identically" not in out:
            sys.exit(f"{name}: {out[-400:]}")
    lines = md5_list(os.path.join(VPX, "vpxdec_c"), ivf)
    if len(lines) != n:
        sys.exit(f"{name}: {len(lines)} frames shown, {n} expected")
    with open(os.path.join(HERE, name + ".md5"), "w") as f:
        f.write("\n".join(lines) + "\n")
    print(f"{name}: {os.path.getsize(ivf)} bytes, {len(lines)} shown frames, list md5 {hashlib.md5(''.join(lines).encode()).hexdigest()[:8]}")


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    names = args or ([k for k, v in STREAMS.items() if v[-1] == ("--big" in sys.argv)] + ([] if "--big" in sys.argv else list(SVC_STREAMS)))
    for nm in names:
        if nm in SVC_STREAMS:
            make_svc(nm)
        else:
            make(nm)
