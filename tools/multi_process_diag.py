#!/usr/bin/env python3
"""N vp9hip_dec processes side by side on one stream with --md5: which frames of which loop differ from the golden list.
    python tools/multi_process_diag.py <name> <processes> <threads> [loops] [extra vp9hip_dec options...]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name, n, thr = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
loops = int(sys.argv[4]) if len(sys.argv) > 4 else 3
extra = sys.argv[5:]
for base in ("tests/streams_big", "tests/golden/streams"):
    ivf = os.path.join(ROOT, base, name + ".ivf")
    if os.path.exists(ivf):
        break
want = [l.split()[0] for l in open(ivf[:-4] + ".md5") if l.strip()]
exe = os.path.join(ROOT, "cuda-vp9_amd", "vp9hip_dec")
procs = [subprocess.Popen([exe, "--md5", "-o", "img-%wx%h-%4.i420", f"--loops={loops}", f"--threads={thr}"] + extra + [ivf],
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE) for _ in range(n)]
for i, p in enumerate(procs):
    out, err = p.communicate(timeout=600)
    got = [l.split()[0] for l in out.decode().splitlines() if l.strip()]
    bad = [(k // len(want), k % len(want)) for k, h in enumerate(got) if h != want[k % len(want)]]
    print(f"proc {i}: rc {p.returncode}, {len(got)} lines, mismatches (loop, frame): {bad[:24]}{' ...' if len(bad) > 24 else ''} {err.decode(errors='replace')[-200:].strip()}")
