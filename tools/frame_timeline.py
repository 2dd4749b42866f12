#!/usr/bin/env python3
"""Start / end of every kernel of a few consecutive frames from a rocprofv3 --kernel-trace CSV of bench.py:
    rocprofv3 --kernel-trace --output-format csv -d out -o tl -- python3 bench.py --no-cpu-baseline --no-stream --streams 0 --no-phase-timers
    python3 tools/frame_timeline.py out/**/tl_kernel_trace.csv [frame_index]"""
import csv, sys
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1])))
def short(n):
    for k, s in (("walk_lf", "WALK+LF"), ("lf_rows2", "LF"), ("intra_island", "ISL"), ("inter_reg", "CONV"), ("intra_residual", "RES"),
                 ("idct_add", "IDCT"), ("fillBuffer", "FILL"), ("copyBuffer", "COPY")):
        if k in n:
            return s
    return n[:12]
convs = [i for i, e in enumerate(ev) if "inter_reg" in e[2]]
i0 = convs[int(sys.argv[2]) if len(sys.argv) > 2 else len(convs) // 2]
t0 = ev[i0][0]
for s, e, n in ev[i0:i0 + 12]:
    print(f"{short(n):8s} start {(s - t0) / 1e3:8.1f} us  end {(e - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:7.1f}")
