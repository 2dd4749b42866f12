#!/usr/bin/env python3
"""GPU-side rate on frames of a real stream with their lists resident (bench.py's real_stream_resident), for A/B of
library builds:  VP9HIP_TOOLS_LIB=tools/build/libX.so python tools/real_replay.py [stream name]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g
hip = g.load_pkg()
if os.environ.get("VP9HIP_TOOLS_LIB"):
    hip.LIB_PATH = os.path.join(ROOT, os.environ["VP9HIP_TOOLS_LIB"])
import bench
name = sys.argv[1] if len(sys.argv) > 1 else "S-1440"
ivf = os.path.join(ROOT, "tests", "streams_big", name + ".ivf")
frames = tuple(range(3, 60, 4)) if name != "S-2176" else tuple(range(1, 12, 2))
r = bench.real_stream_resident(hip, ivf, replay=frames, reps=100)
lf = [f for f in r["frames"] if f["filter_level"] > 0]
print(name, "harmonic mean", r["frames_per_s"], "| frames with a loop filter:", [(f["index"], f["blocks"], f["frames_per_s"]) for f in lf])
