import os, sys
ROOT='/root/repo'
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g
hip = g.load_pkg()
import bench
for name in ("S-1440", "S-1080-8"):
    ivf = os.path.join(ROOT, "tests", "streams_big", name + ".ivf")
    r = bench.real_stream_resident(hip, ivf, replay=(0, 1, 2), reps=30)
    print(name, [(f["index"], f["blocks"], f["filter_level"], round(1e6 / f["frames_per_s"], 1)) for f in r["frames"]], "us per frame")
