"""Warm-loop dx_time fps (vpxdec --summary) of the three big synthesized streams through vpxdec_c / vpxdec_hip /
vpxdec_hip_mt, the way bench.py's stream leg measures S-1440.  GPU box, from the repo root:
    python tools/stream_fps.py > gpurun_out/stream_fps.json"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    out = {}
    for s in ("S-1440", "S-1440-q44", "S-1440-10", "S-1080-8", "S-1080-10", "S-2160", "S-2176"):
        ivf = os.path.join(ROOT, "tests", "streams_big", s + ".ivf")
        if not os.path.exists(ivf):
            continue
        want = [l.rstrip("\n") for l in open(ivf[:-4] + ".md5") if l.strip()]
        row = {"frames": len(want)}
        for name, path in (("vpxdec_hip_mt", "shim/build/vpxdec_hip_mt"), ("vpxdec_hip", "shim/build/vpxdec_hip"),
                           ("vpxdec_c", "oracle/_ref/vpx/vpxdec_c")):
            dec = os.path.join(ROOT, path)
            if not os.path.exists(dec):
                continue
            if name != "vpxdec_c":
                row[name + "_md5_equal"] = bench.run_vpxdec(dec, ivf, md5=True) == want
            runs = bench.run_vpxdec(dec, ivf, loops=1 if name == "vpxdec_c" else 5)
            warm = runs[1:] or runs
            row[name + "_fps"] = round(sum(f for _, f in warm) / len(warm), 1)
            row[name + "_fps_runs"] = [f for _, f in runs]
        if os.path.exists(os.path.join(ROOT, "cuda-vp9_amd", "vp9hip_dec")):
            row["vp9hip_dec_md5_equal"] = bench.run_own_dec(ivf, md5=True) == want
            runs = bench.run_own_dec(ivf, loops=5)
            row["vp9hip_dec_fps"] = round(sum(f for _, f in runs[1:]) / max(1, len(runs[1:])), 1)
            row["vp9hip_dec_fps_runs"] = [f for _, f in runs]
        out[s] = row
        print(s, json.dumps(row), file=sys.stderr, flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
