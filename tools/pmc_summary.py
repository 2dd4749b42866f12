#!/usr/bin/env python3
"""Summarise the passes of tools/pmc_collect.sh into one JSON: per kernel family the mean per-launch
counters, HBM traffic (FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts a
128-byte request as 64 bytes for wide coalesced reads — MI355X_MICROARCH.md §HBM — so both the raw
and the doubled figure are given), LDS bank-conflict rate and the wave-time split.
    python3 tools/pmc_summary.py gpurun_out/pmc_<tag> profiles/<name>.json"""
import collections
import csv
import datetime
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the kernel sources a family's record belongs to (bench.py quotes a record only while this hash matches)
SOURCES = {"convolve": ["inter_kernels.hip"], "idct_add": ["txfm_kernels.hip", "txfm_device.h"],
           "intra": ["intra_kernels.hip", "txfm_device.h"], "intra_residual": ["intra_kernels.hip", "txfm_device.h"],
           "loop_filter": ["lf_kernels.hip"], "walk_lf": ["lf_kernels.hip", "intra_kernels.hip", "txfm_device.h"]}


def source_hash(files):
    h = hashlib.sha256()
    for f in files:
        with open(os.path.join(ROOT, "cuda-vp9_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]

FAMILY = (("inter_fast_kernel", "convolve"), ("inter_fast16_kernel", "convolve"), ("inter_reg_kernel", "convolve"), ("inter_reg16_kernel", "convolve"), ("inter_pred_kernel", "convolve_generic"), ("idct_add", "idct_add"),
          ("walk_lf_kernel", "walk_lf"), ("intra_island_kernel", "intra"), ("intra_residual_kernel", "intra_residual"), ("intra_wave_kernel", "intra_waves"), ("lf_rows", "loop_filter"),
          ("lf_diag", "loop_filter_diag"), ("residual_", "residual"))


def family(name):
    for key, fam in FAMILY:
        if key in name:
            return fam
    return None


def main(src, dst):
    per = collections.defaultdict(lambda: collections.defaultdict(list))  # fam -> counter -> per-dispatch values
    for f in glob.glob(os.path.join(src, "*", "**", "*counter_collection.csv"), recursive=True):
        disp = collections.defaultdict(float)
        names = {}
        for r in csv.DictReader(open(f)):
            fam = family(r["Kernel_Name"])
            if fam is None:
                continue
            key = (r["Dispatch_Id"], r["Counter_Name"])
            disp[key] += float(r["Counter_Value"])
            names[r["Dispatch_Id"]] = fam
        for (d, c), v in disp.items():
            per[names[d]][c].append(v)
    dur = collections.defaultdict(list)
    for f in glob.glob(os.path.join(src, "trace", "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            fam = family(r["Kernel_Name"])
            if fam:
                dur[fam].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
    out = {}
    for fam, counters in sorted(per.items()):
        m = {c: sum(v) / len(v) for c, v in counters.items()}
        e = {"launches_sampled": max(len(v) for v in counters.values()), "counters_mean_per_launch": {k: round(v, 1) for k, v in sorted(m.items())}}
        if fam in dur:
            e["mean_duration_us"] = round(sum(dur[fam]) / len(dur[fam]), 2)
        if "FETCH_SIZE" in m:
            e["hbm_read_bytes_raw"] = int(m["FETCH_SIZE"] * 1024)
            e["hbm_read_bytes_gfx950_x2"] = int(m["FETCH_SIZE"] * 2048)
        if "WRITE_SIZE" in m:
            e["hbm_write_bytes"] = int(m["WRITE_SIZE"] * 1024)
        if m.get("SQ_LDS_IDX_ACTIVE"):
            e["lds_bank_conflict_frac"] = round(m.get("SQ_LDS_BANK_CONFLICT", 0.0) / m["SQ_LDS_IDX_ACTIVE"], 4)
        if m.get("SQ_WAVE_CYCLES"):
            e["wave_time_split"] = {k: round(m.get(c, 0.0) / m["SQ_WAVE_CYCLES"], 3) for k, c in
                                    (("waiting", "SQ_WAIT_ANY"), ("issue_stall", "SQ_WAIT_INST_ANY"), ("issuing", "SQ_ACTIVE_INST_ANY"))}
        if m.get("TCC_HIT_sum") is not None and (m.get("TCC_HIT_sum", 0) + m.get("TCC_MISS_sum", 0)) > 0:
            e["l2_hit_rate"] = round(m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"]), 3)
        if fam in SOURCES:
            e["source_sha256"] = source_hash(SOURCES[fam])
        e["collected"] = datetime.date.today().isoformat() + " tools/pmc_collect.sh (blockgen 1440p frame, phases in separate runs)"
        out[fam] = e
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk != "counters_mean_per_launch"} for k, v in out.items()}, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
