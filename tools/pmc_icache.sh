#!/bin/bash
# Instruction-cache counters of the frame's kernels (phases in sequence; rocprofv3 --pmc serialises kernels).
# Usage (GPU box, repo root): bash tools/pmc_icache.sh  -> gpurun_out/pmc_icache/
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_icache
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_IFETCH --output-format csv -d $OUT/ic -o ic -- python3 $ROOT/tools/profile_phase.py --separate --steps 6 > $OUT/ic.log 2>&1
python3 - <<PY
import csv, glob, collections
per = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/ic/**/*counter_collection.csv", recursive=True):
    d = collections.defaultdict(float); names = {}
    for r in csv.DictReader(open(f)):
        d[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"]); names[r["Dispatch_Id"]] = r["Kernel_Name"][:60]
    for (i, c), v in d.items():
        per[names[i]][c].append(v)
for k, cs in per.items():
    print(k, {c: round(sum(v) / len(v)) for c, v in cs.items()})
PY
