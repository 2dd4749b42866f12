#!/bin/bash
# Round-3 evidence, on the GPU box from the repo root:  bash tools/profile_r03.sh
#  1. rocprofv3 --kernel-trace --stats of bench.py itself (per-kernel averages next to bench.py's HIP-event figures)
#  2. the PMC passes of the bench frame's launches (tools/pmc_collect.sh -> tools/pmc_summary.py)
#  3. the K-conv grid (tools/kconv_bench.py) and the rocprofv3 --kernel-trace --stats summary of the same command
# Everything lands under gpurun_out/ (copied to profiles/ by hand afterwards).
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_r03
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -o bench -- python3 $ROOT/bench.py --no-cpu-baseline --no-stream --streams 0 --no-pmc > $OUT/bench.json 2> $OUT/bench.err
find $OUT/bench -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $ROOT/gpurun_out/r03_bench_kernel_stats.csv
cp $OUT/bench.json $ROOT/gpurun_out/r03_bench_under_rocprof.json
echo "bench under rocprofv3 done"
cd $ROOT && bash tools/pmc_collect.sh r03 && python3 tools/pmc_summary.py gpurun_out/pmc_r03 gpurun_out/r03_pmc.json > /dev/null
echo "pmc done"
cd /tmp
python3 $ROOT/tools/kconv_bench.py > $OUT/kconv.log 2>&1
cp $ROOT/gpurun_out/kconv.json $ROOT/gpurun_out/r03_kconv.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kconv -o kconv -- python3 $ROOT/tools/kconv_bench.py > $OUT/kconv_prof.log 2>&1
find $OUT/kconv -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $ROOT/gpurun_out/r03_kconv_kernel_stats.csv
find $OUT/kconv -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $ROOT/gpurun_out/r03_kconv_kernel_trace.csv
echo done
