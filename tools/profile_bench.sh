#!/bin/bash
# rocprofv3 --kernel-trace --stats of bench.py itself (the per-kernel averages bench.py's HIP-event figures must
# agree with) + the PMC passes.  On the GPU box, from the repo root:  bash tools/profile_bench.sh r02
set -e
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -o bench -- python3 $ROOT/bench.py --no-cpu-baseline --no-stream --streams 0 > $OUT/bench.json 2> $OUT/bench.err
find $OUT/bench -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $ROOT/gpurun_out/${TAG}_bench_kernel_stats.csv
cp $OUT/bench.json $ROOT/gpurun_out/${TAG}_bench_under_rocprof.json
cd $ROOT && bash tools/pmc_collect.sh $TAG && python3 tools/pmc_summary.py gpurun_out/pmc_$TAG gpurun_out/${TAG}_pmc.json > /dev/null
echo done
