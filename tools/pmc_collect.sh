#!/bin/bash
# Collects the PMC counters of the four kernel families in separate passes (gfx950: FETCH_SIZE and
# WRITE_SIZE do not fit one pass; SQ counters in their own pass), as MI355X_MICROARCH.md prescribes.
# Usage (on the GPU box, from the repo root):  bash tools/pmc_collect.sh <tag>
# Output: gpurun_out/pmc_<tag>/{fetch,write,sq,sq2}/...counter_collection.csv  -> tools/pmc_summary.py
set -e
TAG=${1:-run}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $ROOT/tools/profile_phase.py --steps 6 --separate"
TRACE_CMD="python3 $ROOT/tools/profile_phase.py --steps 20 --separate"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o fetch -- $CMD > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o write -- $CMD > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/sq -o sq -- $CMD > $OUT/sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/sq2 -o sq2 -- $CMD > $OUT/sq2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $TRACE_CMD > $OUT/trace.log 2>&1
tail -2 $OUT/trace.log
