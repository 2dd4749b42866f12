#!/bin/bash
set -e
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_kc; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/sq -o sq -- python3 $ROOT/tools/kconv_quick.py > $OUT/sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM --output-format csv -d $OUT/sq2 -o sq2 -- python3 $ROOT/tools/kconv_quick.py > $OUT/sq2.log 2>&1 || true
rocprofv3 --pmc TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/ta -o ta -- python3 $ROOT/tools/kconv_quick.py > $OUT/ta.log 2>&1 || true
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $ROOT/tools/kconv_quick.py > $OUT/trace.log 2>&1
ls $OUT/*
