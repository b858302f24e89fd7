#!/bin/bash
# On the GPU box: kernel trace + the SQ counter passes of bench.py at another window / shape (no HBM-traffic passes).
# Usage: tools/profile_window.sh <tag> <bench args...>      e.g.  r02_2048 --window 2048 --channels-per-gpu 4096 --frames 32
set -u
TAG=$1; shift
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
ARGS="--steps 10 --warmup 2 --no-cpu-baseline --no-extra $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $ARGS > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM --output-format csv -d $OUT/pmc_sq1 -- python3 bench.py $ARGS > $OUT/pmc_sq1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_sq2 -- python3 bench.py $ARGS > $OUT/pmc_sq2.log 2>&1
python3 tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
grep "^{" $OUT/trace.log | tail -1 > $OUT/bench.json
cat $OUT/summary.txt
