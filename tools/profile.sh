#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root: kernel trace + separate PMC passes of bench.py.
# Usage: tools/profile.sh <tag> [extra bench args]
set -u
TAG=${1:-r01}; shift || true
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
BENCH="python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra --no-pmc $*"
# the trace pass runs the bench default step count so its per-kernel average is comparable with bench.py's own
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --no-cpu-baseline --no-extra --no-pmc $* > $OUT/trace.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -- $BENCH > $OUT/pmc_$c.log 2>&1
done
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM --output-format csv -d $OUT/pmc_sq1 -- $BENCH > $OUT/pmc_sq1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_sq2 -- $BENCH > $OUT/pmc_sq2.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_FLAT SQ_THREAD_CYCLES_VALU SQ_WAVE32_INSTS --output-format csv -d $OUT/pmc_sq3 -- $BENCH > $OUT/pmc_sq3.log 2>&1
find $OUT -name "*.csv" | head -50
python3 tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
