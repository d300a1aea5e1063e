#!/bin/bash
# counters of the bf16x3 rollout kernel (one pass per counter group), then a kernel trace
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/pmc_${1:-bf16x3}
mkdir -p $OUT
R="rocprofv3 --output-format csv"
A="bench.py --arithmetic ${1:-bf16x3} --no-secondary --no-cpu-baseline --prewarm 0 --steps 100 --warmup 100 --reps 3"
$R --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/a -- python3 $A > $OUT/a.json 2> $OUT/a.err || exit 1
$R --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY -d $OUT/b -- python3 $A > $OUT/b.json 2> $OUT/b.err || exit 1
$R --kernel-trace --stats -d $OUT/t -- python3 $A > $OUT/t.json 2> $OUT/t.err || exit 1
echo done
