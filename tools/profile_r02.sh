#!/bin/bash
# Round-2 profiling recipe (run on the GPU box from the repo root; outputs under gpurun_out/prof_r02, summaries are copied to profiles/ by
# tools/summarize_r02.py).  rocprofv3 gets the program directly after `--`; counters are collected in their own passes.
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/prof_r02
mkdir -p $OUT
R="rocprofv3 --output-format csv"
# 1. kernel trace + stats of the bench command itself
$R --kernel-trace --stats -d $OUT/trace -- python3 bench.py > $OUT/bench_trace.json 2> $OUT/bench_trace.err || exit 1
# 2. HBM traffic of the rollout kernel at two launch lengths (every launch of a run has the same length: warm-up = K, no pre-warm)
for K in 100 20; do
  for C in FETCH_SIZE WRITE_SIZE; do
    $R --pmc $C -d $OUT/pmc$K/$C -- python3 bench.py --no-secondary --no-cpu-baseline --prewarm 0 --steps $K --warmup $K --reps 3 > $OUT/pmc${K}_$C.json 2> $OUT/pmc${K}_$C.err || exit 1
  done
done
$R --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/pmc100/SQ -- python3 bench.py --no-secondary --no-cpu-baseline --prewarm 0 --steps 100 --warmup 100 --reps 3 > $OUT/pmc100_SQ.json 2> $OUT/pmc100_SQ.err || exit 1
# 3. the HBM-bound entry points: durations and counter bytes with buffers rotated through a 640 MB pool
$R --kernel-trace --stats -d $OUT/kb/trace -- python3 tools/kernel_bench.py --systems cartpole,acrobot,quad2d,nearhover --no-rollouts --json $OUT/kb.json > $OUT/kb.log 2>&1 || exit 1
for C in FETCH_SIZE WRITE_SIZE; do
  $R --pmc $C -d $OUT/kb/$C -- python3 tools/kernel_bench.py --systems cartpole,acrobot,quad2d,nearhover --no-rollouts > $OUT/kb_$C.log 2>&1 || exit 1
done
echo profile_r02 done
