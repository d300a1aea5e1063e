#!/bin/bash
# Round-3 profiling recipe (run on the GPU box from the repo root; outputs under gpurun_out/prof_r03, summaries are copied to profiles/ by
# tools/summarize_r03.py).  rocprofv3 gets the program directly after `--`; counters are collected in their own passes.
# usage: tools/profile_r03.sh [A|B|all]   (two halves that each fit one 20-minute GPU call; the summarizer runs on the merged outputs, anywhere)
set -o pipefail
STAGE=${1:-all}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_r03
mkdir -p $OUT
R="timeout -k 10 400 rocprofv3 --output-format csv"   # (a profiled run that aborts can hang in the profiler's signal handler: bounded)
if [ $STAGE != B ]; then
# 1. kernel trace + stats of the bench command itself (default arithmetic = float32 MFMA; its secondary block runs the opt-in kernels too)
$R --kernel-trace --stats -d $OUT/trace -- python3 bench.py > $OUT/bench_trace.json 2> $OUT/bench_trace.err || exit 1
# 2. HBM traffic of the rollout kernel at two launch lengths (every launch of a run has the same length: warm-up = K, no pre-warm)
for K in 100 20; do
  for C in FETCH_SIZE WRITE_SIZE; do
    $R --pmc $C -d $OUT/pmc$K/$C -- python3 bench.py --no-secondary --no-cpu-baseline --prewarm 0 --steps $K --warmup $K --reps 3 > $OUT/pmc${K}_$C.json 2> $OUT/pmc${K}_$C.err || exit 1
  done
done
# 3. matrix-pipe / issue counters of the float32 rollout kernel
$R --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/sq_f32/a -- python3 bench.py --no-secondary --no-cpu-baseline --prewarm 0 --steps 100 --warmup 100 --reps 3 > $OUT/sq_f32_a.json 2> $OUT/sq_f32_a.err || exit 1
$R --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY -d $OUT/sq_f32/b -- python3 bench.py --no-secondary --no-cpu-baseline --prewarm 0 --steps 100 --warmup 100 --reps 3 > $OUT/sq_f32_b.json 2> $OUT/sq_f32_b.err || exit 1
# 7. the plain bench line (no profiler attached)
python3 bench.py > $OUT/r03_bench_builder_run.json 2> $OUT/bench_builder_run.err || exit 1
fi
if [ $STAGE != A ]; then
# 4. the parameter gradient: durations of its three implementations, HBM bytes and matrix-pipe occupancy of the cooperative kernel
$R --kernel-trace --stats -d $OUT/train/trace -- python3 tools/dev/time_train.py nearhover 1048576 > $OUT/train_trace.log 2>&1 || exit 1
for C in FETCH_SIZE WRITE_SIZE; do
  $R --pmc $C -d $OUT/train/$C -- python3 tools/dev/time_coop.py > $OUT/train_$C.log 2>&1 || exit 1
done
$R --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/train/sq -- python3 tools/dev/time_coop.py > $OUT/train_sq.log 2>&1 || exit 1
# 5. the HBM-bound entry points: durations and counter bytes with buffers rotated through a 640 MB pool
$R --kernel-trace --stats -d $OUT/kb/trace -- python3 tools/kernel_bench.py --systems cartpole,acrobot,quad2d,nearhover --no-rollouts --json $OUT/kb.json > $OUT/kb.log 2>&1 || exit 1
for C in FETCH_SIZE WRITE_SIZE; do
  $R --pmc $C -d $OUT/kb/$C -- python3 tools/kernel_bench.py --systems cartpole,acrobot,quad2d,nearhover --no-rollouts > $OUT/kb_$C.log 2>&1 || exit 1
done
# 6. the fit phase at the reference's minibatch: per-kernel durations inside the replayed graph
$R --kernel-trace --stats -d $OUT/fit/trace -- python3 tools/dev/time_fit.py cartpole > $OUT/fit_trace.log 2>&1 || exit 1
f=$(find $OUT/fit/trace -name "*kernel_trace.csv" | head -1)
python3 tools/dev/trace_gaps.py $f 1500 > $OUT/r03_fit_phase_kernels.txt 2>&1
fi
[ $STAGE = all ] && python3 tools/summarize_r03.py $OUT $OUT > $OUT/summary.log 2>&1
echo profile_r03 done
