#!/bin/bash
# counters of the fused near-hover RK4 rollout kernel (BASELINE configs[4]) in the default arithmetic
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/prof_nearhover_r02
mkdir -p $OUT
A="bench.py --system nearhover --integrator rk4 --no-secondary --no-cpu-baseline --prewarm 0 --steps 100 --warmup 100 --reps 3"
rocprofv3 --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAVE_CYCLES -d $OUT/a -- python3 $A > $OUT/a.json 2> $OUT/a.err || exit 1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --output-format csv --pmc $C -d $OUT/$C -- python3 $A > $OUT/$C.json 2> $OUT/$C.err || exit 1
done
echo done
