#!/bin/bash
# HBM bytes of the parameter-gradient kernels (near-hover, B = 2^20): separate FETCH_SIZE / WRITE_SIZE passes
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/prof_train_pmc_r02
mkdir -p $OUT
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --output-format csv --pmc $C -d $OUT/$C -- python3 tools/dev/time_train.py nearhover 1048576 > $OUT/$C.log 2>&1 || exit 1
done
echo done
