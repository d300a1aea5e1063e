#!/bin/bash
# kernel trace of the parameter-gradient kernels (near-hover, B = 2^20) and of the graphed optimiser step at the reference's minibatch
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/prof_train_r02
mkdir -p $OUT
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/full -- python3 tools/dev/time_train.py nearhover 1048576 > $OUT/full.log 2>&1 || exit 1
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/b256 -- python3 tools/dev/time_train.py cartpole 256 > $OUT/b256.log 2>&1 || exit 1
echo done
