#!/bin/bash
# Development aid (run HERE, no GPU needed): build variants of libhjbx.so that differ only in hjbx_train_coop.o (cartpole-only dev build with
# timing switches) into build/dev/, to be timed on the GPU box with
#   HJBX_LIBRARY=build/dev/libhjbx_<tag>.so python tools/dev/time_coop.py
# usage: tools/dev/coop_variants.sh tag1:"-DFLAG ..." tag2:"..."
set -e
cd "$(dirname "$0")/../.."
C=q_learning_with_hjb_amd/csrc
mkdir -p build/dev
OBJS="$C/hjbx_kernels.o $C/hjbx_mlp_relu.o $C/hjbx_mlp_tanh.o $C/hjbx_mlp_x3.o $C/hjbx_mlp_h2.o $C/hjbx_mlp_sin.o $C/hjbx_train.o $C/hjbx_fit.o $C/hjbx_user.o"
for spec in "$@"; do
  tag="${spec%%:*}"; flags="${spec#*:}"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=on -fPIC -fno-slp-vectorize -DHJBX_TRAIN_DEV $flags -c $C/hjbx_train_coop.hip -o build/dev/coop_$tag.o &
done
wait
for spec in "$@"; do
  tag="${spec%%:*}"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/dev/libhjbx_$tag.so $OBJS build/dev/coop_$tag.o
  rm build/dev/coop_$tag.o
  echo built build/dev/libhjbx_$tag.so
done
