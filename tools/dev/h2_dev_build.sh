#!/bin/bash
# development build: f16x2 unit with cartpole (+ quad2d value_grad) only, linked with the other (full) objects
cd /root/repo/q_learning_with_hjb_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=on -fPIC -DHJBX_MLP_ACT=3 -fno-slp-vectorize -DHJBX_MLP_DEV -DHJBX_MLP_DEV_QUAD2D $HJBX_DEV_FLAGS -c hjbx_mlp.hip -o /tmp/hjbx_mlp_h2_dev.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libhjbx.so hjbx_kernels.o hjbx_mlp_relu.o hjbx_mlp_tanh.o hjbx_mlp_x3.o /tmp/hjbx_mlp_h2_dev.o hjbx_train.o && echo dev lib linked
