#!/bin/bash
# development helper: cartpole-only assembly of the bf16x3 variant + resource usage
cd /root/repo/q_learning_with_hjb_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=on -S --cuda-device-only -DHJBX_MLP_ACT=${1:-2} -fno-slp-vectorize -DHJBX_MLP_DEV -o /tmp/x3.s hjbx_mlp.hip -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|Function Name|VGPRs|Spill|LDS Size|ScratchSize" | sed -e 's/\[-Rpass.*//' -e 's/hjbx_mlp.hip:[0-9]*:1: remark: //' | cut -c1-110
