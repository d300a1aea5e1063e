#!/bin/bash
# Round-3 end-to-end anchors: the notebooks' recipe (tanh value network, examples/cartpole_balancing.ipynb cells 6, 10, 16) now trains through
# the fused parameter-gradient kernel and the device-driven fit phase.  One JSON line per run into gpurun_out/anchors_r03.jsonl; one run with
# HJBX_FUSED_PARAM_GRAD=0 (PyTorch autograd, the round-2 path for tanh) for the training time beside it.
OUT=gpurun_out/anchors_r03.jsonl
LOG=gpurun_out/anchors_r03_logs
mkdir -p $LOG
: > $OUT
run() {  # name, timeout, args...
  local name=$1 to=$2; shift 2
  timeout -k 10 $to python tools/train_anchor.py "$@" > $LOG/$name.log 2>&1
  local rc=$?
  if [ $rc -eq 0 ]; then tail -1 $LOG/$name.log >> $OUT; else echo "{\"run\": \"$name\", \"rc\": $rc}" >> $OUT; fi
  echo "$name rc=$rc"
  [ $rc -le 1 ] || exit $rc      # a killed GPU step: start no further one
}
for s in 0 1 2; do run cartpole_notebook_tanh_s$s 200 --env cartpole --epochs 100 --notebook --activation tanh --seed $s; done
HJBX_FUSED_PARAM_GRAD=0 run cartpole_notebook_tanh_autograd_s0 200 --env cartpole --epochs 100 --notebook --activation tanh --seed 0
run cartpole_stock_relu_s0 120 --env cartpole --epochs 100 --seed 0
for s in 0 1; do run quad2d_notebook_tanh_s$s 300 --env quadrotors2DHovering --epochs 300 --notebook --activation tanh --seed $s; done
echo anchors done
