#!/bin/bash
# Round-2 end-to-end anchors with the final code (fused parameter gradient, f16x2 rollouts): one JSON line per run into
# gpurun_out/anchors_r02.jsonl (the training log of each run goes to gpurun_out/anchors_r02_logs/).  Each run is bounded by its own timeout.
OUT=gpurun_out/anchors_r02.jsonl
LOG=gpurun_out/anchors_r02_logs
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
run cartpole_stock_relu_s0 120 --env cartpole --epochs 100 --seed 0
for s in 0 1 2 3 4 5; do run quad2d_stock_300_s$s 300 --env quadrotors2DHovering --epochs 300 --seed $s; done
for s in 0 1; do run nearhover_stock_200_s$s 400 --env nearHoverQuadcopter --epochs 200 --seed $s --T 20; done
run acrobot_warmstart_s0 300 --env acrobot --epochs 100 --seed 0 --warm_start 64
echo anchors done
