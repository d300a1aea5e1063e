#!/bin/bash
# Is the outcome of the (chaotic) stock quadrotor training sensitive to the value-network arithmetic?  12 seeds x 300 epochs in each.
OUT=${SWEEP_OUT:-gpurun_out/anchors_r02_sweep.jsonl}
LOG=gpurun_out/anchors_r02_logs
mkdir -p $LOG
: > $OUT
for a in f32 f16x2; do
  for s in ${SEEDS:-0 1 2 3 4 5 6 7 8 9 10 11}; do
    timeout -k 10 300 python tools/train_anchor.py --env quadrotors2DHovering --epochs 300 --seed $s --arithmetic $a > $LOG/sweep_${a}_s$s.log 2>&1
    rc=$?
    if [ $rc -eq 0 ]; then tail -1 $LOG/sweep_${a}_s$s.log >> $OUT; else echo "{\"run\": \"sweep_${a}_s$s\", \"rc\": $rc}" >> $OUT; fi
    [ $rc -le 1 ] || exit $rc
  done
done
echo sweep done
