#!/bin/bash
# Does the stock quadrotor training still reach what it reached in round 2 (profiles/r02_anchors.json: f32 rollouts, 36 seeds x 300 epochs:
# final trajectory length > 60 in 22, closed-loop cost < 10 in 9) now that the Adam step rides in the mix kernel and the fit phase is driven
# from the device?  12 seeds x 300 epochs, default (float32) arithmetic.  One JSON line per run.
OUT=${SWEEP_OUT:-gpurun_out/anchors_r03_sweep.jsonl}
LOG=gpurun_out/anchors_r03_logs
mkdir -p $LOG
: > $OUT
for s in ${SEEDS:-0 1 2 3 4 5 6 7 8 9 10 11}; do
  timeout -k 10 300 python tools/train_anchor.py --env quadrotors2DHovering --epochs 300 --seed $s > $LOG/sweep_s$s.log 2>&1
  rc=$?
  if [ $rc -eq 0 ]; then tail -1 $LOG/sweep_s$s.log >> $OUT; else echo "{\"run\": \"sweep_s$s\", \"rc\": $rc}" >> $OUT; fi
  echo "seed $s rc=$rc"
  [ $rc -le 1 ] || exit $rc
done
echo sweep done
