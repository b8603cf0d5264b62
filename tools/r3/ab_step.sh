#!/bin/bash
# usage: tools/r3/ab_step.sh <variant A> <variant B> [reps]: device time per denoising step (bench shape, two chains) of two builds of
# the library, alternating on one box. A variant is the suffix of lib/libdmn_hip_<variant>.so; "-" is the default build.
a=$1; b=$2; reps=${3:-2}
out=gpurun_out/r3_ab_${a}_${b}.txt
: > $out
for r in $(seq $reps); do
  for v in $a $b; do
    if [ "$v" = "-" ]; then unset DMN_LIB_VARIANT; else export DMN_LIB_VARIANT=$v; fi
    timeout -k 10 120 python tools/step_time.py --steps 300 --reps 3 --tag "$v" 2>&1 | grep STEP_US >> $out || exit 1
  done
done
cat $out
