#!/bin/bash
# usage: tools/r3/ab_env.sh <variant A> <variant B> "<ENV=VAL ...>" [reps]: like ab_step.sh with extra environment for both builds
a=$1; b=$2; envs=$3; reps=${4:-2}
for r in $(seq $reps); do
  for v in $a $b; do
    if [ "$v" = "-" ]; then unset DMN_LIB_VARIANT; else export DMN_LIB_VARIANT=$v; fi
    env $envs timeout -k 10 120 python tools/step_time.py --steps 300 --reps 3 --tag "$v [$envs]" 2>&1 | grep STEP_US || exit 1
  done
done
