#!/bin/bash
# usage (on the GPU box): alternating A/B runs on ONE box -- boxes differ by 2-4 %, so only such pairs mean anything.
#   tools/ab.sh step  <variant A> <variant B> [reps] ["ENV=VAL ..."]   device time per denoising step (tools/step_time.py: bench
#                     shape, two chains) of two builds of the library; a variant is the suffix of lib/libdmn_hip_<variant>.so
#                     (tools/build_variant.sh), "-" the product build; the optional environment applies to both
#   tools/ab.sh env   <out-file> "ENV=VAL ..." ["ENV=VAL ..." ...]      one step_time.py run per environment setting, the
#                     unchanged build before and after
#   tools/ab.sh train <VAR> <v1> <v2> ...                               configuration-5 training timing per value of a variable
mode=$1; shift
case $mode in
step)
  a=$1; b=$2; reps=${3:-2}; envs=$4
  out=gpurun_out/ab_${a}_${b}.txt
  : > $out
  for r in $(seq $reps); do
    for v in $a $b; do
      if [ "$v" = "-" ]; then unset DMN_LIB_VARIANT; else export DMN_LIB_VARIANT=$v; fi
      env $envs timeout -k 10 120 python tools/step_time.py --steps 300 --reps 3 --tag "$v $envs" 2>&1 | grep STEP_US >> $out || exit 1
    done
  done
  cat $out
  ;;
env)
  out=$1; shift
  : > $out
  timeout -k 10 120 python tools/step_time.py --tag base >> $out 2>&1 || exit 1
  for e in "$@"; do
    env $e timeout -k 10 120 python tools/step_time.py --tag "$e" >> $out 2>&1 || { echo "FAILED $e" >> $out; exit 1; }
  done
  timeout -k 10 120 python tools/step_time.py --tag base-again >> $out 2>&1
  grep STEP_US $out
  ;;
train)
  var=$1; shift
  for v in "$@"; do
    echo "== $var=$v"
    env $var=$v timeout -k 10 120 python tools/train_bench.py --steps 30 --warmup 5 || exit 1
  done
  ;;
*) echo "usage: tools/ab.sh step|env|train ..."; exit 2;;
esac
