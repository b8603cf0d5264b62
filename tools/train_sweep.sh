#!/bin/bash
# usage: tools/train_sweep.sh <env var> <values...>: the configuration-5 timing under each value of a tuning variable
var=$1; shift
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  echo "$var=$v" >> gpurun_out/sweep_$var.log
  env $var=$v timeout -k 10 200 python tools/train_bench.py --steps 10 --warmup 3 2>/dev/null | tail -1 >> gpurun_out/sweep_$var.log || exit 1
done
