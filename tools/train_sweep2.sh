#!/bin/bash
# usage: tools/train_sweep2.sh: configuration-5 timing over chain counts with and without the weight-gradient side streams
cd $GRAFT_REPO_ROOT
for c in 1 2 4; do for sd in 1 0; do
  echo "chains=$c side=$sd" >> gpurun_out/sweep_chains.log
  env DMN_TRAIN_CHAINS=$c DMN_TRAIN_SIDE_STREAM=$sd timeout -k 10 200 python tools/train_bench.py --steps 10 --warmup 3 2>/dev/null | tail -1 >> gpurun_out/sweep_chains.log || exit 1
done; done
