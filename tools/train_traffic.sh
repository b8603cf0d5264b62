#!/bin/bash
# usage: tools/train_traffic.sh <tag>: HBM read / write bytes of ONE training step (configuration 5), two counter passes.
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_train_${tag}_$c -- python tools/train_bench.py --steps 1 --warmup 1 > gpurun_out/pmc_train_${tag}_$c.log 2>&1 || exit 1
done
python tools/train_traffic.py gpurun_out/pmc_train_${tag}_FETCH_SIZE gpurun_out/pmc_train_${tag}_WRITE_SIZE gpurun_out/train_traffic_${tag}.json
find gpurun_out/pmc_train_${tag}_* -name "*.csv" -size +20M -delete
echo ok
