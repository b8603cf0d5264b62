#!/bin/bash
# usage: tools/train_kernel_times.sh <pattern>   (on the GPU box): average duration of the training kernels whose name matches
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/kt
rm -rf $o
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o -- python tools/train_bench.py --steps 3 --warmup 1 > $o.log 2>&1 || { tail -5 $o.log; exit 1; }
for t in $(find $o -name "*_kernel_stats.csv"); do grep -E "$1" $t | cut -d, -f1-4 | cut -c 1-140; done
