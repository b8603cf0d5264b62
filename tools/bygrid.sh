#!/bin/bash
# usage: tools/bygrid.sh <tag> [extra bench args]: a short two-chain bench under rocprofv3 --kernel-trace, summarised per (kernel, grid)
tag=$1; shift
o=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $o/${tag}_prof -- python bench.py --steps 1 --warmup 1 --timesteps 300 --no-train --no-cpu-baseline "$@" > $o/${tag}_under_rocprof.log 2>&1 || { tail -5 $o/${tag}_under_rocprof.log; exit 1; }
for t in $(find $o/${tag}_prof -name "*_kernel_trace.csv"); do python tools/trace_by_grid.py $t > $o/${tag}_by_grid.csv; done
find $o/${tag}_prof -name "*_kernel_trace.csv" -size +20M -delete
head -60 $o/${tag}_by_grid.csv | cut -c 1-170
