#!/bin/bash
# usage: tools/ab_paths.sh <tag> <mask A> <mask B>: bench (two chains and one chain) with two DMN_UNET_PATHS settings on
# one box, then a one-chain kernel trace of setting A with its step timeline.
tag=$1; A=$2; B=$3
o=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for m in $A $B $A $B; do
  for ch in 2 1; do
    DMN_UNET_PATHS=$m DMN_SAMPLE_CHAINS=$ch timeout -k 10 200 python bench.py --no-train --no-cpu-baseline --steps 2 > $o/${tag}_m${m}_c${ch}.log 2>&1 || { tail -5 $o/${tag}_m${m}_c${ch}.log; exit 1; }
    echo "mask $m chains $ch: $(grep -v '^[WE]2026' $o/${tag}_m${m}_c${ch}.log | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["step"]["device_ms"])')"
  done
done
DMN_UNET_PATHS=$A DMN_SAMPLE_CHAINS=1 timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $o/${tag}_prof_single -- python bench.py --steps 1 --warmup 1 --timesteps 200 --no-train --no-cpu-baseline > $o/${tag}_single_under_rocprof.log 2>&1 || { tail -5 $o/${tag}_single_under_rocprof.log; exit 1; }
for t in $(find $o/${tag}_prof_single -name "*_kernel_trace.csv"); do python tools/step_timeline.py $t > $o/${tag}_step_timeline.txt; done
find $o/${tag}_prof_single -name "*_kernel_trace.csv" -size +20M -delete
tail -40 $o/${tag}_step_timeline.txt
