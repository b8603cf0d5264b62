#!/bin/bash
# usage: tools/profile_round.sh <tag>
# 1. rocprofv3 kernel trace + stats of the bench command (one timed 1000-step chain), 2. two counter passes
# (FETCH_SIZE, WRITE_SIZE; separate runs, kernel trace only) over a short chain plus the dominant-kernel loop.
# Leaves the summaries under gpurun_out/prof_<tag>_*; the big per-dispatch trace of pass 1 is deleted.
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_${tag}
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d ${out}_bench -- python bench.py --steps 1 --warmup 1 --no-train > ${out}_bench.log 2>&1 || exit 1
grep -v "^[WE]2026" ${out}_bench.log | tail -1 > ${out}_bench.json
for t in $(find ${out}_bench -name "*_kernel_trace.csv"); do python tools/trace_by_grid.py $t > ${out}_bench_by_grid.csv; done
find ${out}_bench -name "*_kernel_trace.csv" -size +20M -delete
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d ${out}_pmc_$c -- python bench.py --steps 1 --warmup 0 --timesteps 10 --no-cpu-baseline --no-train > ${out}_pmc_$c.log 2>&1 || exit 1
done
echo profiled
