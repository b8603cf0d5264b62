#!/bin/bash
# usage: tools/gpu_check.sh <tag> [timesteps]   -- GPU tests, then a rocprofv3 kernel trace of a short bench run
tag=$1; ts=${2:-100}
timeout -k 10 600 python -u -m pytest tests -x -q -m gpu --timeout 300 --timeout-method thread > gpurun_out/test_$tag.log 2>&1
rc=$?; echo exit=$rc >> gpurun_out/test_$tag.log; tail -4 gpurun_out/test_$tag.log
[ $rc -ne 0 ] && exit $rc
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python bench.py --steps 1 --warmup 1 --timesteps $ts --no-cpu-baseline --no-train > gpurun_out/prof_$tag.log 2>&1 || exit 1
grep -v "^[WE]2026" gpurun_out/prof_$tag.log | tail -1 | cut -c 1-220
