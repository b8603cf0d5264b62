#!/bin/bash
# usage: tools/r2_check.sh <tag>  -- GPU tests (incl. the uncollected allocation-free test), default bench, then kernel
# traces of a 100-step chain of the DDPM and of the improved-DDPM configuration
tag=$1
o=gpurun_out
timeout -k 10 900 python -u -m pytest tests -x -q -m gpu --timeout 300 --timeout-method thread > $o/test_$tag.log 2>&1
rc=$?; echo exit=$rc >> $o/test_$tag.log; tail -5 $o/test_$tag.log
[ $rc -ne 0 ] && exit $rc
if [ -f tests/wip_no_alloc.py ]; then
  timeout -k 10 300 python -u -m pytest tests/wip_no_alloc.py -x -q --timeout 120 --timeout-method thread > $o/test_noalloc_$tag.log 2>&1
  rc=$?; echo exit=$rc >> $o/test_noalloc_$tag.log; tail -5 $o/test_noalloc_$tag.log
  [ $rc -ne 0 ] && exit $rc
fi
timeout -k 10 600 python bench.py --no-train > $o/bench_$tag.log 2>&1 || { tail -5 $o/bench_$tag.log; exit 1; }
grep -v "^[WE]2026" $o/bench_$tag.log | tail -1 > $o/bench_$tag.json; cut -c 1-400 $o/bench_$tag.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in ddpm improved; do
  extra=""; [ $cfg = improved ] && extra="--learned-variance"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_${tag}_$cfg -- python bench.py --steps 1 --warmup 1 --timesteps 100 --no-cpu-baseline --no-train $extra > $o/prof_${tag}_$cfg.log 2>&1 || { tail -5 $o/prof_${tag}_$cfg.log; exit 1; }
  for t in $(find $o/prof_${tag}_$cfg -name "*_kernel_trace.csv"); do python tools/trace_by_grid.py $t > $o/prof_${tag}_${cfg}_by_grid.csv; done
  find $o/prof_${tag}_$cfg -name "*_kernel_trace.csv" -size +20M -delete
  grep -v "^[WE]2026" $o/prof_${tag}_$cfg.log | tail -1 | cut -c 1-300
done
echo checked
