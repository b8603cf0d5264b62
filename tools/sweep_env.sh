#!/bin/bash
# usage: tools/sweep_env.sh <out-file> "<ENV=val ...>" ...   -- one step_time.py run per environment setting
out=$1; shift
: > $out
timeout -k 10 120 python tools/step_time.py --tag base >> $out 2>&1 || exit 1
for e in "$@"; do
  env $e timeout -k 10 120 python tools/step_time.py --tag "$e" >> $out 2>&1 || { echo "FAILED $e" >> $out; exit 1; }
done
timeout -k 10 120 python tools/step_time.py --tag base-again >> $out 2>&1
grep STEP_US $out
