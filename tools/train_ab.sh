#!/bin/bash
# usage: tools/train_ab.sh <variant> [<rounds> [<train_bench arguments>]]   (on the GPU box): training step time of lib/libdmn_hip_<variant>.so and of the
# default library, alternating, <rounds> times each (default 3)
v=$1; n=${2:-3}; shift; shift
for i in $(seq $n); do
  for t in $v -; do
    if [ "$t" = "-" ]; then unset DMN_LIB_VARIANT; else export DMN_LIB_VARIANT=$t; fi
    ms=$(timeout -k 10 120 python tools/train_bench.py --steps 30 --warmup 5 "$@" 2>/dev/null | grep "^{" | tail -1 | python -c 'import sys,json; print(json.loads(sys.stdin.read())["ms_per_step"])') || exit 1
    echo "TRAIN_MS $t $ms"
  done
done
