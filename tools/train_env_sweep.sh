#!/bin/bash
# usage: tools/train_env_sweep.sh <ENV_NAME> <value> [<value> ...] [-- <train_bench arguments>]   (on the GPU box)
# training step time for each value of one environment variable ("-" = unset), two passes
name=$1; shift
vals=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do vals+=("$1"); shift; done; [ "$1" = "--" ] && shift
for pass in 1 2; do
  for v in "${vals[@]}"; do
    if [ "$v" = "-" ]; then unset $name; else export $name=$v; fi
    ms=$(timeout -k 10 120 python tools/train_bench.py --steps 30 --warmup 5 "$@" 2>/dev/null | grep "^{" | tail -1 | python -c 'import sys,json; print(json.loads(sys.stdin.read())["ms_per_step"])') || exit 1
    echo "TRAIN_MS $name=$v $* $ms"
  done
done
