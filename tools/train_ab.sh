#!/bin/bash
# usage: tools/train_ab.sh VAR v1 v2 ...: the configuration-5 training timing once per value of an environment variable
var=$1; shift
for v in "$@"; do
  echo "== $var=$v"
  env $var=$v timeout -k 10 120 python tools/train_bench.py --steps 30 --warmup 5 || exit 1
done
