#!/bin/bash
# usage: tools/train_profile.sh <tag>: training parity tests, the configuration-5 timing and a kernel trace of it.
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -u -m pytest tests/test_gpu_training.py -x -q --timeout 180 --timeout-method thread > gpurun_out/train_test_${tag}.log 2>&1 || exit 1
timeout -k 10 300 python tools/train_bench.py --steps 10 --warmup 3 > gpurun_out/train_bench_${tag}.log 2>&1 || exit 1
timeout -k 10 300 python tools/train_bench.py --steps 10 --warmup 3 --activations bf16 > gpurun_out/train_bench_${tag}_bf16act.log 2>&1 || exit 1
timeout -k 10 300 python tools/train_bench.py --steps 10 --warmup 3 --dtype fp32 > gpurun_out/train_bench_${tag}_fp32.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_train_${tag} -- python tools/train_bench.py --steps 3 --warmup 1 > gpurun_out/prof_train_${tag}.log 2>&1 || exit 1
find gpurun_out/prof_train_${tag} -name "*_kernel_trace.csv" -size +20M -delete
echo ok
