#!/bin/bash
# usage: tools/r2_train_profiles.sh <tag>: training-step profile set (configuration 5): timings in the three modes, a
# kernel trace with per-kernel stats, and the HBM traffic of one step (two counter passes) stamped with the ms/step of
# the same build.
tag=$1
o=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 python tools/train_bench.py --steps 20 --warmup 5 > $o/${tag}_train_bench_bf16.log 2>&1 || { tail -5 $o/${tag}_train_bench_bf16.log; exit 1; }
grep "^{" $o/${tag}_train_bench_bf16.log | tail -1 > $o/${tag}_train_bench_bf16.json
timeout -k 10 300 python tools/train_bench.py --steps 20 --warmup 5 --activations bf16 > $o/${tag}_train_bench_bf16act.log 2>&1 || exit 1
grep "^{" $o/${tag}_train_bench_bf16act.log | tail -1 > $o/${tag}_train_bench_bf16_activations.json
timeout -k 10 300 python tools/train_bench.py --steps 20 --warmup 5 --dtype fp32 > $o/${tag}_train_bench_fp32.log 2>&1 || exit 1
grep "^{" $o/${tag}_train_bench_fp32.log | tail -1 > $o/${tag}_train_bench_fp32.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/${tag}_prof_train -- python tools/train_bench.py --steps 3 --warmup 1 > $o/${tag}_prof_train.log 2>&1 || exit 1
for t in $(find $o/${tag}_prof_train -name "*_kernel_stats.csv"); do cp $t $o/${tag}_train_kernel_stats.csv; done
find $o/${tag}_prof_train -name "*_kernel_trace.csv" -size +20M -delete
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $o/${tag}_pmc_train_$c -- python tools/train_bench.py --steps 1 --warmup 1 > $o/${tag}_pmc_train_$c.log 2>&1 || exit 1
done
python tools/train_traffic.py $o/${tag}_pmc_train_FETCH_SIZE $o/${tag}_pmc_train_WRITE_SIZE $o/${tag}_train_traffic.json > /dev/null || exit 1
python - <<PY
import json
t = json.load(open("$o/${tag}_train_traffic.json"))
b = json.loads(open("$o/${tag}_train_bench_bf16.json").read())
t["ms_per_step_of_this_build"] = b["ms_per_step"]
t["achieved_TBs_over_the_step"] = t["total_GB"] / b["ms_per_step"]
json.dump(t, open("$o/${tag}_train_traffic.json", "w"), indent=1)
PY
find $o/${tag}_pmc_train_* -name "*.csv" -size +20M -delete
echo train-profiled
