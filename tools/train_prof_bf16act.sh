#!/bin/bash
# on the GPU box: per-kernel times of the training step with bf16 activation storage, and its HBM traffic
o=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/r03_prof_train_bf16act -- python tools/train_bench.py --steps 3 --warmup 1 --activations bf16 > $o/r03_prof_train_bf16act.log 2>&1 || { tail -5 $o/r03_prof_train_bf16act.log; exit 1; }
for t in $(find $o/r03_prof_train_bf16act -name "*_kernel_stats.csv"); do cp $t $o/r03_train_bf16act_kernel_stats.csv; done
find $o/r03_prof_train_bf16act -name "*_kernel_trace.csv" -size +20M -delete
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $o/r03_pmc_trainb_$c -- python tools/train_bench.py --steps 1 --warmup 1 --activations bf16 > $o/r03_pmc_trainb_$c.log 2>&1 || { tail -20 $o/r03_pmc_trainb_$c.log; exit 1; }
done
python tools/train_traffic.py $o/r03_pmc_trainb_FETCH_SIZE $o/r03_pmc_trainb_WRITE_SIZE $o/r03_train_traffic_bf16_activations.json > /dev/null || exit 1
find $o/r03_pmc_trainb_* -name "*.csv" -size +20M -delete
echo done
