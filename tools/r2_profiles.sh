#!/bin/bash
# usage: tools/r2_profiles.sh <tag>: the profile set of the sampling path.
#  1. the default bench command, plain;  2. the same command under rocprofv3 --kernel-trace --stats (per-kernel and
#  per-(kernel, grid) summaries; the big per-dispatch trace is deleted);  3. FETCH_SIZE and WRITE_SIZE passes (one per
#  run) over a short chain;  4. one matrix-core counter pass.
tag=$1
o=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python bench.py > $o/${tag}_bench_default.log 2>&1 || { tail -5 $o/${tag}_bench_default.log; exit 1; }
grep -v "^[WE]2026" $o/${tag}_bench_default.log | tail -1 > $o/${tag}_bench_default.json
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $o/${tag}_prof_bench -- python bench.py > $o/${tag}_bench_under_rocprof.log 2>&1 || { tail -5 $o/${tag}_bench_under_rocprof.log; exit 1; }
grep -v "^[WE]2026" $o/${tag}_bench_under_rocprof.log | tail -1 > $o/${tag}_bench_under_rocprof.json
for t in $(find $o/${tag}_prof_bench -name "*_kernel_trace.csv"); do python tools/trace_by_grid.py $t > $o/${tag}_bench_kernel_by_grid.csv; done
for t in $(find $o/${tag}_prof_bench -name "*_kernel_stats.csv"); do cp $t $o/${tag}_bench_kernel_stats.csv; done
find $o/${tag}_prof_bench -name "*_kernel_trace.csv" -size +20M -delete
# the same with ONE launch chain: without a second chain in flight the profiler's per-kernel average, the device-clock span
# and the back-to-back time of the dominant launch (now 64 images) can be compared directly
DMN_SAMPLE_CHAINS=1 timeout -k 10 600 python bench.py --no-train --no-cpu-baseline > $o/${tag}_bench_single_chain.log 2>&1 || { tail -5 $o/${tag}_bench_single_chain.log; exit 1; }
grep -v "^[WE]2026" $o/${tag}_bench_single_chain.log | tail -1 > $o/${tag}_bench_single_chain.json
DMN_SAMPLE_CHAINS=1 timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $o/${tag}_prof_single -- python bench.py --steps 1 --warmup 1 --no-train --no-cpu-baseline > $o/${tag}_single_chain_under_rocprof.log 2>&1 || { tail -5 $o/${tag}_single_chain_under_rocprof.log; exit 1; }
for t in $(find $o/${tag}_prof_single -name "*_kernel_trace.csv"); do python tools/trace_by_grid.py $t > $o/${tag}_single_chain_kernel_by_grid.csv; python tools/step_timeline.py $t > $o/${tag}_step_timeline.txt; done
find $o/${tag}_prof_single -name "*_kernel_trace.csv" -size +20M -delete
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $o/${tag}_pmc_$c -- python bench.py --steps 1 --warmup 0 --timesteps 10 --probe-steps 10 --no-cpu-baseline --no-train > $o/${tag}_pmc_$c.log 2>&1 || { tail -5 $o/${tag}_pmc_$c.log; exit 1; }
done
f=$(find $o/${tag}_pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1)
w=$(find $o/${tag}_pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
python tools/traffic_summary.py $f $w conv_igemm_kernelIDF16bLi0ELi16ELi16ELi32ELi2ELi2ELi2ELb0E 131072 $o/${tag}_dominant_kernel_traffic.json 32 64 bf16 > /dev/null || exit 1
find $o/${tag}_pmc_FETCH_SIZE $o/${tag}_pmc_WRITE_SIZE -name "*.csv" -size +20M -delete
timeout -s KILL 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $o/${tag}_pmc_mfma -- python bench.py --steps 1 --warmup 0 --timesteps 10 --probe-steps 10 --no-cpu-baseline --no-train > $o/${tag}_pmc_mfma.log 2>&1 || { tail -5 $o/${tag}_pmc_mfma.log; exit 1; }
python tools/mfma_util.py $o/${tag}_pmc_mfma $o/${tag}_mfma_utilisation.txt > /dev/null
find $o/${tag}_pmc_mfma -name "*.csv" -size +20M -delete
echo profiled
