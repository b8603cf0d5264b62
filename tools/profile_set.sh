#!/bin/bash
# usage: tools/profile_set.sh <tag> <part> [<part> ...]      (on the GPU box; results under gpurun_out/<tag>_*)
# The measurement set behind profiles/<tag>_*, one part per kind of evidence:
#   check     GPU tests, then the default bench command
#   sampling  default bench plain and under rocprofv3 --kernel-trace --stats (per-kernel and per-(kernel, grid) summaries,
#             the per-dispatch trace is deleted), the same with ONE launch chain plus its step timeline, FETCH_SIZE and
#             WRITE_SIZE passes (one counter per run) over a short chain, one matrix-core counter pass
#   counters  two SQ counter passes over a 10-step chain, summarised per kernel
#   training  configuration-5 timings in the three modes, kernel stats, HBM traffic of one step (two counter passes), the
#             last two also with bf16 activation storage
#   extras    improved-DDPM bench line, exchange stress, A/B of the in-launch exchanges against separate closing passes
#   bygrid    a 300-step two-chain bench under the kernel trace, summarised per (kernel, grid); extra bench arguments
#             after "--"
tag=$1; shift
o=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
last_json() { grep -v "^[WE]2026" "$1" | tail -1; }
run_log() {   # run_log <log> <seconds> <command...>: run under a time limit, show the log's tail on failure
  log=$1; lim=$2; shift 2
  timeout -k 10 $lim "$@" > $log 2>&1 || { tail -5 $log; exit 1; }
}
pmc_run() {   # pmc_run <dir> <counters...> -- <program...>: one counter pass, kernel trace only
  dir=$1; shift
  ctr=(); while [ "$1" != "--" ]; do ctr+=("$1"); shift; done; shift
  timeout -s KILL 300 rocprofv3 --pmc "${ctr[@]}" --kernel-trace --output-format csv -d $dir -- "$@" > $dir.log 2>&1 || { tail -20 $dir.log; exit 1; }
}
short="python bench.py --steps 1 --warmup 0 --timesteps 10 --probe-steps 10 --no-cpu-baseline --no-train"
while [ $# -gt 0 ]; do
  part=$1; shift
  case $part in
  check)
    timeout -k 10 900 python -u -m pytest tests -x -q -m gpu --timeout 300 --timeout-method thread > $o/${tag}_gpu_tests.log 2>&1
    rc=$?; tail -4 $o/${tag}_gpu_tests.log; [ $rc -ne 0 ] && exit $rc
    run_log $o/${tag}_bench_check.log 600 python bench.py
    last_json $o/${tag}_bench_check.log | cut -c 1-400
    ;;
  sampling)
    run_log $o/${tag}_bench_default.log 600 python bench.py
    last_json $o/${tag}_bench_default.log > $o/${tag}_bench_default.json
    run_log $o/${tag}_bench_under_rocprof.log 900 rocprofv3 --kernel-trace --stats --output-format csv -d $o/${tag}_prof_bench -- python bench.py
    last_json $o/${tag}_bench_under_rocprof.log > $o/${tag}_bench_under_rocprof.json
    for t in $(find $o/${tag}_prof_bench -name "*_kernel_trace.csv"); do python tools/trace_by_grid.py $t > $o/${tag}_bench_kernel_by_grid.csv; done
    for t in $(find $o/${tag}_prof_bench -name "*_kernel_stats.csv"); do cp $t $o/${tag}_bench_kernel_stats.csv; done
    find $o/${tag}_prof_bench -name "*_kernel_trace.csv" -size +20M -delete
    # ONE launch chain: without a second chain in flight the profiler's per-kernel average, the device-clock span and the
    # back-to-back time of the dominant launch (now 64 images) can be compared directly
    DMN_SAMPLE_CHAINS=1 run_log $o/${tag}_bench_single_chain.log 600 python bench.py --no-train --no-cpu-baseline
    last_json $o/${tag}_bench_single_chain.log > $o/${tag}_bench_single_chain.json
    DMN_SAMPLE_CHAINS=1 run_log $o/${tag}_single_chain_under_rocprof.log 900 rocprofv3 --kernel-trace --stats --output-format csv -d $o/${tag}_prof_single -- python bench.py --steps 1 --warmup 1 --no-train --no-cpu-baseline
    for t in $(find $o/${tag}_prof_single -name "*_kernel_trace.csv"); do
      python tools/trace_by_grid.py $t > $o/${tag}_single_chain_kernel_by_grid.csv
      python tools/step_timeline.py $t > $o/${tag}_step_timeline.txt
    done
    find $o/${tag}_prof_single -name "*_kernel_trace.csv" -size +20M -delete
    for c in FETCH_SIZE WRITE_SIZE; do pmc_run $o/${tag}_pmc_$c $c -- $short; done
    f=$(find $o/${tag}_pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1)
    w=$(find $o/${tag}_pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
    python tools/traffic_summary.py $f $w conv_igemm_kernelIDF16bLi0ELi32ELi16ELi32ELi4ELi1ELi2ELb0ELb1E 65536 $o/${tag}_dominant_kernel_traffic.json 32 64 bf16 > /dev/null || exit 1
    find $o/${tag}_pmc_FETCH_SIZE $o/${tag}_pmc_WRITE_SIZE -name "*.csv" -size +20M -delete
    pmc_run $o/${tag}_pmc_mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE -- $short
    python tools/mfma_util.py $o/${tag}_pmc_mfma $o/${tag}_mfma_utilisation.txt > /dev/null
    find $o/${tag}_pmc_mfma -name "*.csv" -size +20M -delete
    ;;
  counters)
    pmc_run $o/pmc_${tag}_1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -- $short
    pmc_run $o/pmc_${tag}_2 SQ_WAVES SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS -- $short
    python tools/pmc_by_kernel.py $o/pmc_${tag}.txt $o/pmc_${tag}_1 $o/pmc_${tag}_2 > /dev/null
    find $o/pmc_${tag}_1 $o/pmc_${tag}_2 -name "*.csv" -size +8M -delete
    ;;
  training)
    run_log $o/${tag}_train_bench_bf16.log 300 python tools/train_bench.py --steps 20 --warmup 5
    grep "^{" $o/${tag}_train_bench_bf16.log | tail -1 > $o/${tag}_train_bench_bf16.json
    run_log $o/${tag}_train_bench_bf16act.log 300 python tools/train_bench.py --steps 20 --warmup 5 --activations bf16
    grep "^{" $o/${tag}_train_bench_bf16act.log | tail -1 > $o/${tag}_train_bench_bf16_activations.json
    run_log $o/${tag}_train_bench_fp32.log 300 python tools/train_bench.py --steps 20 --warmup 5 --dtype fp32
    grep "^{" $o/${tag}_train_bench_fp32.log | tail -1 > $o/${tag}_train_bench_fp32.json
    run_log $o/${tag}_prof_train.log 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/${tag}_prof_train -- python tools/train_bench.py --steps 3 --warmup 1
    for t in $(find $o/${tag}_prof_train -name "*_kernel_stats.csv"); do cp $t $o/${tag}_train_kernel_stats.csv; done
    find $o/${tag}_prof_train -name "*_kernel_trace.csv" -size +20M -delete
    for c in FETCH_SIZE WRITE_SIZE; do pmc_run $o/${tag}_pmc_train_$c $c -- python tools/train_bench.py --steps 1 --warmup 1; done
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
    # the same with the saved activations stored in bf16: per-kernel times and traffic
    run_log $o/${tag}_prof_train_bf16act.log 300 rocprofv3 --kernel-trace --stats --output-format csv -d $o/${tag}_prof_train_bf16act -- python tools/train_bench.py --steps 3 --warmup 1 --activations bf16
    for t in $(find $o/${tag}_prof_train_bf16act -name "*_kernel_stats.csv"); do cp $t $o/${tag}_train_bf16_activations_kernel_stats.csv; done
    find $o/${tag}_prof_train_bf16act -name "*_kernel_trace.csv" -size +20M -delete
    for c in FETCH_SIZE WRITE_SIZE; do pmc_run $o/${tag}_pmc_trainb_$c $c -- python tools/train_bench.py --steps 1 --warmup 1 --activations bf16; done
    python tools/train_traffic.py $o/${tag}_pmc_trainb_FETCH_SIZE $o/${tag}_pmc_trainb_WRITE_SIZE $o/${tag}_train_traffic_bf16_activations.json > /dev/null || exit 1
    python - <<PY
import json
t = json.load(open("$o/${tag}_train_traffic_bf16_activations.json"))
b = json.loads(open("$o/${tag}_train_bench_bf16_activations.json").read())
t["ms_per_step_of_this_build"] = b["ms_per_step"]
t["achieved_TBs_over_the_step"] = t["total_GB"] / b["ms_per_step"]
json.dump(t, open("$o/${tag}_train_traffic_bf16_activations.json", "w"), indent=1)
PY
    find $o/${tag}_pmc_trainb_* -name "*.csv" -size +20M -delete
    ;;
  extras)
    run_log $o/${tag}_bench_improved_ddpm.log 300 python bench.py --learned-variance --no-train --no-cpu-baseline
    last_json $o/${tag}_bench_improved_ddpm.log > $o/${tag}_bench_improved_ddpm.json
    run_log $o/${tag}_tail_phases.txt 200 python tools/tail_phases.py 32
    run_log $o/${tag}_handoff_stress.txt 300 python tools/handoff_stress.py
    : > $o/${tag}_ab_exchange.txt
    for m in 0 128 0 128; do
      DMN_UNET_PATHS=$m run_log $o/${tag}_ab_m${m}.log 200 python bench.py --no-train --no-cpu-baseline --steps 2
      echo "DMN_UNET_PATHS=$m: $(last_json $o/${tag}_ab_m${m}.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], "img/s", d["step"]["device_ms"], "ms/step")')" >> $o/${tag}_ab_exchange.txt
    done
    cat $o/${tag}_ab_exchange.txt
    ;;
  bygrid)
    extra=(); if [ "$1" = "--" ]; then shift; extra=("$@"); set --; fi
    run_log $o/${tag}_under_rocprof.log 600 rocprofv3 --kernel-trace --stats --output-format csv -d $o/${tag}_prof -- python bench.py --steps 1 --warmup 1 --timesteps 300 --no-train --no-cpu-baseline "${extra[@]}"
    for t in $(find $o/${tag}_prof -name "*_kernel_trace.csv"); do python tools/trace_by_grid.py $t > $o/${tag}_by_grid.csv; done
    find $o/${tag}_prof -name "*_kernel_trace.csv" -size +20M -delete
    head -60 $o/${tag}_by_grid.csv | cut -c 1-170
    ;;
  *) echo "unknown part $part"; exit 2;;
  esac
done
echo "profile_set $tag done"
