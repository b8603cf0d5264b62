#!/bin/bash
# usage: tools/r2_pmc.sh <tag>: two SQ counter passes over a 10-step chain (one pass per run), summarised per kernel
tag=$1
o=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 -L > $o/counters_list.txt 2>&1
p1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
p2="SQ_WAVES SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS"
i=1
for p in "$p1" "$p2"; do
  timeout -s KILL 300 rocprofv3 --pmc $p --kernel-trace --output-format csv -d $o/pmc_${tag}_$i -- python bench.py --steps 1 --warmup 0 --timesteps 10 --probe-steps 10 --no-cpu-baseline --no-train > $o/pmc_${tag}_$i.log 2>&1 || { tail -20 $o/pmc_${tag}_$i.log; exit 1; }
  i=$((i+1))
done
python tools/pmc_by_kernel.py $o/pmc_${tag}.txt $o/pmc_${tag}_1 $o/pmc_${tag}_2 > /dev/null
find $o/pmc_${tag}_1 $o/pmc_${tag}_2 -name "*.csv" -size +8M -delete
echo pmc-done
