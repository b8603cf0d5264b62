#!/bin/bash
# usage: tools/mfma_util.sh <tag>: matrix-core busy cycles per kernel of the sampling step (one counter pass over a short
# bench.py run, whose measured_peaks part also launches the bare MFMA loops that calibrate "100 % busy").
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc_mfma_${tag}
timeout -s KILL 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out -- python bench.py --steps 1 --warmup 0 --timesteps 10 --no-cpu-baseline --no-train > $out.log 2>&1 || { tail -20 $out.log; exit 1; }
python tools/mfma_util.py $out gpurun_out/mfma_util_${tag}.txt
find $out -name "*.csv" -size +20M -delete
echo ok
