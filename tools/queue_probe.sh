#!/bin/bash
# Experiment: does a third / fourth launch chain pay once the process owns more hardware queues?
# Independent single-chain handles side by side (eager launches, closing passes as their own launches so that no
# in-launch exchange can starve), with 4 (default) and 8 hardware queues.
set -e
out=gpurun_out/r3_queue_probe.txt
: > $out
for q in 4 8; do
  for split in 64 32,32 22,21,21 16,16,16,16; do
    echo "== GPU_MAX_HW_QUEUES=$q split=$split" >> $out
    GPU_MAX_HW_QUEUES=$q DMN_UNET_PATHS=128 timeout -k 10 150 python tools/multi_chain_probe.py $split 300 >> $out 2>&1
  done
done
cat $out
