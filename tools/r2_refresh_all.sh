set -o pipefail
o=gpurun_out
bash tools/r2_profiles.sh r02 || exit 1
timeout -k 10 300 python bench.py --learned-variance --no-train --no-cpu-baseline > $o/r02_bench_improved_ddpm.log 2>&1 || { tail -5 $o/r02_bench_improved_ddpm.log; exit 1; }
grep -v "^[WE]2026" $o/r02_bench_improved_ddpm.log | tail -1 > $o/r02_bench_improved_ddpm.json
timeout -k 10 200 python tools/tail_phases.py 32 > $o/r02_tail_phases.txt 2>&1 || { tail -5 $o/r02_tail_phases.txt; exit 1; }
timeout -k 10 300 python tools/handoff_stress.py > $o/r02_handoff_stress.txt 2>&1 || { tail -5 $o/r02_handoff_stress.txt; exit 1; }
for m in 0 128 0 128; do
  DMN_UNET_PATHS=$m timeout -k 10 200 python bench.py --no-train --no-cpu-baseline --steps 2 > $o/r02_ab_m${m}.log 2>&1 || { tail -5 $o/r02_ab_m${m}.log; exit 1; }
  echo "DMN_UNET_PATHS=$m: $(grep -v '^[WE]2026' $o/r02_ab_m${m}.log | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], "img/s", d["step"]["device_ms"], "ms/step")')" >> $o/r02_ab_exchange_new.txt
done
cat $o/r02_ab_exchange_new.txt
cat $o/r02_bench_default.json
