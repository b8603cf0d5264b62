o=gpurun_out/train_sweep.txt; : > $o
run() { env $1 timeout -k 10 120 python tools/train_bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d.get('ms_per_step'))" >> $o || exit 1; }
run X=base; run DMN_LINATTN_WGS=1024; run DMN_LINATTN_WGS=4096; run DMN_PGEMM_SMALL=1024; run DMN_PGEMM_SMALL=256; run DMN_PGEMM_C32_TILES=4; run DMN_PGEMM_C32_TILES=16; run DMN_WGEMM_WGS=192; run DMN_WGEMM_WGS=320; run X=base2
cat $o
