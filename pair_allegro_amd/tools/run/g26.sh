# Last evidence stamp of round 6 (tree with the l_max = 3 tensor-product change in gemm.hip; the fused kernels' device code is unchanged since r06_d): traffic + kernel stats of configs 4, 5 and 6,
# TD / TA counters of k_fused, the default line, smoke, the whole GPU suite.
mkdir -p gpurun_out/r06v
export AHIP_NO_ARITH_SELFCHECK=1
bash pair_allegro_amd/tools/final_profile.sh r06_4 "" > gpurun_out/final_r06_4.log 2>&1
bash pair_allegro_amd/tools/final_profile.sh r06_5 "--config 5" > gpurun_out/final_r06_5.log 2>&1
bash pair_allegro_amd/tools/traffic_passes.sh r06_6 "--config 6" > gpurun_out/traffic_r06_6.log 2>&1
bash pair_allegro_amd/tools/pmc_mem.sh r06_k_fused_216k "--config 4 --ncell 30" > gpurun_out/r06v/pmc_mem.log 2>&1
unset AHIP_NO_ARITH_SELFCHECK
python bench.py > gpurun_out/r06v/bench_default.json 2> gpurun_out/r06v/bench_default.err
python bench.py --config 6 > gpurun_out/r06v/bench_config6.json 2> gpurun_out/r06v/bench_config6.err
python bench.py --config 2 --l-max 3 --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/r06v/bench_config2_l3.json 2> gpurun_out/r06v/bench_config2_l3.err
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06v/smoke.txt 2>&1
python -m pytest tests -q -m gpu --tb=line 2>&1 | tail -12 > gpurun_out/r06v/suite.txt
grep -E "^k_" gpurun_out/final_r06_4.log gpurun_out/final_r06_5.log gpurun_out/traffic_r06_6.log; grep -E "BUSY_sum /" gpurun_out/pmc_r06_k_fused_216k/mem.txt; tail -1 gpurun_out/r06v/smoke.txt; grep -E "passed|failed" gpurun_out/r06v/suite.txt
for f in bench_default bench_config6 bench_config2_l3; do tail -1 gpurun_out/r06v/$f.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$f', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_ms'])"; done
