# Final evidence stamp of round 6 (the tree after the hygiene edits; kernels' device code unchanged since r06_c): traffic + kernel stats of configs 4 and 5, TD / TA counters, the default line, smoke, the whole GPU suite.
mkdir -p gpurun_out/r06u
export AHIP_NO_ARITH_SELFCHECK=1
bash pair_allegro_amd/tools/final_profile.sh r06_4 "" > gpurun_out/final_r06_4.log 2>&1
bash pair_allegro_amd/tools/final_profile.sh r06_5 "--config 5" > gpurun_out/final_r06_5.log 2>&1
bash pair_allegro_amd/tools/pmc_mem.sh r06_k_fused_216k "--config 4 --ncell 30" > gpurun_out/r06u/pmc_mem.log 2>&1
unset AHIP_NO_ARITH_SELFCHECK
python bench.py > gpurun_out/r06u/bench_default.json 2> gpurun_out/r06u/bench_default.err
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06u/smoke.txt 2>&1
python -m pytest tests -q -m gpu --tb=line 2>&1 | tail -12 > gpurun_out/r06u/suite.txt
grep -E "^k_" gpurun_out/final_r06_4.log gpurun_out/final_r06_5.log; grep -E "BUSY_sum /" gpurun_out/pmc_r06_k_fused_216k/mem.txt; tail -1 gpurun_out/r06u/smoke.txt; grep -E "passed|failed" gpurun_out/r06u/suite.txt
tail -1 gpurun_out/r06u/bench_default.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_ms'])"
