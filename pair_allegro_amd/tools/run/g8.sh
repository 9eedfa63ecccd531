# Round-6 evidence stamp, ONE call: kernel trace + FETCH / WRITE passes of configs 4 and 5, TD / TA counters of k_fused, bench lines of configs 2 / 3 / 6 and the generic path, the default line, the whole GPU suite.
mkdir -p gpurun_out/r06t
export AHIP_NO_ARITH_SELFCHECK=1          # counter passes aggregate dispatches by kernel-name substring: no f32-instance dispatch of the first-evaluation self-check among them
bash pair_allegro_amd/tools/final_profile.sh r06_4 "" > gpurun_out/final_r06_4.log 2>&1
bash pair_allegro_amd/tools/final_profile.sh r06_5 "--config 5" > gpurun_out/final_r06_5.log 2>&1
bash pair_allegro_amd/tools/pmc_mem.sh r06_k_fused_216k "--config 4 --ncell 30" > gpurun_out/r06t/pmc_mem.log 2>&1
unset AHIP_NO_ARITH_SELFCHECK
python bench.py --config 2 --steps 1000 --warmup 100 > gpurun_out/r06t/bench_config2.json 2> gpurun_out/r06t/bench_config2.err
python bench.py --config 3 --steps 100 --warmup 10 > gpurun_out/r06t/bench_config3.json 2> gpurun_out/r06t/bench_config3.err
python bench.py --config 6 > gpurun_out/r06t/bench_config6.json 2> gpurun_out/r06t/bench_config6.err
python bench.py --config 2 --path generic --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/r06t/bench_config2_generic.json 2> gpurun_out/r06t/bench_config2_generic.err
python bench.py > gpurun_out/r06t/bench_default.json 2> gpurun_out/r06t/bench_default.err
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06t/smoke.txt 2>&1
python -m pytest tests -q -m gpu --tb=line 2>&1 | tail -12 > gpurun_out/r06t/suite.txt
cat gpurun_out/final_r06_4.log | tail -8; cat gpurun_out/final_r06_5.log | tail -8; cat gpurun_out/pmc_r06_k_fused_216k/mem.txt; tail -3 gpurun_out/r06t/smoke.txt; cat gpurun_out/r06t/suite.txt | tail -4
for f in bench_config2 bench_config3 bench_config6 bench_default; do tail -1 gpurun_out/r06t/$f.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$f', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_ms'], d.get('parity_vs_oracle',{}) and {k: d['parity_vs_oracle'][k] for k in ('max_abs_dF','max_abs_dF_f32_instance')})"; done
