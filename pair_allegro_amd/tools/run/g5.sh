mkdir -p gpurun_out/r06e
python -m pytest tests/test_gpu_fused.py tests/test_gpu_parity.py tests/test_gpu_soak.py -q -m gpu --tb=short 2>&1 | tail -30 > gpurun_out/r06e/tests.txt
for rep in 1 2; do
for L in abl_c2.so liballegro_hip.so; do
  ALLEGRO_HIP_LIB=$PWD/pair_allegro_amd/$L timeout 200 python bench.py --config 4 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', 'cfg4', d['ms_per_step'], d['config']['stage_ms_rank0']['model_fused'])"
done
done > gpurun_out/r06e/ab.txt 2>&1
python bench.py --config 2 --path generic --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/r06e/bench_config2_generic.json 2> gpurun_out/r06e/bench_config2_generic.err
cat gpurun_out/r06e/tests.txt gpurun_out/r06e/ab.txt; tail -c 1500 gpurun_out/r06e/bench_config2_generic.json
