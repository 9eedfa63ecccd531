mkdir -p gpurun_out/r06f
python -m pytest tests/test_gpu_fused_lx.py tests/test_gpu_soak.py tests/test_gpu_equivariance.py tests/test_gpu_configs.py -q -m gpu --tb=short 2>&1 | tail -15 > gpurun_out/r06f/tests.txt
for rep in 1 2; do
for L in abl_c3.so liballegro_hip.so; do
  ALLEGRO_HIP_LIB=$PWD/pair_allegro_amd/$L timeout 300 python bench.py --config 6 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', 'cfg6', d['ms_per_step'], d['config']['stage_ms_rank0']['model_fused'])"
done
done > gpurun_out/r06f/ab.txt 2>&1
timeout 300 python bench.py --config 5 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('tree', 'cfg5', d['ms_per_step'], d['config']['stage_ms_rank0']['model_fused'])" >> gpurun_out/r06f/ab.txt
cat gpurun_out/r06f/tests.txt gpurun_out/r06f/ab.txt
