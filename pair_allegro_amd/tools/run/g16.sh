mkdir -p gpurun_out/r06x
python -m pytest tests/test_gpu_fused.py tests/test_gpu_soak.py tests/test_gpu_equivariance.py -q -m gpu --tb=short 2>&1 | tail -4 > gpurun_out/r06x/tests.txt
run() { ALLEGRO_HIP_LIB=$PWD/pair_allegro_amd/$1 timeout 300 python bench.py --config $2 --steps $3 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', 'cfg$2', d['ms_per_step'], d['config']['stage_ms_rank0']['model_fused'])"; }
for rep in 1 2; do
  run abl_red0.so 4 5; run liballegro_hip.so 4 5
done > gpurun_out/r06x/ab.txt 2>&1
run abl_red0.so 3 20 >> gpurun_out/r06x/ab.txt; run liballegro_hip.so 3 20 >> gpurun_out/r06x/ab.txt
run abl_red0.so 2 300 >> gpurun_out/r06x/ab.txt; run liballegro_hip.so 2 300 >> gpurun_out/r06x/ab.txt
cat gpurun_out/r06x/tests.txt gpurun_out/r06x/ab.txt
