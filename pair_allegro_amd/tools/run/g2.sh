mkdir -p gpurun_out/r06b
python -m pytest tests/test_gpu_fused.py tests/test_gpu_soak.py tests/test_gpu_equivariance.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r06b/tests.txt
for rep in 1 2; do
for L in abl_c1.so abl_nostg.so liballegro_hip.so; do
  ALLEGRO_HIP_LIB=$PWD/pair_allegro_amd/$L timeout 200 python bench.py --config 4 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', 'cfg4', d['ms_per_step'], d['config']['stage_ms_rank0']['model_fused'])"
done
done > gpurun_out/r06b/ab.txt 2>&1
for L in abl_c1.so liballegro_hip.so; do
  ALLEGRO_HIP_LIB=$PWD/pair_allegro_amd/$L timeout 200 python bench.py --config 3 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', 'cfg3', d['ms_per_step'], d['config']['stage_ms_rank0']['model_fused'])"
  ALLEGRO_HIP_LIB=$PWD/pair_allegro_amd/$L timeout 200 python bench.py --config 2 --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', 'cfg2', d['ms_per_step'], d['config']['stage_ms_rank0']['model_fused'])"
done >> gpurun_out/r06b/ab.txt 2>&1
cat gpurun_out/r06b/*.txt
