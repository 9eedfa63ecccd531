mkdir -p gpurun_out/r06g
ALLEGRO_HIP_LIB=$PWD/pair_allegro_amd/abl_tbre.so python -m pytest tests/test_gpu_fused.py -q -m gpu -x --tb=short -k "golden or two_types or many_species or f16x2_arithmetic" 2>&1 | tail -4 > gpurun_out/r06g/tests_tbre.txt
run() { ALLEGRO_HIP_LIB=$PWD/pair_allegro_amd/$1 timeout 200 python bench.py --config 4 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', '$2', d['ms_per_step'], d['config']['stage_ms_rank0']['model_fused'])"; }
for rep in 1 2; do
  run liballegro_hip.so base; run abl_tbre.so tbre
  AHIP_TCHUNK=2 run liballegro_hip.so tchunk2; AHIP_TCHUNK=8 run liballegro_hip.so tchunk8
done > gpurun_out/r06g/ab.txt 2>&1
cat gpurun_out/r06g/tests_tbre.txt gpurun_out/r06g/ab.txt
