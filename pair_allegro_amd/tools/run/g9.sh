mkdir -p gpurun_out/r06h
ALLEGRO_HIP_LIB=$PWD/pair_allegro_amd/abl_envpad.so python -m pytest tests/test_gpu_fused_lx.py -q -m gpu -x --tb=short 2>&1 | tail -3 > gpurun_out/r06h/tests_envpad.txt
run() { ALLEGRO_HIP_LIB=$PWD/pair_allegro_amd/$1 timeout 300 python bench.py --config $2 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', 'cfg$2', d['ms_per_step'], d['config']['stage_ms_rank0']['model_fused'])"; }
for rep in 1 2; do
  run liballegro_hip.so 5; run abl_envpad.so 5; run liballegro_hip.so 6; run abl_envpad.so 6
done > gpurun_out/r06h/ab.txt 2>&1
cat gpurun_out/r06h/tests_envpad.txt gpurun_out/r06h/ab.txt
