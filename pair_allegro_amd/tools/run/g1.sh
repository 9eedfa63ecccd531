set -x
mkdir -p gpurun_out/r06a
python -m pytest tests/test_gpu_fused.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r06a/test_fused.txt
AHIP_FUSED_NW=2 python -m pytest tests/test_gpu_fused.py -x -q -m gpu -k "golden or depth or f16x2" 2>&1 | tail -5 > gpurun_out/r06a/test_fused_nw2.txt
for rep in 1 2; do
for L in pair_allegro_amd/abl_noparkv.so pair_allegro_amd/liballegro_hip.so; do
  ALLEGRO_HIP_LIB=$PWD/$L timeout 200 python bench.py --config 4 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', d['ms_per_step'], d['config']['stage_ms_rank0']['model_fused'], d.get('parity'))"
done
AHIP_FUSED_NW=2 timeout 200 python bench.py --config 4 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('NW2', d['ms_per_step'], d['config']['stage_ms_rank0']['model_fused'], d.get('parity'))"
done > gpurun_out/r06a/ab.txt 2>&1
cat gpurun_out/r06a/*.txt
