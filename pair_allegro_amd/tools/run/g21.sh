python -m pytest tests/test_gpu_parity.py -q -m gpu --tb=short 2>&1 | grep -E "passed|failed"
for L in abl_tp3old.so liballegro_hip.so; do
  ALLEGRO_HIP_LIB=$PWD/pair_allegro_amd/$L timeout 300 python bench.py --config 2 --l-max 3 --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', d['config']['kernel_path'], d['ms_per_step'], d['value'], d['config']['stage_ms_rank0'])"
done
