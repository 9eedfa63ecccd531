mkdir -p gpurun_out/r06i
python -m pytest tests/test_gpu_fused.py tests/test_gpu_fused_lx.py -q -m gpu --tb=short 2>&1 | tail -25 > gpurun_out/r06i/tests.txt
cat gpurun_out/r06i/tests.txt
