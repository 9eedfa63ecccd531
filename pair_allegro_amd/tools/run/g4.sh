mkdir -p gpurun_out/r06d
python -m pytest tests/test_gpu_fused.py -q -m gpu -k "auto_is_never or per_centre_type" --tb=short 2>&1 | tail -60 > gpurun_out/r06d/t1.txt
python -m pytest tests -q -m gpu --tb=line --deselect tests/test_gpu_fused.py::test_auto_is_never_less_robust_than_float32 2>&1 | tail -25 > gpurun_out/r06d/tests.txt
cat gpurun_out/r06d/t1.txt gpurun_out/r06d/tests.txt
