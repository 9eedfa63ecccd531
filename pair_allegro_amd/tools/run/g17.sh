python -m pytest tests/test_gpu_fused.py -q -m gpu --tb=short -k "device_resident_path" 2>&1 | tail -30
