mkdir -p gpurun_out/r06w
timeout 900 python pair_allegro_amd/tools/nve_drift.py --steps 5000 --every 100 --out gpurun_out/r06w/nve_drift.txt > gpurun_out/r06w/nve.log 2>&1
AHIP_SOAK_LAUNCHES=1000 timeout 1500 python -m pytest tests/test_gpu_soak.py -q -m gpu 2>&1 | tail -3 > gpurun_out/r06w/soak_1000.txt
tail -12 gpurun_out/r06w/nve_drift.txt; cat gpurun_out/r06w/soak_1000.txt
