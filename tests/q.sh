#!/bin/bash
# scratch
AHIP_EDGES_ONLY=1 timeout 200 python bench.py --config 4 --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | grep -o '"edge_build": [0-9.]*'
AHIP_EDGES_ONLY=1 timeout 200 python bench.py --config 4 --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | grep -o '"edge_build": [0-9.]*'
AHIP_EDGES_ONLY=1 timeout 200 python bench.py --config 5 --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | grep -o '"edge_build": [0-9.]*'
AHIP_EDGES_ONLY=1 timeout 200 python bench.py --config 3 --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | grep -o '"edge_build": [0-9.]*'
timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py tests/test_gpu_fused_lx.py -m gpu -x -q 2>&1 | tail -3
