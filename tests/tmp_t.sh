#!/bin/bash
timeout 300 python bench.py --config 5 --ncell 24 --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | grep -o '"model_fused": [0-9.]*'
timeout 300 python bench.py --config 5 --ncell 24 --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | grep -o '"model_fused": [0-9.]*'
timeout 600 python -m pytest tests/test_gpu_fused_lx.py -m gpu -x -q 2>&1 | tail -1
