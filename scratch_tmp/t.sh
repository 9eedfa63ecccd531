#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | grep -E "^E|FAILED|Error" | head -20
