#!/bin/bash
timeout 1200 python -m pytest tests/test_gpu_md.py -m gpu -x -q 2>&1 | tail -1
for a in "--config 2" "--config 4 --ncell 25" "--config 4"; do timeout 300 python bench.py $a --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['stage_ms_rank0'], d['value'], d['config']['rebuilds'])"; done
