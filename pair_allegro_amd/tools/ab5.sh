#!/bin/bash
# usage: ab5.sh lib1 lib2 ... (paths relative to repo root); prints model_fused ms for config 5, 41k atoms
for L in "$@"; do
  ALLEGRO_HIP_LIB=$PWD/$L timeout 200 python bench.py --config 5 --ncell 24 --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', d['config']['stage_ms_rank0']['model_fused'])"
done
