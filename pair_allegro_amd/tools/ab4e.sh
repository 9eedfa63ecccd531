#!/bin/bash
# usage: ab4e.sh lib1 lib2 ...: model_fused ms of BASELINE config 4 (1 M Si) per library at FIXED positions (bench.py --eval-only): for timing builds whose
# forces are wrong by construction (ABL_NOW ...), which an NVE run would carry out of the box
for L in "$@"; do
  ALLEGRO_HIP_LIB=$PWD/$L timeout 200 python bench.py --config 4 --steps 4 --warmup 2 --no-cpu-baseline --eval-only 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', d['config']['stage_ms_rank0']['model_fused'])"
done
