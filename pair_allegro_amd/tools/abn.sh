#!/bin/bash
# usage: abn.sh reps lib1 lib2 ...  -- alternating bench runs of several builds on one box; prints fused ms per lib
reps=$1; shift
for i in $(seq $reps); do
  for L in "$@"; do
    ALLEGRO_HIP_LIB=$PWD/$L timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', d['ms_per_step'], d['config']['stage_ms_rank0']['model_fused'])"
  done
done
