#!/bin/bash
# usage (GPU box, repo root): ab_all.sh lib...   -- same-box A/B of builds on config 4 (1 M Si), config 5 / 6 on the 41 472-atom water box, alternating
for rep in 1 2; do for L in "$@"; do
ALLEGRO_HIP_LIB=$PWD/$L timeout 300 python bench.py --config 4 --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$(basename $L)', 'config4', d['ms_per_step'], d['config']['stage_ms_rank0'].get('model_fused'), d['roofline']['frac'])"
for C in 5 6; do
ALLEGRO_HIP_LIB=$PWD/$L timeout 300 python bench.py --config $C --ncell 24 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$(basename $L)', 'water41k config$C', d['ms_per_step'], d['config']['stage_ms_rank0'].get('model_fused'), d['roofline']['frac'])"
done; done; done
