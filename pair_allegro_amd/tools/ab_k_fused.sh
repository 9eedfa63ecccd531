#!/bin/bash
# usage (GPU box, repo root): ab_k_fused.sh lib...   -- same-box A/B of builds: 1 M-atom Si (config 4, f32) and the 10 648-atom box (config 2), alternating
for rep in 1 2 3; do for L in "$@"; do
ALLEGRO_HIP_LIB=$PWD/$L timeout 300 python bench.py --config 4 --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$(basename $L)', 'config4', d['ms_per_step'], d['config']['stage_ms_rank0'].get('model_fused'), d['roofline']['frac'])"
ALLEGRO_HIP_LIB=$PWD/$L timeout 300 python bench.py --config 2 --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$(basename $L)', 'config2', d['ms_per_step'], d['config']['stage_ms_rank0'].get('model_fused'), d['roofline']['frac'])"
done; done
