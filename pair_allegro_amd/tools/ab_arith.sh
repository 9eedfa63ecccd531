#!/bin/bash
# usage (GPU box, repo root): ab_arith.sh libA [libB ...]   -- same-box A/B of builds x arithmetics: 1 M-atom Si (k_fused: f32, tf32eq, bf16x3)
# and the 41 472-atom water box (k_fused_lx2), alternating; prints ms/step and the model kernel's HIP-event time
for rep in 1 2; do
  for L in "$@"; do
    for AR in f32 tf32eq b3; do
      ALLEGRO_HIP_LIB=$L AHIP_FUSED_ARITH=$AR timeout 300 python bench.py --config 4 --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 |
        python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$(basename $L)', 'config4', '$AR', d['ms_per_step'], d['config']['stage_ms_rank0'].get('model_fused'))"
    done
    ALLEGRO_HIP_LIB=$L timeout 300 python bench.py --config 5 --ncell 24 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 |
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$(basename $L)', 'water41k', 'lx2', d['ms_per_step'], d['config']['stage_ms_rank0'].get('model_fused'))"
  done
done
