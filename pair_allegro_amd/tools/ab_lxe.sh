#!/bin/bash
# usage (GPU box, repo root): ab_lxe.sh <config 5|6> lib...   -- same-box A/B of builds on the 41 472-atom water box at FIXED positions (bench.py --eval-only: timing-experiment
# builds compute garbage and an NVE run would carry the atoms away): config 5 (model L, k_fused_lx2) or config 6 (U = 32, k_fused_lx)
C=$1; shift
for L in "$@"; do
ALLEGRO_HIP_LIB=$PWD/$L timeout 300 python bench.py --config $C --ncell 24 --steps 10 --warmup 2 --no-cpu-baseline --eval-only 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$(basename $L)', 'water41k config$C', d['ms_per_step'], d['config']['stage_ms_rank0'].get('model_fused'))"
done
