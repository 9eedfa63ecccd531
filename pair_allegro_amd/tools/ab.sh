#!/bin/bash
# usage: ab.sh libA libB [reps]   -- alternating bench runs of two builds on one box (same-box A/B)
A=$1; B=$2; reps=${3:-2}
for i in $(seq $reps); do
  for L in $A $B; do
    ALLEGRO_HIP_LIB=$L AHIP_FUSED_CLK=1 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > /tmp/ab.out 2> /tmp/ab.err
    tail -1 /tmp/ab.out | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$(basename $L)', d['ms_per_step'], d['config']['stage_ms_rank0'])"
    grep "fused clk" /tmp/ab.err | tail -2
  done
done
