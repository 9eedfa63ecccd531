#!/bin/bash
# usage: abprof.sh libA libB  -- phase profile + effective clock of both builds on one box
for L in "$@"; do
  echo "== $L"
  ALLEGRO_HIP_LIB=$L AHIP_FUSED_PROF=1 timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | grep "fused prof" | tail -2
done
