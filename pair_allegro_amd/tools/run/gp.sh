#!/bin/bash
# usage: gp.sh <timeout-seconds> <script under pair_allegro_amd/tools/run/>   -- gpurun with retries while no slot is free
t=$1; sc=$2
for i in $(seq 1 30); do
  out=$(gpurun --timeout $t -- "bash pair_allegro_amd/tools/run/$sc" 2>&1); rc=$?
  if echo "$out" | grep -q "status=transient"; then sleep 90; continue; fi
  echo "$out" | tail -60; exit $rc
done
echo "no slot after 30 tries"
