export AHIP_NO_ARITH_SELFCHECK=1
for NWV in 4 7 8 7; do
  AHIP_FUSED_NW=$NWV timeout 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('NW=$NWV', d['config']['kernel_path'], d['ms_per_step'], d['value'], d['config']['stage_ms_rank0'], d['config']['pe_per_atom'], d['roofline'].get('slot_occupancy'))"
done
