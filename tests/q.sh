python tests/lx_check.py 2>&1 | grep -v Warning | tail -2 | cut -c1-200
AHIP_FUSED_PROF=1 python bench.py --config 5 --ncell 30 --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -2 | cut -c1-520
python bench.py --config 5 --ncell 30 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['config']['stage_ms_rank0'])"
