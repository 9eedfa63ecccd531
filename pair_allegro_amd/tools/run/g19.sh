python -m pytest tests -x -q -m gpu 2>&1 | tail -4
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['dtype'], d['roofline']['bound'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline'].get('traffic_stale'), d['cpu_baseline']['value'], d['parity_vs_oracle']['max_abs_dF'])"
