mkdir -p gpurun_out/r06v
python bench.py > gpurun_out/r06v/bench_default_final.json 2> gpurun_out/r06v/bench_default_final.err
tail -1 gpurun_out/r06v/bench_default_final.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], d['value_with_amortised_rebuilds'], r['frac'], r['avg_ms'], r['traffic'], r.get('traffic_stale'), r.get('vector_memory_path'), d['cpu_baseline']['value'])"
