mkdir -p gpurun_out/r06x
python bench.py --config 5 > gpurun_out/r06x/bench_config5.json 2> gpurun_out/r06x/bench_config5.err
python bench.py --config 6 > gpurun_out/r06x/bench_config6.json 2> gpurun_out/r06x/bench_config6.err
for f in bench_config5 bench_config6; do tail -1 gpurun_out/r06x/$f.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$f', d['value'], d['ms_per_step'], r['frac'], r['avg_ms'], r['traffic'], r.get('traffic_stale'), (r.get('vector_memory_path') or {}).get('stale'))"; done
