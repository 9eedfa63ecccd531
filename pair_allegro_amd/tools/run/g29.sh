mkdir -p gpurun_out/r06w
python bench.py --config 2 --steps 1000 --warmup 100 > gpurun_out/r06w/bench_config2.json 2> gpurun_out/r06w/bench_config2.err
python bench.py --config 3 --steps 100 --warmup 10 > gpurun_out/r06w/bench_config3.json 2> gpurun_out/r06w/bench_config3.err
python bench.py --config 2 --path generic --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/r06w/bench_config2_generic.json 2> gpurun_out/r06w/bench_config2_generic.err
for f in bench_config2 bench_config3 bench_config2_generic; do tail -1 gpurun_out/r06w/$f.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); p=d.get('parity_vs_oracle') or {}; print('$f', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_ms'], d['config'].get('rebuilds_in_timed_steps'), p.get('max_abs_dF'), p.get('max_abs_dF_f32_instance'))"; done
