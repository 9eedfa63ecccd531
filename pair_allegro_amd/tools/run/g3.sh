mkdir -p gpurun_out/r06c
python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r06c/tests.txt
python bench.py > gpurun_out/r06c/bench_default.json 2> gpurun_out/r06c/bench_default.err
python bench.py --config 3 --steps 20 --warmup 5 > gpurun_out/r06c/bench_config3.json 2> gpurun_out/r06c/bench_config3.err
cat gpurun_out/r06c/tests.txt; tail -c 3000 gpurun_out/r06c/bench_default.json; tail -3 gpurun_out/r06c/bench_default.err
