mkdir -p gpurun_out/r06y
for n in 2 8; do
AHIP_BENCH_ONE_DEVICE=1 timeout 900 python bench.py --gpus $n --steps 5 --warmup 2 > gpurun_out/r06y/bench_one_device_$n.json 2> gpurun_out/r06y/bench_one_device_$n.err
echo "rc $?"; tail -1 gpurun_out/r06y/bench_one_device_$n.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print(d['n_gpus'], d['value'], d['ms_per_step'], c['grid'], c['comm'], c['comm_transport'], c['comm_autotune'], c['rebuild_ms'], c.get('rebuild_cadence_measured'), d['value_with_amortised_rebuilds'])"
grep -E "rank [0-9]+/" gpurun_out/r06y/bench_one_device_$n.err | head -3
done
