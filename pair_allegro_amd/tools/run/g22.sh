root=$(pwd); mkdir -p gpurun_out/r06l3
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/r06l3/trace -o t -- python3 $root/bench.py --config 2 --l-max 3 --steps 20 --warmup 3 --no-cpu-baseline > $root/gpurun_out/r06l3/trace.log 2>&1
cd $root
f=$(find gpurun_out/r06l3 -name "*kernel_stats.csv" | head -1); head -22 $f | cut -c1-150
