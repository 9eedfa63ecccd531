#!/bin/bash
# Kernel-level profile of the generic (any-shape) path: model S on 13 824 Si atoms with --path generic.
root=$(pwd); out=$root/gpurun_out/prof_generic; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o g -- python3 $root/bench.py --ncell 12 --path generic --steps 3 --warmup 1 --no-cpu-baseline > $out/log.txt 2>&1
tail -1 $out/log.txt | cut -c1-400
python3 - "$out" <<'PY'
import csv, sys, glob
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:16]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us  {float(r['Percentage']):6.2f} %")
PY
