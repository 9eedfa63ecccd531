#!/bin/bash
# Round-end evidence run (from the repo root on the GPU box): kernel trace + stats of the default bench, HBM-side
# traffic of the fused kernel (separate --pmc passes), the plain bench line, the micro-benchmarks.
# Usage: pair_allegro_amd/tools/final_profile.sh <tag> ["bench.py arguments"]     (default: the default bench = config 4)
tag=${1:-r01f}; bargs=${2:-}
root=$(pwd)
out=$root/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python3 $root/bench.py $bargs --steps 5 --warmup 2 --no-cpu-baseline > $out/trace.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_$c -o p -- python3 $root/bench.py $bargs --steps 2 --warmup 1 --no-cpu-baseline > $out/pmc_$c.log 2>&1
done
cd $root
python3 - "$out" <<'PY'
import sys, glob, csv, collections
out = sys.argv[1]
for kern in ("k_fused", "k_build_edges"):
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        per = collections.defaultdict(float)
        for f in glob.glob(out + f'/pmc_{c}/**/*counter_collection.csv', recursive=True):
            for r in csv.DictReader(open(f)):
                if kern in r['Kernel_Name'] and r['Counter_Name'] == c:
                    per[r['Dispatch_Id']] += float(r['Counter_Value'])
        v = list(per.values())
        print(kern, c, 'dispatches', len(v), 'avg', sum(v) / max(len(v), 1))
PY
tail -1 $out/trace.log | cut -c1-300
python bench.py $bargs > $out/bench_default.json 2> $out/bench_default.err; tail -1 $out/bench_default.json
find $out -name "*kernel_stats.csv" | head -2
