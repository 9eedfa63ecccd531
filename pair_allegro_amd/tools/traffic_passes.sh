#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes only (the byte counts of profiles/traffic.json): usage traffic_passes.sh <tag> ["bench.py arguments"]
tag=${1:-x}; bargs=${2:-}
root=$(pwd); out=$root/gpurun_out/traffic_$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
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
