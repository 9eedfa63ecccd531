#!/bin/bash
# scratch: instruction-cache counters for k_fused (config 4) and k_fused_lx (config 5)
root=$(pwd); out=$root/gpurun_out/pmc_icache; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for cfg in "4 --ncell 20" "5 --ncell 24"; do
for ctrs in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQC_TC_INST_REQ"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out/p$i -o p -- python3 $root/bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline > $out/p$i.log 2>&1
done; done
cd $root
python3 - "$out" <<'PY'
import sys, glob, csv, collections
out = sys.argv[1]
agg = collections.defaultdict(list)
for f in sorted(glob.glob(out + '/p*/**/*counter_collection.csv', recursive=True)):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'k_fused' not in k: continue
        kn = 'k_fused_lx' if 'k_fused_lx' in k else 'k_fused'
        per[(kn, r['Dispatch_Id'], r['Counter_Name'])] += float(r['Counter_Value'])
    for (kn, d, c), v in per.items(): agg[(kn, c)].append(v)
with open(out + '/summary.txt', 'w') as fo:
    for c, v in sorted(agg.items()):
        line = f"{c[0]:12s} {c[1]:32s} n={len(v):3d} avg={sum(v)/len(v):.6g}"
        print(line); fo.write(line + '\n')
PY
