#!/bin/bash
# PMC passes over the fused kernel (separate runs, --kernel-trace only).
# Usage (from the repo root, on the GPU box): pair_allegro_amd/tools/pmc_passes.sh <tag> ["bench.py arguments"]   (default: --ncell 20)
# Writes gpurun_out/pmc_<tag>/summary.txt: per-dispatch averages of every counter for k_fused.
tag=${1:-x}; bargs=${2:---ncell 20}
root=$(pwd)
out=$root/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
while read -r ctrs; do
  [ -z "$ctrs" ] && continue
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out/p$i -o p -- python3 $root/bench.py $bargs --steps 2 --warmup 1 --no-cpu-baseline > $out/p$i.log 2>&1
done <<'LIST'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_SALU
SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM
FETCH_SIZE WRITE_SIZE
TCC_HIT_sum TCC_MISS_sum
LIST
cd $root
python3 - "$out" <<'PY'
import sys, glob, csv, collections
out = sys.argv[1]
agg = collections.defaultdict(list)
for f in sorted(glob.glob(out + '/p*/**/*counter_collection.csv', recursive=True)):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if 'k_fused' not in r['Kernel_Name']: continue
        per[(r['Dispatch_Id'], r['Counter_Name'])] += float(r['Counter_Value'])
    for (d, c), v in per.items(): agg[c].append(v)
with open(out + '/summary.txt', 'w') as fo:
    for c, v in sorted(agg.items()):
        line = f"{c:32s} n={len(v):3d} avg={sum(v)/len(v):.6g}"
        print(line); fo.write(line + '\n')
PY
