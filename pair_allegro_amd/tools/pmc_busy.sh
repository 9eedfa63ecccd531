#!/bin/bash
# Two PMC passes for the "how busy are the SIMDs" question: usage pmc_busy.sh <tag> "<bench args>"  -> gpurun_out/pmc_<tag>/busy.txt
tag=${1:-x}; bargs=${2:---config 4 --ncell 30}
root=$(pwd); out=$root/gpurun_out/pmc_$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_SALU" "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out/p$i -o p -- python3 $root/bench.py $bargs --steps 2 --warmup 1 --no-cpu-baseline > $out/p$i.log 2>&1
done
cd $root
python3 - "$out" <<'PY'
import sys, glob, csv, collections
out = sys.argv[1]
agg = collections.defaultdict(list); dur = []
for f in sorted(glob.glob(out + '/p*/**/*counter_collection.csv', recursive=True)):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if 'k_fused' not in r['Kernel_Name']: continue
        per[(r['Dispatch_Id'], r['Counter_Name'])] += float(r['Counter_Value'])
    for (d, c), v in per.items(): agg[c].append(v)
for f in sorted(glob.glob(out + '/p1/**/*kernel_trace.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        if 'k_fused' in r['Kernel_Name']: dur.append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-6)
a = {c: sum(v) / len(v) for c, v in agg.items()}
with open(out + '/busy.txt', 'w') as fo:
    def P(s):
        print(s); fo.write(s + '\n')
    for c in sorted(a): P(f"{c:28s} {a[c]:.6g}")
    if dur: P(f"kernel duration ms (profiled)  {sum(dur)/len(dur):.3f}")
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in a and 'SQ_INSTS_MFMA' in a:
        nm, nv = a['SQ_INSTS_MFMA'], a['SQ_INSTS_VALU'] - a['SQ_INSTS_MFMA']
        P(f"MFMA issue cycles 32 x {nm:.4g} = {32*nm:.4g};  other VALU 4 x {nv:.4g} = {4*nv:.4g};  ratio VALU/MFMA time {4*nv/(32*nm):.3f}")
        if 'SQ_BUSY_CYCLES' in a:
            fm = a['SQ_VALU_MFMA_BUSY_CYCLES'] / 32 / a['SQ_BUSY_CYCLES']
            P(f"MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / 32 / SQ_BUSY_CYCLES = {fm:.3f};  MFMA + VALU (f32 MFMA and VALU do not overlap) = {fm * (1 + 4*nv/(32*nm)):.3f} of SIMD time")
    if 'SQ_WAVE_CYCLES' in a:
        P(f"SQ_WAIT_ANY / wave cycles {a.get('SQ_WAIT_ANY',0)/a['SQ_WAVE_CYCLES']:.3f}   SQ_WAIT_INST_ANY / wave cycles {a.get('SQ_WAIT_INST_ANY',0)/a['SQ_WAVE_CYCLES']:.3f}   active {a.get('SQ_ACTIVE_INST_ANY',0)/a['SQ_WAVE_CYCLES']:.3f}")
PY
