#!/bin/bash
# PMC passes for "what is the vector-memory path doing": texture-addresser / texture-data busy, buffer wavefronts, L1 stalls, LDS.
# usage (GPU box, repo root): pmc_mem.sh <tag> "<bench args>"  -> gpurun_out/pmc_<tag>/mem.txt   (counters only: no --stats / sys-trace beside --pmc)
tag=${1:-x}; bargs=${2:---config 4 --ncell 30}
root=$(pwd); out=$root/gpurun_out/pmc_$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TD_TD_BUSY_sum TA_BUSY_avr TA_BUSY_max" \
            "TA_BUFFER_TOTAL_CYCLES_sum TA_BUFFER_READ_WAVEFRONTS_sum TA_BUFFER_WRITE_WAVEFRONTS_sum TD_LOAD_WAVEFRONT_sum TD_TC_STALL_sum" \
            "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
            "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_CYCLES_VMEM_RD"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out/m$i -o p -- python3 $root/bench.py $bargs --steps 2 --warmup 1 --no-cpu-baseline > $out/m$i.log 2>&1
done
cd $root
python3 - "$out" <<'PY'
import sys, glob, csv, collections
out = sys.argv[1]
agg = collections.defaultdict(list); dur = []
for f in sorted(glob.glob(out + '/m*/**/*counter_collection.csv', recursive=True)):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if 'k_fused' not in r['Kernel_Name']: continue
        per[(r['Dispatch_Id'], r['Counter_Name'])] += float(r['Counter_Value'])
    for (d, c), v in per.items(): agg[c].append(v)
for f in sorted(glob.glob(out + '/m1/**/*kernel_trace.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        if 'k_fused' in r['Kernel_Name']: dur.append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-6)
a = {c: sum(v) / len(v) for c, v in agg.items()}
with open(out + '/mem.txt', 'w') as fo:
    def P(s):
        print(s); fo.write(s + '\n')
    for c in sorted(a): P(f"{c:36s} {a[c]:.6g}")
    if dur: P(f"kernel duration ms (profiled)  {sum(dur)/len(dur):.3f}")
    g = a.get('GRBM_GUI_ACTIVE')
    if g:
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs; the *_sum counters over all 256 CUs' instances
        cyc = g / 8.0
        P(f"cycles per XCD {cyc:.4g}")
        for c in ('TA_TA_BUSY_sum', 'TD_TD_BUSY_sum'):
            if c in a: P(f"{c} / (256 CUs x cycles) = {a[c] / (256 * cyc):.3f}")
    if 'TA_BUFFER_TOTAL_CYCLES_sum' in a and g:
        P(f"TA_BUFFER_TOTAL_CYCLES_sum / (256 x cycles) = {a['TA_BUFFER_TOTAL_CYCLES_sum'] / (256 * g / 8.0):.3f}")
    for c in ('TCP_TCP_TA_DATA_STALL_CYCLES_sum', 'TCP_TD_TCP_STALL_CYCLES_sum', 'TCP_PENDING_STALL_CYCLES_sum'):
        if c in a and 'TCP_GATE_EN1_sum' in a: P(f"{c} / TCP_GATE_EN1_sum = {a[c] / a['TCP_GATE_EN1_sum']:.3f}")
    if 'SQ_WAVE_CYCLES' in a:
        for c in ('SQ_ACTIVE_INST_VMEM', 'SQ_ACTIVE_INST_LDS', 'SQ_WAIT_INST_LDS', 'SQ_INST_LEVEL_VMEM'):
            if c in a: P(f"{c} / SQ_WAVE_CYCLES = {a[c] / a['SQ_WAVE_CYCLES']:.3f}")
    if 'SQ_LDS_IDX_ACTIVE' in a and a['SQ_LDS_IDX_ACTIVE']: P(f"SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = {a.get('SQ_LDS_BANK_CONFLICT', 0) / a['SQ_LDS_IDX_ACTIVE']:.3f}")
PY
