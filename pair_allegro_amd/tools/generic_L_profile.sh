#!/bin/bash
root=$(pwd); out=$root/gpurun_out/prof_generic_L; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o g -- python3 $root/pair_allegro_amd/tools/generic_L_profile.py 10 3 > $out/log.txt 2>&1
grep "model L" $out/log.txt
python3 - "$out" <<'PY'
import csv, sys, glob
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows if 'ahip' in r['Name'])
print('total ahip ms', tot / 1e6)
for r in rows[:14]:
    print(f"{r['Name'][:64]:64s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us  {float(r['Percentage']):6.2f} %")
PY
