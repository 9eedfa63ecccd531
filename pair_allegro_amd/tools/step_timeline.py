"""Per-step kernel timeline from a rocprofv3 rocpd database (kernel trace of `bench.py --config 2`): kernels and idle gaps between two consecutive
k_fused launches, averaged over the steady steps (steps with a re-neighboring are left out), plus one step in full.
usage: python pair_allegro_amd/tools/step_timeline.py <results.db>"""
import sqlite3
import statistics
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
short = lambda n: n.split("(")[0].replace("void ", "").replace("ahip::", "")[:56]
fused = [i for i, r in enumerate(rows) if "k_fused" in r[0]]
per = []
for a, b in zip(fused[20:-1], fused[21:]):
    seg = rows[a:b]
    if any("neigh" in r[0] for r in seg):
        continue
    span = (rows[b][1] - rows[a][1]) / 1e3
    busy = sum(r[2] - r[1] for r in seg) / 1e3
    per.append((span, busy, len(seg)))
print(f"steady steps: {len(per)};  per step: span {statistics.mean(p[0] for p in per):.1f} us, kernels busy {statistics.mean(p[1] for p in per):.1f} us, "
      f"{statistics.mean(p[2] for p in per):.1f} kernels, idle between kernels {statistics.mean(p[0] - p[1] for p in per):.1f} us")
a, b = fused[30], fused[31]
t0 = rows[a][1]
print("\none step (start us, duration us, kernel):")
for r in rows[a:b + 1]:
    print(f"{(r[1] - t0) / 1e3:9.1f} {(r[2] - r[1]) / 1e3:8.1f}  {short(r[0])}")
