"""Round-end bookkeeping: copies the artefacts of two tools/final_profile.sh runs (config 4, config 5) from gpurun_out/ into profiles/ and rewrites
profiles/traffic.json from their FETCH_SIZE / WRITE_SIZE averages, stamped with the kernel-source hash of THIS tree (bench.py flags a mismatch as
traffic_stale).  usage (repo root): python pair_allegro_amd/tools/install_profile.py <tag of the config-4 run> <tag of the config-5 run> <profiles prefix, e.g. r04_h>"""
import json, os, re, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench

t4, t5, prefix = sys.argv[1:4]
h = bench.kernel_source_hash()
tj = os.path.join(ROOT, "profiles", "traffic.json")
t = json.load(open(tj))


def averages(tag):
    out = {}
    for line in open(os.path.join(ROOT, "gpurun_out", f"final_{tag}.log")):
        m = re.match(r"^(k_fused|k_build_edges) (FETCH_SIZE|WRITE_SIZE) dispatches \d+ avg ([0-9.eE+]+)", line)
        if m:
            out[(m.group(1), m.group(2))] = float(m.group(3))
    return out


for tag, cfg, natoms, label in ((t4, 4, 1000000, "config4_1M_Si"), (t5, 5, 499125, "config5_500k_water")):
    a = averages(tag)
    line = open(os.path.join(ROOT, "gpurun_out", f"prof_{tag}", "bench_default.json")).read().strip().splitlines()[-1]
    path = json.loads(line)["config"]["kernel_path"]          # the entry is keyed by the kernel path the run actually took (fused_f16x2 since round 5)
    for kern, key in (("k_fused", f"config{cfg}:{path}:{natoms}"), ("k_build_edges", f"config{cfg}:k_build_edges:{natoms}")):
        if key not in t:
            t[key] = dict(t[f"config{cfg}:fused_f32:{natoms}"], kernel=f"{'k_fused' if cfg == 4 else 'k_fused_lx2'} on {path}")
        e = t[key]
        e["source"] = f"profiles/{prefix}_final.md (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, tools/final_profile.sh on the final tree)"
        e["fetch_size_kb"], e["write_size_kb"] = a[(kern, "FETCH_SIZE")], a[(kern, "WRITE_SIZE")]
        e["traffic_bytes_per_launch"] = (e["fetch_size_kb"] * e["fetch_correction"] + e["write_size_kb"]) * 1024.0
        e["kernel_hash"] = h
        e["measured_on"] = f"final tree, run {tag}"
        print(f"{key}: {e['traffic_bytes_per_launch'] / 1e9:.3f} GB")
    shutil.copy(os.path.join(ROOT, "gpurun_out", f"prof_{tag}", "trace", "t_kernel_stats.csv"), os.path.join(ROOT, "profiles", f"{prefix}_kernel_stats_{label}.csv"))
    open(os.path.join(ROOT, "profiles", f"{prefix}_bench_config{cfg}.json"), "w").write(line + "\n")
json.dump(t, open(tj, "w"), indent=1)
print("kernel hash", h)
