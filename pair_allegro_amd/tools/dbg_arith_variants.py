"""Fault bisection helper (GPU box): runs the 256-atom CuPd box, 2-layer model S, on every library pair_allegro_amd/var/liballegro_hip_*.so
(assembly-level variants of the bf16-split k_fused instances, built by tools/asm_variant.sh) and prints how many atoms have wrong forces.
usage: python pair_allegro_amd/tools/dbg_arith_variants.py [reps]   (one subprocess per library)"""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import os, sys, tempfile
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import util
from pair_allegro_amd import model_file, capi
reps = int(sys.argv[1]); layers = [int(a) for a in sys.argv[2].split(",")]
lib = capi.Library()
g = util.load_golden("CuPd-cubic-big_r5")
names = ["Cu", "Pd"]
types = np.array([names.index(s) + 1 for s in g["symbols"]], dtype=np.int32)
nb = float(len(util.glue.brute_force_edges(g["cell"], g["pos"], 5.0)[0])) / len(g["pos"])
for nl in layers:
    cfg = model_file.model_S(type_names=names, num_layers=nl, seed=7, avg_num_neighbors=nb)
    w = model_file.init_weights(cfg)
    path = os.path.join(tempfile.mkdtemp(), "m.ahip")
    model_file.save_ahip(path, cfg, w)
    a = util.run_pair(lib, path, g["cell"], g["pos"], types, names, options={"path": "fused", "fused_arith": "f32"})
    for arith in ("tf32eq", "bf16x3"):
        bad = []; mx = []
        for rep in range(reps):
            b = util.run_pair(lib, path, g["cell"], g["pos"], types, names, options={"path": "fused", "fused_arith": arith})
            df = np.abs(b["forces"] - a["forces"]).max(axis=1)
            bad.append(int((df > 1e-2).sum())); mx.append(float(df.max()))
        print(f"  layers {nl} {arith:7s}: bad atoms per run {bad}  max|dF| {max(mx):.3e}", flush=True)
'''
reps = sys.argv[1] if len(sys.argv) > 1 else "6"
layers = sys.argv[2] if len(sys.argv) > 2 else "2"
libs = sorted(glob.glob(os.path.join(ROOT, "pair_allegro_amd", "var", "liballegro_hip_*.so")))
libs = [os.path.join(ROOT, "pair_allegro_amd", "liballegro_hip.so")] + libs
for lib in libs:
    print(os.path.basename(lib), flush=True)
    env = dict(os.environ, ALLEGRO_HIP_LIB=lib)
    r = subprocess.run([sys.executable, "-c", CHILD, reps, layers], env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = r.stdout.decode()
    print(out if r.returncode == 0 else out[-1500:], flush=True)
