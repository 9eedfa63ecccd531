"""Where do two arithmetics of k_fused differ?  usage (GPU box, repo root): python pair_allegro_amd/tools/dbg_arith_case.py [arith]
Runs the 256-atom CuPd box on the f32 path and on `arith` (tf32eq | bf16x3) with the library named by ALLEGRO_HIP_LIB and prints the
differences of per-atom energies (forward pass only), total energy, forces and virial (forward + backward)."""
import os, sys, tempfile
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import util
from pair_allegro_amd import model_file, capi

arith = sys.argv[1] if len(sys.argv) > 1 else "tf32eq"
lib = capi.Library()
g = util.load_golden("CuPd-cubic-big_r5")
names = ["Cu", "Pd"]
types = np.array([names.index(s) + 1 for s in g["symbols"]], dtype=np.int32)
nb = float(len(util.glue.brute_force_edges(g["cell"], g["pos"], 5.0)[0])) / len(g["pos"])
for nl in (1, 2, 3):
    cfg = model_file.model_S(type_names=names, num_layers=nl, seed=7, avg_num_neighbors=nb)
    w = model_file.init_weights(cfg)
    path = os.path.join(tempfile.mkdtemp(), "m.ahip")
    model_file.save_ahip(path, cfg, w)
    a = util.run_pair(lib, path, g["cell"], g["pos"], types, names, options={"path": "fused", "fused_arith": "f32"})
    for rep in range(3):
        b = util.run_pair(lib, path, g["cell"], g["pos"], types, names, options={"path": "fused", "fused_arith": arith})
        de = np.abs(b["eatom"] - a["eatom"])
        df = np.abs(b["forces"] - a["forces"]).max(axis=1)
        print(f"layers {nl} run {rep}: path {b['info']['path']}  max|dE_i| {de.max():.3e} (atoms > 1e-3: {(de > 1e-3).sum()})  max|dF| {df.max():.3e} (atoms > 1e-2: {(df > 1e-2).sum()})  |F|max {np.abs(a['forces']).max():.3f}")
