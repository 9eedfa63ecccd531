"""Per-edge view of an arithmetic of k_fused (AHIP_FUSED_DBG=1 dump: g[3], dd, dfc, dY[3] per edge).
usage (GPU box, repo root): AHIP_FUSED_DBG=1 [ALLEGRO_HIP_LIB=...] python pair_allegro_amd/tools/dbg_arith_edges.py arith out.npy
Writes the [E][8] dump of the 256-atom CuPd box, two-layer model S, plus the edge list; compare two dumps with numpy."""
import os, sys, tempfile, ctypes as C
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import util
from pair_allegro_amd import model_file, capi, lmp_like
from pair_allegro_amd.pair import PairAllegro, atom_from_rank_system, list_from_rank_system

arith, out = sys.argv[1], sys.argv[2]
lib = capi.Library()
g = util.load_golden("CuPd-cubic-big_r5")
names = ["Cu", "Pd"]
types = np.array([names.index(s) + 1 for s in g["symbols"]], dtype=np.int32)
nb = float(len(util.glue.brute_force_edges(g["cell"], g["pos"], 5.0)[0])) / len(g["pos"])
cfg = model_file.model_S(type_names=names, num_layers=2, seed=7, avg_num_neighbors=nb)
w = model_file.init_weights(cfg)
path = os.path.join(tempfile.mkdtemp(), "m.ahip")
model_file.save_ahip(path, cfg, w)
pair = PairAllegro(me=0, nprocs=1, lib=lib, quiet=True)
pair.settings([]); pair.coeff(["*", "*", path] + names, ntypes=len(names))
pair.model.set_option("path", "fused"); pair.model.set_option("fused_arith", arith); pair.init_style()
rs = lmp_like.build_rank_system(g["cell"], g["pos"], types, pair.init_one(1, 1) + 1.0)
lst = list_from_rank_system(rs)
dumps = []
for k in range(3):
    atom = atom_from_rank_system(rs, len(names))
    pair.compute(atom, lst)
    E = pair.model.nedges()
    buf = np.zeros((E, 8), dtype=np.float32)
    rc = lib.lib.ahip_debug_fused_edges(pair.model.h, buf.ctypes.data_as(C.POINTER(C.c_float)), C.c_longlong(E))
    assert rc == 0, rc
    dumps.append(buf)
ei, r = pair.model.get_edges()
np.save(out, dict(dumps=np.stack(dumps), ei=ei, path=pair.model.last_path), allow_pickle=True)
d = np.stack(dumps)
print("path", pair.model.last_path, "edges", E, "run-to-run max diff per column", np.abs(d[1] - d[0]).max(axis=0), np.abs(d[2] - d[0]).max(axis=0))
