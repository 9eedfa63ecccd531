import sys, os, tempfile
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import numpy as np
import util
from oracle import allegro_torch
from pair_allegro_amd import model_file, capi
lib = capi.Library()
g = util.load_golden("CuPd-cubic-big_r5")
symbols = ["O" if s == "Cu" else "H" for s in g["symbols"]]
nb = float(len(util.glue.brute_force_edges(g["cell"], g["pos"], 5.0)[0])) / len(g["pos"])
nl = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cfg = model_file.model_L(avg_num_neighbors=nb, num_layers=nl, num_tensor_features=64)
w = model_file.init_weights(cfg)
td = tempfile.mkdtemp()
path = f"{td}/m.nequip.pth"
allegro_torch.export_nequip_pth(path, cfg, w)
names = sorted(set(symbols))
types = np.array([names.index(s) + 1 for s in symbols], dtype=np.int32)
ref = util.oracle_run(dict(cfg, model_dtype="float64"), w, g["cell"], g["pos"], types, names)
res = util.run_pair(lib, path, g["cell"], g["pos"], types, names, options={"path": "fused"})
gen = util.run_pair(lib, path, g["cell"], g["pos"], types, names, options={"path": "generic"})
df = np.abs(res["forces"] - ref["forces"]).max(axis=1)
de = np.abs(res["eatom"] - ref["eatom"])
print("path", res["info"], "max dF", df.max(), "max dE_i", de.max(), "generic max dF", np.abs(gen["forces"]-ref["forces"]).max())
bad = np.where(df > 1e-4)[0]
print("bad force atoms", len(bad), bad[:64])
bade = np.where(de > 1e-4)[0]
print("bad energy atoms", len(bade), bade[:64])
res2 = util.run_pair(lib, path, g["cell"], g["pos"], types, names, options={"path": "fused"})
print("repeat identical:", np.array_equal(res2["forces"], res["forces"]), np.abs(res2["forces"]-res["forces"]).max())
# ---- which centres / edge slots explain the wrong forces?
ei, ej, _ = res["edges"]
order = np.argsort(ei, kind="stable")
ei, ej = ei[order], ej[order]
d = res["forces"] - ref["forces"]
badset = set(np.where(np.abs(d).max(axis=1) > 1e-4)[0].tolist())
cands = []
for c in range(len(df)):
    nb_ = ej[ei == c]
    hit = [k for k, j in enumerate(nb_) if j in badset]
    if c in badset and len(hit) >= 3:
        cands.append((c, len(nb_), hit))
print("candidate centres (centre, degree, bad neighbour slots):")
for c in cands[:12]:
    print(c)
print("second run max dF vs ref", np.abs(res2["forces"] - ref["forces"]).max())
for k in range(3):
    r3 = util.run_pair(lib, path, g["cell"], g["pos"], types, names, options={"path": "fused"})
    print("run", k + 3, "max dF vs ref", np.abs(r3["forces"] - ref["forces"]).max())
# ---- the same model object evaluated repeatedly (scratch, weights and TLB warm after the first call)
from pair_allegro_amd import lmp_like
from pair_allegro_amd.pair import PairAllegro, atom_from_rank_system, list_from_rank_system
pair = PairAllegro(me=0, nprocs=1, lib=lib, quiet=True)
pair.settings([]); pair.coeff(["*", "*", path] + list(names), ntypes=len(names))
pair.model.set_option("path", "fused"); pair.init_style()
rs = lmp_like.build_rank_system(g["cell"], g["pos"], types, pair.init_one(1, 1) + 1.0)
lst = list_from_rank_system(rs)
for k in range(5):
    atom = atom_from_rank_system(rs, len(names))
    pair.compute(atom, lst)
    f = np.zeros((len(g["pos"]), 3)); np.add.at(f, rs.tag - 1, atom.f)
    print("same model call", k, "max dF vs ref", np.abs(f - ref["forces"]).max())
