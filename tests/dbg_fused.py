import sys, os, numpy as np, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["AHIP_FUSED_DBG"] = "1"
import util, tempfile
from oracle import allegro_torch, glue
from pair_allegro_amd import capi, lmp_like
from pair_allegro_amd.pair import PairAllegro, atom_from_rank_system, list_from_rank_system
lib = capi.Library()
g = util.load_golden("Si64_r5")
d = tempfile.mkdtemp()
path, cfg, w = util.golden_model(g, d, "float32")
types, names = util.lammps_types(g)
rs = lmp_like.build_rank_system(g["cell"], g["pos"], types, 6.0)
pair = PairAllegro(lib=lib, quiet=True); pair.coeff(["*", "*", path, "Si"], ntypes=1); pair.model.set_option("path", "fused")
atom = atom_from_rank_system(rs, 1)
pair.compute(atom, list_from_rank_system(rs))
E = pair.model.get_edges()[0].shape[1]
out = np.zeros((E, 8), dtype=np.float32)
lib.lib.ahip_debug_fused_edges.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_longlong]
rc = lib.lib.ahip_debug_fused_edges(pair.model.h, out.ctypes.data_as(C.POINTER(C.c_float)), E)
print("rc", rc, "E", E); np.save(os.environ.get("DBG_OUT","/tmp/dbg.npy"), out)
# oracle per-edge gradient
cfg64 = dict(cfg, model_dtype="float64")
m = allegro_torch.build(cfg64, w)
inp = glue.preprocess(rs.x, rs.type, rs.nlocal, rs.ilist, rs.numneigh, rs.firstneigh, np.array([0]), np.array([[5.0]]))
pos = torch.from_numpy(inp["pos"]); ei = torch.from_numpy(inp["edge_index"]); ty = torch.from_numpy(inp["atom_types"])
rvec = (pos[ei[1]] - pos[ei[0]]).detach().requires_grad_(True)
eps = m.edge_energy(rvec, ty[ei[0]], ty[ei[1]], ei[0], pos.shape[0])
esum = torch.zeros(pos.shape[0], dtype=eps.dtype).index_add(0, ei[0], eps)
ea = m.scale[ty] * (esum * m.inv_sqrt_nn) + m.shift[ty]
gref = torch.autograd.grad([ea.sum()], [rvec])[0].numpy()
err = np.abs(out[:, :3] - gref).max(axis=1)
print("max |dg|", err.max(), " bad edges:", int((err > 1e-4).sum()), "of", E)
bad = np.flatnonzero(err > 1e-4)
# slot within tile: Si 28 edges/atom, 4 atoms per tile
slot = bad % 112
print("bad slot histogram by wave (slot//32):", np.bincount(slot // 32, minlength=4))
print("bad slots:", np.unique(slot)[:60])
print("tiles with bad edges:", np.unique(bad // 112))
i = bad[:5]
print("examples fused g:", out[i, :3], "\nref g:", gref[i])
