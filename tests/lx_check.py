import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import util
from oracle import allegro_torch
from pair_allegro_amd import capi, lmp_like, model_file
lib = capi.Library()
g = util.load_golden("CuPd-cubic-big_r5")
symbols = ["O" if s == "Cu" else "H" for s in g["symbols"]]
nb = float(len(util.glue.brute_force_edges(g["cell"], g["pos"], 5.0)[0])) / len(g["pos"])
for nl, U in ((1, 64), (2, 64), (3, 64), (1, 32), (2, 32), (3, 32)):
    cfg = model_file.model_L(avg_num_neighbors=nb, num_layers=nl, num_tensor_features=U)
    w = model_file.init_weights(cfg)
    path = f"/tmp/modelL_{nl}_{U}.nequip.pth"
    allegro_torch.export_nequip_pth(path, cfg, w)
    names = sorted(set(symbols))
    types = np.array([names.index(s) + 1 for s in symbols], dtype=np.int32)
    ref = util.oracle_run(dict(cfg, model_dtype="float64"), w, g["cell"], g["pos"], types, names)
    gen = util.run_pair(lib, path, g["cell"], g["pos"], types, names, options={"path": "generic"})
    try:
        res = util.run_pair(lib, path, g["cell"], g["pos"], types, names, options={"path": "fused"})
    except Exception as e:
        print("fused failed:", e); continue
    print(f"NL={nl} U={U} path={res['info']['path']} maxdeg={res['info']['max_degree']} max|dF| fused-oracle={np.abs(res['forces']-ref['forces']).max():.3e} "
          f"generic-oracle={np.abs(gen['forces']-ref['forces']).max():.3e} |F|max={np.abs(ref['forces']).max():.3f} "
          f"dPE={abs(res['pe']-ref['pe']):.3e} dEatom={np.abs(res['eatom']-ref['eatom']).max():.3e} dvir={np.abs(res['virial']-ref['virial']).max():.3e}", flush=True)

# water: a few centres with more than 64 edges -> fused kernel + layer-at-a-time kernels for those
for mm in (14,):
    cell, pos, types = lmp_like.water(mm)
    cfg = model_file.model_L(avg_num_neighbors=53.6)
    w = model_file.init_weights(cfg)
    path = "/tmp/modelL_w.ahip"
    model_file.save_ahip(path, cfg, w)
    t0 = time.time()
    gen = util.run_pair(lib, path, cell, pos, types, ["O", "H"], options={"path": "generic"})
    t1 = time.time()
    res = util.run_pair(lib, path, cell, pos, types, ["O", "H"], options={"path": "fused"})
    t2 = time.time()
    print(f"water {len(pos)} atoms: path={res['info']['path']} maxdeg={res['info']['max_degree']} max|dF| fused-generic={np.abs(res['forces']-gen['forces']).max():.3e} "
          f"dPE={abs(res['pe']-gen['pe']):.3e} dEatom={np.abs(res['eatom']-gen['eatom']).max():.3e} dvir={np.abs(res['virial']-gen['virial']).max():.3e} |F|max={np.abs(gen['forces']).max():.2f}")
