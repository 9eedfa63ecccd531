"""Force evaluations of model L (l_max = 2, 64 tensor features, 3 layers: BASELINE config 5's shape, generic path) on a
Si box, for `rocprofv3 --kernel-trace --stats -- python3 pair_allegro_amd/tools/generic_L_profile.py [ncell] [reps]`."""
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pair_allegro_amd import capi, lmp_like, model_file  # noqa: E402

ncell = int(sys.argv[1]) if len(sys.argv) > 1 else 10
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cfg = model_file.model_L(type_names=["Si"], avg_num_neighbors=28.0)
w = model_file.init_weights(cfg)
path = os.path.join(tempfile.mkdtemp(prefix="ahip_L_"), "modelL.ahip")
model_file.save_ahip(path, cfg, w)
lib = capi.Library()
m = capi.Model(path, 0, lib)
cell, pos, types = lmp_like.diamond_si(ncell)
rs = lmp_like.build_rank_system(cell, pos, types + 1 if types.min() == 0 else types, cfg["r_max"] + 1.0)
m.neigh_update_paged(rs.nall, rs.ilist, rs.numneigh, rs.firstneigh, 0x1FFFFFFF)
mapper = np.zeros(1, dtype=np.int32)
cm = np.full((1, 1), cfg["r_max"])
t = []
for _ in range(reps + 1):
    f = np.zeros_like(rs.x)
    t0 = time.perf_counter()
    m.compute(rs.nlocal, rs.nghost, rs.x, rs.type, mapper, cm, f, None, want_virial=True)
    t.append(time.perf_counter() - t0)
print(f"model L, {len(pos)} atoms, path {m.last_path}: {1e3 * np.mean(t[1:]):.2f} ms per evaluation (host-pointer call)")
