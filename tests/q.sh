timeout 120 python -m pytest tests/test_gpu_configs.py tests/test_gpu_md.py -q -m gpu -k "edge_index or neighbor_list" 2>&1 | tail -2
export AHIP_EDGES_ONLY=1
timeout 200 python - <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from pair_allegro_amd import capi, lmp_like, md, model_file
cfg = model_file.model_S(); w = model_file.init_weights(cfg)
model_file.save_ahip('/tmp/s.ahip', cfg, w)
cell, pos, _ = lmp_like.diamond_si(50)
for name, libp in (("new", None),):
    lib = capi.Library(libp)
    m = capi.Model('/tmp/s.ahip', 0, lib)
    m.set_option("timing", "1")
    dev = torch.device("cuda", 0)
    sim = md.Simulation(md.HipBackend(m, [28.0855]), np.diag(cell), 5.0, 1.0, pos, np.zeros(len(pos), np.int32), None, dev, overlap=False)
    ts = []
    for _ in range(8):
        sim.backend.compute(sim.x, sim.mtype, sim.f, sim.nlocal, sim.engvir)
        torch.cuda.synchronize()
        ts.append(m.timings().get("edge_build"))
    print(name, m.last_path, np.round(ts[2:], 4))
    m.close()
PY
