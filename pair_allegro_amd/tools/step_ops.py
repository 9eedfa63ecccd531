"""Which torch operators still run inside one MD step?  (python pair_allegro_amd/tools/step_ops.py)"""
import os, sys, tempfile
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
from pair_allegro_amd import capi, md, model_file
wl = bench.workload(2, 6)
cfg = wl["cfg"]; w = model_file.init_weights(cfg)
path = os.path.join(tempfile.mkdtemp(), "m.ahip"); model_file.save_ahip(path, cfg, w)
lib = capi.Library(); model = capi.Model(path, 0, lib)
dev = torch.device("cuda", 0)
vel = md.maxwell_boltzmann(len(wl["pos"]), np.full(len(wl["pos"]), 28.0855), 300.0, 1)
sim = md.Simulation(md.HipBackend(model, wl["masses"]), np.diag(wl["cell"]), cfg["r_max"], 1.0, wl["pos"], wl["mtype"], vel, dev)
sim.setup()
for _ in range(3): sim.step()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    for _ in range(2): sim.step()
for e in prof.key_averages(group_by_stack_n=6):
    if e.key.startswith("aten::") and e.count >= 2:
        print(e.count, e.key, " | ".join(s for s in e.stack[:4] if "md.py" in s or "capi.py" in s or "bench" in s))
