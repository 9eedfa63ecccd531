"""Where does a re-neighboring go?  usage (GPU box, repo root): python pair_allegro_amd/tools/rebuild_profile.py [ncell]
Times the parts of md.Simulation.rebuild() (single rank) with a device synchronisation after each."""
import os, sys, time, tempfile
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from pair_allegro_amd import capi, lmp_like, md, model_file

ncell = int(sys.argv[1]) if len(sys.argv) > 1 else 11
cell, pos, types = lmp_like.diamond_si(ncell)
cfg = model_file.model_S()
path = os.path.join(tempfile.mkdtemp(), "m.ahip")
model_file.save_ahip(path, cfg, model_file.init_weights(cfg))
model = capi.Model(path, 0, capi.Library())
dev = torch.device("cuda", 0)
vel = md.maxwell_boltzmann(len(pos), np.full(len(pos), 28.0855), 300.0, 1)
sim = md.Simulation(md.HipBackend(model, [28.0855]), np.diag(cell), cfg["r_max"], 1.0, pos, np.zeros(len(pos), np.int32), vel, dev, overlap=False)
sim.setup()
for _ in range(5):
    sim.step()
acc = {}
def timed(name, fn):
    torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize()
    acc[name] = acc.get(name, 0.0) + (time.perf_counter() - t)
N = 20
for _ in range(N):
    timed("migrate", sim._migrate)
    timed("borders", sim._borders)
    timed("set_comm_plan", sim._set_comm_plan)
    def rest():
        sim.f = torch.zeros((sim.nall, 3), dtype=torch.float64, device=dev)
        lo = sim.lo - sim.rc - 1e-6; hi = sim.hi + sim.rc + 1e-6
        sim.backend.build_neighbors(sim.x, sim.nlocal, lo, hi, sim.rc)
        sim.x_hold = sim.x[: sim.nlocal].clone()
    timed("neighbor_build", rest)
    timed("whole_rebuild", sim.rebuild)
print({k: round(1e3 * v / N, 3) for k, v in acc.items()}, "ms per call;", len(pos), "atoms")
