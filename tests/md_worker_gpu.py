"""Worker of tests/test_gpu_md.py::test_multi_rank_on_one_gpu (one process per rank, all on cuda:0, messages staged through
gloo by md.HostStagedDist): the real HIP kernels under the real multi-rank schedule."""
import os
import sys

os.environ.setdefault("GPU_PINNED_MIN_XFER_SIZE", "4095")   # harness default, before torch initialises HIP (capi.harness_pinned_copy_default)
import numpy as np
import torch
import torch.distributed as dist

from pair_allegro_amd import capi, lmp_like, md, model_file


def run(lib, path, cell, pos, vel, cfg, grid, rank, d, nsteps, overlap):
    dev = torch.device("cuda", 0)
    model = capi.Model(path, 0, lib)
    model.set_option("edge_schedule", "dynamic")      # several processes share this GPU: the resident-grid assumption of the static unit schedule does not hold
    sim = md.Simulation(md.HipBackend(model, [28.0855]), np.diag(cell), cfg["r_max"], 1.0, pos, np.zeros(len(pos), np.int32),
                        vel, dev, grid=grid, rank=rank, dist=d, dt=0.001, overlap=overlap)
    sim.setup()
    f = sim.gather_forces()
    for _ in range(nsteps):
        sim.step()
    x = torch.zeros((len(pos), 3), dtype=torch.float64, device=dev)
    x[sim.tag[: sim.nlocal]] = sim.x[: sim.nlocal]
    if d is not None and sim.nranks > 1:
        d.all_reduce(x)
    th = sim.thermo([28.0855])
    used = model.last_path
    nloc, nint = sim.nlocal, getattr(sim, "n_int", 0)
    torch.cuda.synchronize()
    model.close()
    return f, x.cpu().numpy(), th["pe"], np.array(th["virial"]), used, nloc, nint


def main():
    out, model_dir = sys.argv[1:3]
    nsteps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    staged = md.HostStagedDist(dist)
    grid = md.choose_grid(world)
    lib = capi.Library()
    cfg = model_file.model_S()                                   # the bench model: float32, runs on the fused kernel
    w = model_file.init_weights(cfg)
    path = os.path.join(model_dir, f"md_gpu_r{rank}.ahip")
    model_file.save_ahip(path, cfg, w)
    cell, pos, _ = lmp_like.diamond_si(5)                        # 27.2 A box: bricks of 13.6 A > r_max + skin
    vel = md.maxwell_boltzmann(len(pos), np.full(len(pos), 28.0855), 600.0, 4321)
    f2, x2, e2, v2, used, nloc, nint = run(lib, path, cell, pos, vel, cfg, grid, rank, staged, nsteps, None)   # overlapped schedule (default for > 1 rank)
    nl = torch.tensor([nloc]); dist.all_reduce(nl)
    assert int(nl.item()) == len(pos), "atoms lost or duplicated"
    f3, x3, e3, v3, _, _, _ = run(lib, path, cell, pos, vel, cfg, grid, rank, staged, nsteps, False)            # serial exchange
    if rank == 0:
        f1, x1, e1, v1, used1, _, _ = run(lib, path, cell, pos, vel, cfg, (1, 1, 1), 0, None, nsteps, False)
        box = np.diag(cell)
        wrap = lambda x: x - np.floor(x / box) * box
        np.savez(out, f1=f1, f2=f2, f3=f3, x1=wrap(x1), x2=wrap(x2), x3=wrap(x3), e1=e1, e2=e2, e3=e3, v1=v1, v2=v2, v3=v3,
                 used=used, used1=used1, nint=nint, nloc=nloc)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
