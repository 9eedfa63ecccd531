"""Worker of tests/test_md_emu.py::test_world_size_2_gloo_matches_single_rank (one process per rank)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

from pair_allegro_amd import capi, lmp_like, md, model_file


def run(lib, path, cell, pos, vel, cfg, grid, rank, d, nsteps, skin=1.0, toggle=False):
    model = capi.Model(path, 0, lib)
    sim = md.Simulation(md.HipBackend(model, [28.0855]), np.diag(cell), cfg["r_max"], skin, pos, np.zeros(len(pos), np.int32),
                        vel, torch.device("cpu"), grid=grid, rank=rank, dist=d, dt=0.001)
    sim.setup()
    f = sim.gather_forces()
    for k in range(nsteps):
        if toggle and sim.nranks > 1 and k % 3 == 1:
            sim.set_overlap(not sim.overlap)         # bench.py's schedule choice: switching between the two schedules must not change the trajectory
        sim.step()
    x = torch.zeros((len(pos), 3), dtype=torch.float64)
    x[sim.tag[: sim.nlocal]] = sim.x[: sim.nlocal]
    if d is not None and sim.nranks > 1:
        d.all_reduce(x)
    e = sim.thermo([28.0855])["pe"]
    nreb = sim.nrebuild
    nloc = sim.nlocal
    run.lib_plan = bool(getattr(sim, "_lib_plan", False))          # the last re-neighboring went through ahip_comm_migrate / ahip_comm_borders
    run.ghosts = (sim.x[sim.nlocal: sim.nall].clone().numpy(), sim.mtype[sim.nlocal: sim.nall].clone().numpy(), sim.tag[: sim.nlocal].clone().numpy())
    model.close()
    return f, x.numpy(), e, nreb, nloc


def main():
    out, libpath, model_dir = sys.argv[1:4]
    temperature = float(sys.argv[4]) if len(sys.argv) > 4 else 300.0
    nsteps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
    skin = float(sys.argv[6]) if len(sys.argv) > 6 else 1.0
    dist.init_process_group(backend="gloo")
    rank = dist.get_rank()
    world = dist.get_world_size()
    grid = md.choose_grid(world)
    lib = capi.Library(libpath)
    cfg = model_file.model_S(model_dtype="float64", num_scalar_features=16, num_tensor_features=8, mlp_width=16,
                             readout_width=8, avg_num_neighbors=28.0)
    w = model_file.init_weights(cfg)
    path = os.path.join(model_dir, f"md_small_r{rank}.ahip")
    model_file.save_ahip(path, cfg, w)
    cell, pos, _ = lmp_like.diamond_si(3)
    if len(sys.argv) > 7 and sys.argv[7] == "cluster":
        # a 64-atom cluster in a 40 A box, all of it inside the first brick of a 2x1x1 grid and within reach of the second: rank 0 owns every
        # atom, receives no ghost and still has a slab to send; rank 1 owns nothing and receives ghosts (the peer's receive must be met)
        _, pos, _ = lmp_like.diamond_si(2)
        pos = pos - pos.min(axis=0) + np.array([8.5, 14.0, 14.0])
        cell = np.diag([40.0, 40.0, 40.0])
    vel = md.maxwell_boltzmann(len(pos), np.full(len(pos), 28.0855), temperature, 12345)
    toggle = len(sys.argv) > 7 and sys.argv[7] == "toggle"
    f2, x2, e2, nreb, nloc = run(lib, path, cell, pos, vel, cfg, grid, rank, dist, nsteps, skin, toggle)
    nl = torch.tensor([nloc]); dist.all_reduce(nl)
    assert int(nl.item()) == len(pos), "atoms lost or duplicated in migration"
    lib_plan = run.lib_plan
    want_lib = os.environ.get("AHIP_LIB_BORDERS", "1") != "0" and world > 1
    assert lib_plan == want_lib, f"re-neighboring path: library={lib_plan}, expected {want_lib}"
    if os.environ.get("AHIP_TEST_COMPARE_BORDERS") == "1":
        # the same run once more with the torch swap chain (md.py: _migrate / _borders): every rank must end with the same owned atoms in the same order and
        # the same ghost rows -- positions bit for bit, types, ORDER included (both build their send lists in ascending row order)
        g_lib = run.ghosts
        os.environ["AHIP_LIB_BORDERS"] = "0"
        run(lib, path, cell, pos, vel, cfg, grid, rank, dist, nsteps, skin, toggle)
        os.environ["AHIP_LIB_BORDERS"] = "1"
        g_ref = run.ghosts
        assert not run.lib_plan
        assert g_lib[0].shape == g_ref[0].shape and np.array_equal(g_lib[2], g_ref[2]), (rank, g_lib[0].shape, g_ref[0].shape)
        assert np.array_equal(g_lib[0], g_ref[0]) and np.array_equal(g_lib[1], g_ref[1]), f"rank {rank}: ghost rows differ"
    if rank == 0:
        f1, x1, e1, nreb1, _ = run(lib, path, cell, pos, vel, cfg, (1, 1, 1), 0, None, nsteps, skin)
        box = np.diag(cell)
        x1 = x1 - np.floor(x1 / box) * box
        x2 = x2 - np.floor(x2 / box) * box
        np.savez(out, f1=f1, f2=f2, x1=x1, x2=x2, e1=e1, e2=e2, nreb=nreb, nreb1=nreb1)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
