"""MD driver logic on the CPU: Simulation (decomposition, borders, forward/reverse comm, rebuild,
NVE) running on CPU tensors through the host-emulation library (same kernel sources as the GPU),
single rank and world_size-2 gloo."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import util
from oracle import allegro_torch
from pair_allegro_amd import capi, lmp_like, md, model_file

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _small_cfg():
    # small widths keep the emulated kernels fast; float64 makes conservation checks sharp
    return model_file.model_S(model_dtype="float64", num_scalar_features=16, num_tensor_features=8, mlp_width=16,
                              readout_width=8, avg_num_neighbors=28.0)


def test_single_rank_forces_and_energy_conservation(emu_lib, model_dir):
    cfg = _small_cfg()
    w = model_file.init_weights(cfg)
    path = os.path.join(model_dir, "md_small.ahip")
    model_file.save_ahip(path, cfg, w)
    cell, pos, types = lmp_like.diamond_si(3)
    ref = util.oracle_run(cfg, w, cell, pos, types, ["Si"])
    model = capi.Model(path, 0, emu_lib)
    vel = md.maxwell_boltzmann(len(pos), np.full(len(pos), 28.0855), 300.0, 12345)
    sim = md.Simulation(md.HipBackend(model, [28.0855]), np.diag(cell), cfg["r_max"], 1.0, pos, np.zeros(len(pos), np.int32),
                        vel, torch.device("cpu"), dt=0.001)
    sim.setup()
    np.testing.assert_allclose(sim.gather_forces(), ref["forces"], atol=1e-9)
    t0 = sim.thermo([28.0855])
    np.testing.assert_allclose(t0["pe"], ref["pe"], rtol=1e-10)
    np.testing.assert_allclose(t0["virial"], ref["virial"], atol=1e-8)
    for _ in range(5):
        sim.step()
    t1 = sim.thermo([28.0855])
    assert abs((t1["pe"] + t1["ke"]) - (t0["pe"] + t0["ke"])) < 2e-5 * len(pos)
    assert abs(t1["pe"] - t0["pe"]) > 1e-9            # something actually moved
    model.close()


def _run_workers(emu_lib, model_dir, tmp_path, world, port, extra=(), env_extra=None):
    out = tmp_path / f"mr{world}.npz"
    worker = os.path.join(ROOT, "tests", "md_worker.py")
    env = dict(os.environ, PYTHONPATH=ROOT + ":" + os.path.join(ROOT, "tests"), MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    env.update(env_extra or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), worker, str(out), emu_lib.path, model_dir] + [str(e) for e in extra]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1500)
    assert r.returncode == 0, r.stdout.decode()[-3000:]
    return np.load(out)


@pytest.mark.parametrize("world,port", [(2, 29731), (4, 29733), (8, 29735)])
def test_gloo_ranks_match_single_rank(emu_lib, model_dir, tmp_path, world, port):
    """2 / 4 / 8 processes (gloo) on 2x1x1 / 2x2x1 / 2x2x2 bricks -- the grids bench.py uses on 2 / 4 / 8 GPUs:
    forces after setup, positions and energy after 3 steps equal the single-rank run (borders, forward and
    reverse communication across ranks and across periodic self-images)."""
    z = _run_workers(emu_lib, model_dir, tmp_path, world, port)
    np.testing.assert_allclose(z["f2"], z["f1"], atol=1e-10)
    np.testing.assert_allclose(z["x2"], z["x1"], atol=1e-12)
    np.testing.assert_allclose(z["e2"], z["e1"], rtol=1e-12)


def test_gloo_rebuild_and_migration(emu_lib, model_dir, tmp_path):
    """Hot system (6000 K), skin 0.2 A, 40 steps on 2x2x1 bricks: several re-neighborings, atoms cross brick
    faces (migration) -- no atom lost, trajectory equals the single-rank run."""
    z = _run_workers(emu_lib, model_dir, tmp_path, 4, 29737, extra=(6000.0, 40, 0.2))
    assert int(z["nreb"]) >= 3 and int(z["nreb1"]) >= 3
    np.testing.assert_allclose(z["x2"], z["x1"], atol=1e-8)
    np.testing.assert_allclose(z["e2"], z["e1"], rtol=1e-9)


@pytest.mark.parametrize("world,port", [(2, 29751), (4, 29753), (8, 29755)])
def test_gloo_library_borders_equal_the_swap_chain(emu_lib, model_dir, tmp_path, world, port):
    """at several ranks a re-neighboring is library kernels + the communicator's transport (ahip_comm_migrate, ahip_comm_borders: csrc/comm.hip),
    not six rounds of torch mask / nonzero / cat.  Hot system, small skin, 12 steps on 2 / 4 / 8 bricks (migrations included): after the last re-neighboring
    every rank holds the same owned atoms in the same order and the same ghost rows -- bit for bit, order included -- as the torch swap chain
    (AHIP_LIB_BORDERS=0, what LAMMPS' Comm::exchange / Comm::borders do for the reference, /root/reference/pair_nequip_allegro.cpp:366-368), and the
    trajectory equals the single-rank run."""
    z = _run_workers(emu_lib, model_dir, tmp_path, world, port, extra=(4000.0, 12, 0.3), env_extra={"AHIP_TEST_COMPARE_BORDERS": "1"})
    assert int(z["nreb"]) >= 2
    np.testing.assert_allclose(z["x2"], z["x1"], atol=1e-8)
    np.testing.assert_allclose(z["e2"], z["e1"], rtol=1e-9)


def test_gloo_library_borders_overflow_is_agreed_on_and_retried(emu_lib, model_dir, tmp_path):
    """ahip_comm_borders with arrays that are too small (8 ghost rows): every rank gets the same "needs N rows" answer -- the ranks that overflow keep exchanging messages of the
    announced sizes, nobody waits for ever -- and the caller's retry with larger arrays gives the ordinary result (4 ranks, forces and trajectory equal the single-rank run)."""
    z = _run_workers(emu_lib, model_dir, tmp_path, 4, 29759, extra=(300.0, 3, 1.0), env_extra={"AHIP_TEST_SMALL_BORDERS_CAP": "1"})
    np.testing.assert_allclose(z["f2"], z["f1"], atol=1e-10)
    np.testing.assert_allclose(z["x2"], z["x1"], atol=1e-12)


def test_gloo_torch_swap_chain_still_works(emu_lib, model_dir, tmp_path):
    """the torch re-neighboring path (AHIP_LIB_BORDERS=0; also what a backend without the library communicator runs): migration test on 2x2x1 bricks"""
    z = _run_workers(emu_lib, model_dir, tmp_path, 4, 29757, extra=(6000.0, 20, 0.2), env_extra={"AHIP_LIB_BORDERS": "0"})
    assert int(z["nreb"]) >= 2
    np.testing.assert_allclose(z["x2"], z["x1"], atol=1e-8)


def test_gloo_switching_the_exchange_schedule_keeps_the_trajectory(emu_lib, model_dir, tmp_path):
    """`bench.py --gpus N` times the overlapped and the serial schedule and keeps the faster one (`Simulation.set_overlap`).  Hot system, small skin
    (re-neighborings in both modes, one of them forced by switching overlap back on after a serial re-neighboring), the schedule flipped every third
    step on 2x2x1 bricks: positions and energy equal the single-rank run."""
    z = _run_workers(emu_lib, model_dir, tmp_path, 4, 29741, extra=(4000.0, 24, 0.3, "toggle"))
    assert int(z["nreb"]) >= 2
    np.testing.assert_allclose(z["x2"], z["x1"], atol=1e-8)
    np.testing.assert_allclose(z["e2"], z["e1"], rtol=1e-9)


def test_gloo_empty_brick_next_to_a_cluster(emu_lib, model_dir, tmp_path):
    """2x1x1 bricks, every atom in brick 0 (a cluster in a large box): rank 0 receives no ghosts but sends a slab, rank 1 owns nothing.
    The exchange plan of a rank without ghosts must still post its sends (the single-rank plan was taken whenever
    nall == nlocal and the peer's receive waited forever); trajectory and energy equal the single-rank run."""
    z = _run_workers(emu_lib, model_dir, tmp_path, 2, 29739, extra=(300.0, 3, 1.0, "cluster"))
    np.testing.assert_allclose(z["f2"], z["f1"], atol=1e-10)
    np.testing.assert_allclose(z["x2"], z["x1"], atol=1e-12)
    np.testing.assert_allclose(z["e2"], z["e1"], rtol=1e-12)
    assert np.abs(z["f1"]).max() > 1e-3


def test_interior_boundary_split_equals_one_call(emu_lib, model_dir):
    """The overlapped schedule (interior-first ordering, three ahip_compute_dev_range calls per evaluation, exchange between
    them) gives the forces, energy and virial of the single ahip_compute_dev call on the unordered atoms -- equal up to
    the order of the float64 sums -- and the same trajectory."""
    cfg = _small_cfg()
    w = model_file.init_weights(cfg)
    path = os.path.join(model_dir, "md_small_ov.ahip")
    model_file.save_ahip(path, cfg, w)
    cell, pos, _ = lmp_like.diamond_si(4)                 # 21.7 A box: a genuine interior (atoms > 6 A from every face)
    vel = md.maxwell_boltzmann(len(pos), np.full(len(pos), 28.0855), 300.0, 7)
    out = {}
    for ov in (True, False):
        model = capi.Model(path, 0, emu_lib)
        sim = md.Simulation(md.HipBackend(model, [28.0855]), np.diag(cell), cfg["r_max"], 1.0, pos, np.zeros(len(pos), np.int32),
                            vel, torch.device("cpu"), dt=0.001, overlap=ov)
        sim.setup()
        f0 = sim.gather_forces()
        t0 = sim.thermo([28.0855])
        for _ in range(3):
            sim.step()
        out[ov] = (f0, t0, sim.gather_forces(), sim.thermo([28.0855]), sim.n_int, sim.nlocal)
        model.close()
    assert 0 < out[True][4] < out[True][5]                # both classes are populated
    np.testing.assert_allclose(out[True][0], out[False][0], atol=1e-12)
    np.testing.assert_allclose(out[True][1]["pe"], out[False][1]["pe"], rtol=1e-13)
    np.testing.assert_allclose(out[True][1]["virial"], out[False][1]["virial"], atol=1e-10)
    np.testing.assert_allclose(out[True][2], out[False][2], atol=1e-10)
    np.testing.assert_allclose(out[True][3]["pe"], out[False][3]["pe"], rtol=1e-12)


def test_neigh_modify_every_delay_check(emu_lib, model_dir):
    """`neigh_modify every N delay M check yes|no`: with `check no` the list is rebuilt exactly on the allowed steps;
    with `check yes` and a generous skin never within a few steps; forces stay those of the always-fresh list."""
    cfg = _small_cfg()
    w = model_file.init_weights(cfg)
    path = os.path.join(model_dir, "md_small_nm.ahip")
    model_file.save_ahip(path, cfg, w)
    cell, pos, _ = lmp_like.diamond_si(3)
    vel = md.maxwell_boltzmann(len(pos), np.full(len(pos), 28.0855), 300.0, 3)

    def run(**kw):
        model = capi.Model(path, 0, emu_lib)
        sim = md.Simulation(md.HipBackend(model, [28.0855]), np.diag(cell), cfg["r_max"], 1.0, pos, np.zeros(len(pos), np.int32),
                            vel, torch.device("cpu"), dt=0.001, **kw)
        sim.setup()
        builds = []
        for _ in range(8):
            n0 = sim.nrebuild
            sim.step()
            builds.append(sim.nrebuild - n0)
        f = sim.gather_forces()
        model.close()
        return builds, f

    b_default, f_default = run()
    assert sum(b_default) == 0                                        # 8 fs at 300 K: nobody moves skin/2 = 0.5 A
    b_every, f_every = run(neigh_every=3, neigh_delay=0, neigh_check=False)
    assert b_every == [0, 0, 1, 0, 0, 1, 0, 0]
    b_delay, _ = run(neigh_every=1, neigh_delay=4, neigh_check=False)
    assert b_delay == [0, 0, 0, 1, 0, 0, 0, 1]
    np.testing.assert_allclose(f_every, f_default, atol=1e-9)         # a fresher list changes nothing while the skin holds
    with pytest.raises(ValueError):
        run(neigh_every=0)


@pytest.mark.parametrize("ncell,two_types", [(3, False), (2, True)])
def test_library_borders_equal_the_swap_chain_borders(emu_lib, model_dir, ncell, two_types):
    """One rank: ahip_borders_local_dev (count / scan / fill in the library) produces the same SET of ghosts -- (source atom, shift) pairs, positions,
    types -- as the per-dimension swap chain (reference: LAMMPS Comm::borders, which pair_allegro relies on for its ghosts), also when an atom needs
    images in several directions and when a capacity guess is too small (the call reports the count, the caller retries)."""
    cfg = _small_cfg()
    if two_types:
        cfg = dict(cfg, type_names=["Si", "C"])
    path = os.path.join(model_dir, f"md_borders{ncell}.ahip")
    model_file.save_ahip(path, cfg, model_file.init_weights(cfg))
    cell, pos, _ = lmp_like.diamond_si(ncell)
    rng = np.random.default_rng(5)
    pos = pos + rng.normal(0, 0.05, pos.shape)
    box = np.diag(cell)
    pos -= np.floor(pos / box) * box
    mt = (np.arange(len(pos)) % 2).astype(np.int32) if two_types else np.zeros(len(pos), np.int32)
    vel = np.zeros_like(pos)

    def ghosts(use_lib):
        model = capi.Model(path, 0, emu_lib)
        sim = md.Simulation(md.HipBackend(model, [28.0855, 12.011]), box, cfg["r_max"], 1.0, pos, mt, vel, torch.device("cpu"), dt=0.001)
        nl = sim.nlocal                                           # (the constructor built the borders once; build them again the chosen way)
        sim.x, sim.mtype = sim.x[:nl].contiguous(), sim.mtype[:nl].contiguous()
        if not use_lib:
            sim._borders_local = lambda: False
        else:
            sim._nghost_last = 8                                  # a deliberately small first guess: exercises the retry
        sim._borders()
        assert sim.nall == sim.x.shape[0] == sim.mtype.shape[0]
        g = np.concatenate([sim._ghost_src.numpy()[:, None].astype(np.float64), sim._ghost_shift.numpy(), sim.x[nl:].numpy(),
                            sim.mtype[nl:].numpy()[:, None].astype(np.float64)], axis=1)
        xl = sim.x[:nl].numpy().copy()
        model.close()
        return xl, g[np.lexsort(g.T[::-1])]

    xa, ga = ghosts(False)
    xb, gb = ghosts(True)
    np.testing.assert_array_equal(xa, xb)
    assert ga.shape == gb.shape and ga.shape[0] > 0
    np.testing.assert_array_equal(ga[:, :4], gb[:, :4])          # same (source, shift) set
    np.testing.assert_allclose(ga[:, 4:7], gb[:, 4:7], rtol=0, atol=1e-12)
    np.testing.assert_array_equal(ga[:, 7], gb[:, 7])
    np.testing.assert_allclose(gb[:, 4:7], xb[gb[:, 0].astype(int)] + gb[:, 1:4], rtol=0, atol=1e-12)
