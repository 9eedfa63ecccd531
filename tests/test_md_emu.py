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


def test_world_size_2_gloo_matches_single_rank(emu_lib, model_dir, tmp_path):
    """Two processes (gloo), 2x1x1 bricks: forces after setup and positions after 3 steps equal the
    single-rank run; exercises borders, forward and reverse comm across ranks."""
    out = tmp_path / "mr.npz"
    worker = os.path.join(ROOT, "tests", "md_worker.py")
    env = dict(os.environ, PYTHONPATH=ROOT + ":" + os.path.join(ROOT, "tests"), MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29731", worker, str(out), emu_lib.path, model_dir]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert r.returncode == 0, r.stdout.decode()[-3000:]
    z = np.load(out)
    np.testing.assert_allclose(z["f2"], z["f1"], atol=1e-10)
    np.testing.assert_allclose(z["x2"], z["x1"], atol=1e-12)
    np.testing.assert_allclose(z["e2"], z["e1"], rtol=1e-12)
