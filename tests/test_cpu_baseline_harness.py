"""The libtorch CPU-baseline harness (oracle/cpu_baseline.cpp): the reference's load / freeze / preprocess / forward(Dict) /
scatter sequence (pair_nequip_allegro.cpp:214-231, 409-430, 457-650, 358-393) in C++ on the oracle's TorchScript export must
reproduce the Python oracle (oracle/glue.py + oracle/allegro_torch.py) -- the two restatements of the reference's host glue pin
each other, and bench.py's cpu_baseline leg times exactly this binary."""
import json
import os
import subprocess

import numpy as np
import pytest

import util
from oracle import allegro_torch
from pair_allegro_amd import lmp_like, model_file

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def harness():
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "cpu_baseline"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        pytest.skip("libtorch headers / libraries not usable here: " + r.stdout.decode()[-200:])
    return os.path.join(ROOT, "oracle", "_build", "cpu_baseline")


def _write_system(path, rs, ntypes):
    with open(path, "wb") as f:
        np.array([rs.nlocal, rs.nghost, ntypes, int(rs.offsets[-1])], dtype=np.int32).tofile(f)
        rs.x.astype(np.float64).tofile(f); rs.type.astype(np.int32).tofile(f); rs.tag.astype(np.int32).tofile(f)
        rs.numneigh.astype(np.int32).tofile(f); rs.flat.astype(np.int32).tofile(f)


def test_harness_reproduces_the_python_oracle(harness, tmp_path):
    g = util.load_golden("Cu2AgO4_r5")                       # 3 types, triclinic, LAMMPS order != model order
    cfg = model_file.model_S(model_dtype="float64", type_names=["Cu", "Ag", "O"], per_edge_type_cutoff=[[5.0, 4.5, 4.0], [4.5, 5.0, 4.2], [4.0, 4.2, 4.8]],
                             num_scalar_features=16, num_tensor_features=8, mlp_width=16, readout_width=8, avg_num_neighbors=30.0)
    w = model_file.init_weights(cfg)
    pth = str(tmp_path / "m.nequip.pth")
    allegro_torch.export_nequip_pth(pth, cfg, w)
    types, names = util.lammps_types(g)
    rs = lmp_like.build_rank_system(g["cell"], g["pos"], types, 6.0)
    sysf, outf = str(tmp_path / "sys.bin"), str(tmp_path / "out.bin")
    _write_system(sysf, rs, len(names))
    r = subprocess.run([harness, sysf, pth, "--out", outf, "--warmup", "1", "--reps", "3"] + names, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    info = json.loads(r.stdout.decode().strip().splitlines()[-1])
    assert info["reps"] == 3 and info["nlocal"] == rs.nlocal and info["threads"] >= 1
    out = np.fromfile(outf)
    ref = util.oracle_run(cfg, w, g["cell"], g["pos"], types, names)
    assert info["nedges"] == ref["inputs"]["edge_index"].shape[1]
    f = out[7:7 + 3 * rs.nall].reshape(-1, 3)
    forces = np.zeros_like(ref["forces"])
    np.add.at(forces, rs.tag - 1, f)
    np.testing.assert_allclose(out[0], ref["pe"], rtol=1e-12)
    np.testing.assert_allclose(forces, ref["forces"], atol=1e-10)
    np.testing.assert_allclose(out[1:7], ref["virial"], atol=1e-9)
    eatom = out[7 + 3 * rs.nall:]
    np.testing.assert_allclose(eatom[: rs.nlocal], ref["eatom"][rs.tag[: rs.nlocal] - 1], atol=1e-11)
