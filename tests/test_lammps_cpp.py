"""The C++ LAMMPS `Pair` subclass (pair_allegro_amd/lammps/pair_allegro_hip.cpp) compiled against a minimal
test-only shim of the LAMMPS declarations it uses and driven with the LAMMPS call sequence
(settings -> coeff -> init_style -> init_one -> compute), linked to the host-emulation library.
Checks the real marshalling code, not its Python mirror."""
import os
import subprocess

import numpy as np
import pytest

import util
from oracle import allegro_torch
from pair_allegro_amd import lmp_like, model_file

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "tests", "lammps_shim")


@pytest.fixture(scope="module")
def driver(emu_lib):
    subprocess.run(["make", "-C", SHIM], check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    return os.path.join(SHIM, "_build", "driver")


def _write_system(path, rs, ntypes):
    with open(path, "wb") as f:
        np.array([rs.nlocal, rs.nghost, ntypes, int(rs.offsets[-1])], dtype=np.int32).tofile(f)
        rs.x.astype(np.float64).tofile(f)
        rs.type.astype(np.int32).tofile(f)
        rs.tag.astype(np.int32).tofile(f)
        rs.numneigh.astype(np.int32).tofile(f)
        rs.flat.astype(np.int32).tofile(f)


def test_cpp_pair_style_matches_oracle(driver, tmp_path, model_dir):
    g = util.load_golden("Cu2AgO4_r5")                       # 3 model types, non-identity LAMMPS->model type map
    cfg = model_file.model_S(model_dtype="float64", type_names=["Cu", "Ag", "O"], num_scalar_features=16,
                             num_tensor_features=8, mlp_width=16, readout_width=8, avg_num_neighbors=37.0)
    w = model_file.init_weights(cfg)
    mpath = os.path.join(model_dir, "cpp.nequip.pth")
    allegro_torch.export_nequip_pth(mpath, cfg, w)
    types, names = util.lammps_types(g)                       # Ag Cu O (alphabetical)
    rs = lmp_like.build_rank_system(g["cell"], g["pos"], types, 6.0)
    sysf, outf = str(tmp_path / "sys.bin"), str(tmp_path / "out.bin")
    _write_system(sysf, rs, len(names))
    r = subprocess.run([driver, sysf, outf, mpath] + names, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       env=dict(os.environ, DRIVER_COMPUTES="1"))
    text = r.stdout.decode()
    assert r.returncode == 0, text
    assert "restartinfo=0 manybody=1 no_fdotr=1 setflag11=1" in text
    assert "NequIP/Allegro: Loading model from" in text and "0 | Cu | 2 | Cu" in text
    out = np.fromfile(outf, dtype=np.float64)
    cut, eng, vir = out[0], out[1], out[2:8]
    f = out[8:8 + 3 * rs.nall].reshape(-1, 3)
    eatom = out[8 + 3 * rs.nall: 8 + 4 * rs.nall]
    extra = out[8 + 4 * rs.nall:]                              # the three `compute allegro` results (C++ ComputeAllegroHIP)
    c_vir, c_f, c_e = extra[:9].reshape(3, 3), extra[9:9 + 3 * rs.nlocal].reshape(-1, 3), extra[9 + 3 * rs.nlocal:]
    ref = util.oracle_run(cfg, w, g["cell"], g["pos"], types, names)
    forces = np.zeros_like(ref["forces"])
    np.add.at(forces, rs.tag - 1, f)
    assert cut == 5.0
    np.testing.assert_allclose(forces, 2.0 * ref["forces"], atol=1e-9)      # two compute() calls: f is accumulated
    np.testing.assert_allclose(eng, ref["pe"], rtol=1e-10)
    np.testing.assert_allclose(vir, ref["virial"], atol=1e-8)
    np.testing.assert_allclose(eatom[: rs.nlocal], ref["eatom"][rs.tag[: rs.nlocal] - 1], atol=1e-10)
    xx, yy, zz, xy, xz, yz = ref["virial"]
    np.testing.assert_allclose(c_vir, [[xx, xy, xz], [xy, yy, yz], [xz, yz, zz]], atol=1e-8)      # compute allegro virial 9
    np.testing.assert_allclose(c_f, ref["forces"][rs.tag[: rs.nlocal] - 1], atol=1e-9)             # allegro/atom forces 3 1
    np.testing.assert_allclose(c_e, ref["eatom"][rs.tag[: rs.nlocal] - 1], atol=1e-10)             # allegro/atom atomic_energy 1 0


def test_cpp_pair_style_list_rebuilt_twice_at_one_timestep(driver, tmp_path, model_dir):
    """`run 0` -> atoms displaced + list rebuilt in another order -> `run 0` at the SAME timestep: the second run must use
    the fresh list (hand-over keyed on neighbor->ago == 0 and reset by init_style, not on the timestep of the last build).
    The reference walks the list every step (pair_nequip_allegro.cpp:488-512) and has no such state."""
    cfg = model_file.model_S(model_dtype="float64", num_scalar_features=16, num_tensor_features=8, mlp_width=16, readout_width=8)
    w = model_file.init_weights(cfg)
    mpath = os.path.join(model_dir, "cpp_si_ago.nequip.pth")
    allegro_torch.export_nequip_pth(mpath, cfg, w)
    cell, pos, types = lmp_like.diamond_si(2)
    rs = lmp_like.build_rank_system(cell, pos, types, 6.0)
    sysf, outf = str(tmp_path / "sys.bin"), str(tmp_path / "out.bin")
    _write_system(sysf, rs, 1)
    r = subprocess.run([driver, sysf, outf, mpath, "Si"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       env=dict(os.environ, DRIVER_REBUILD_SAME_STEP="1"))
    assert r.returncode == 0, r.stdout.decode()
    out = np.fromfile(outf, dtype=np.float64)
    f = out[8:8 + 3 * rs.nall].reshape(-1, 3)
    ref = util.oracle_run(cfg, w, cell, pos, types, ["Si"])
    forces = np.zeros_like(ref["forces"])
    np.add.at(forces, rs.tag - 1, f)
    np.testing.assert_allclose(forces, 2.0 * ref["forces"], atol=1e-9)
    np.testing.assert_allclose(out[1], ref["pe"], rtol=1e-10)


def test_cpp_pair_style_deck_errors(driver, tmp_path, model_dir):
    cfg = model_file.model_S(model_dtype="float64", num_scalar_features=16, num_tensor_features=8, mlp_width=16, readout_width=8)
    mpath = os.path.join(model_dir, "cpp_si.nequip.pth")
    allegro_torch.export_nequip_pth(mpath, cfg)
    cell, pos, types = lmp_like.diamond_si(2)
    rs = lmp_like.build_rank_system(cell, pos, types, 6.0)
    sysf, outf = str(tmp_path / "sys.bin"), str(tmp_path / "out.bin")
    _write_system(sysf, rs, 1)
    run = lambda *a: subprocess.run([driver, sysf, outf] + list(a), stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    r = run(mpath, "Si", "Extra")
    assert r.returncode == 10 and b"Incorrect args for pair coefficients, should be * * <model>.nequip.pth/pt2" in r.stdout
    r = run(str(tmp_path / "model.pt"), "Si")
    assert r.returncode == 11 and b"Only accepts model paths with extension" in r.stdout
    r = run(mpath, "Ge")                                       # type name not in the model -> unmapped -> compute error
    assert r.returncode == 10 and b"not mapped" in r.stdout


def test_cpp_plugin_registers_the_three_styles(driver):
    """`plugin load` packaging: lammpsplugin_init registers pair allegro, compute allegro and compute allegro/atom, and the
    pair factory returns a usable object (no LAMMPS rebuild needed)."""
    r = subprocess.run([driver, "--plugin"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    text = r.stdout.decode()
    assert r.returncode == 0, text
    for name in ("registered pair:allegro", "registered compute:allegro", "registered compute:allegro/atom"):
        assert name in text
    assert "pair object ok restartinfo=0 manybody=1" in text
    # `pair_style nequip` (pair_nequip_allegro.cpp:86-89): registered only to stop with a clear message
    assert "registered pair:nequip" in text and "pair_style nequip -> error->all: pair_style nequip is not provided" in text
