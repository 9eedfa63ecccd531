"""The KOKKOS coupling class (pair_allegro_amd/lammps/pair_allegro_hip_kokkos.cpp, `pair_style allegro/kk`) compiled against
the test-only Kokkos / LAMMPS-KOKKOS shim and driven with the LAMMPS call sequence.  On the CPU the "device" views are host
memory and the class talks to the host-emulation library; the `gpu` variant builds the same driver with hipMalloc'ed views
against the real liballegro_hip.so, so x / f / type / the neighbor table are genuine device pointers.
Reference behaviour mirrored: /root/reference/pair_nequip_allegro_kokkos.cpp:86-353 (compute), :364-406 (coeff, init_style)."""
import os
import subprocess

import numpy as np
import pytest

import util
from oracle import allegro_torch
from pair_allegro_amd import capi, lmp_like, model_file
from test_lammps_cpp import _write_system

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "tests", "lammps_shim")


def _build(target):
    subprocess.run(["make", "-C", SHIM, target], check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)


@pytest.fixture(scope="module")
def driver_kk(emu_lib):
    _build("kk")
    return os.path.join(SHIM, "_build", "driver_kk")


def _run_and_check(drv, tmp_path, model_dir, dtype, atol_f, rtol_e):
    g = util.load_golden("Cu2AgO4_r5")                       # 3 model types, non-identity LAMMPS->model type map
    cfg = model_file.model_S(model_dtype=dtype, type_names=["Cu", "Ag", "O"], num_scalar_features=16,
                             num_tensor_features=8, mlp_width=16, readout_width=8, avg_num_neighbors=37.0)
    w = model_file.init_weights(cfg)
    mpath = os.path.join(model_dir, f"kk_{dtype}.nequip.pth")
    allegro_torch.export_nequip_pth(mpath, cfg, w)
    types, names = util.lammps_types(g)
    rs = lmp_like.build_rank_system(g["cell"], g["pos"], types, 6.0)
    sysf, outf = str(tmp_path / "sys.bin"), str(tmp_path / "out.bin")
    _write_system(sysf, rs, len(names))
    r = subprocess.run([drv, sysf, outf, mpath] + names, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=dict(os.environ, DRIVER_COMPUTES="1"))
    text = r.stdout.decode()
    assert r.returncode == 0, text
    assert "restartinfo=0 manybody=1 no_fdotr=1 respa=0 kokkosable=1" in text
    out = np.fromfile(outf, dtype=np.float64)
    cut, eng, vir = out[0], out[1], out[2:8]
    f = out[8:8 + 3 * rs.nall].reshape(-1, 3)
    eatom = out[8 + 3 * rs.nall: 8 + 4 * rs.nall]
    eng2 = out[8 + 4 * rs.nall]
    c_f = out[9 + 4 * rs.nall: 9 + 4 * rs.nall + 3 * rs.nlocal].reshape(-1, 3)     # compute allegro/atom forces 3 1 (device path)
    c_vir = out[9 + 4 * rs.nall + 3 * rs.nlocal:].reshape(3, 3)                        # compute allegro virial 9
    ref = util.oracle_run(cfg, w, g["cell"], g["pos"], types, names)
    forces = np.zeros_like(ref["forces"])
    np.add.at(forces, rs.tag - 1, f)
    assert cut == 5.0
    np.testing.assert_allclose(forces, 2.0 * ref["forces"], atol=atol_f)      # two compute() calls accumulate into the force view
    np.testing.assert_allclose(eng, ref["pe"], rtol=rtol_e)
    np.testing.assert_allclose(eng2, ref["pe"], rtol=rtol_e)
    np.testing.assert_allclose(vir, ref["virial"], atol=100 * atol_f)         # virial of the first call (the second had vflag = 0)
    np.testing.assert_allclose(eatom[: rs.nlocal], ref["eatom"][rs.tag[: rs.nlocal] - 1], atol=10 * atol_f)
    np.testing.assert_allclose(c_f, ref["forces"][rs.tag[: rs.nlocal] - 1], atol=atol_f)
    xx, yy, zz, xy, xz, yz = ref["virial"]
    np.testing.assert_allclose(c_vir, [[xx, xy, xz], [xy, yy, yz], [xz, yz, zz]], atol=100 * atol_f)


def test_kokkos_pair_style_matches_oracle(driver_kk, tmp_path, model_dir):
    _run_and_check(driver_kk, tmp_path, model_dir, "float64", 1e-9, 1e-10)


def test_kokkos_pair_style_requires_neigh_half(driver_kk, tmp_path, model_dir):
    """`package kokkos neigh full` stops with the reference's message (pair_nequip_allegro_kokkos.cpp:399-401)."""
    cfg = model_file.model_S(model_dtype="float64", num_scalar_features=16, num_tensor_features=8, mlp_width=16, readout_width=8)
    mpath = os.path.join(model_dir, "kk_si.nequip.pth")
    allegro_torch.export_nequip_pth(mpath, cfg)
    cell, pos, types = lmp_like.diamond_si(2)
    rs = lmp_like.build_rank_system(cell, pos, types, 6.0)
    sysf, outf = str(tmp_path / "sys.bin"), str(tmp_path / "out.bin")
    _write_system(sysf, rs, 1)
    r = subprocess.run([driver_kk, sysf, outf, mpath, "Si"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       env=dict(os.environ, DRIVER_KK_NEIGH_FULL="1"))
    assert r.returncode == 10 and b"pair style allegro/kk requires the 'neigh half' flag due to 'newton on'" in r.stdout
    r = subprocess.run([driver_kk, sysf, outf, mpath, "Ge"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)   # unmapped type
    assert r.returncode == 10 and b"not mapped" in r.stdout


def test_table_list_and_strict_cutoff_on_the_emulation(emu_lib, model_dir):
    """ahip_neigh_update_dev_table (row-major AND column-major tables) installs the same rows as the pointer-list hand-over, and
    cutoff_compare=lt drops exactly the pairs sitting ON the cutoff (ideal diamond, r_max = the third-shell distance)."""
    a = 5.431
    r3 = a * np.sqrt(11.0) / 4.0                               # third neighbour shell of diamond
    cfg = model_file.model_S(model_dtype="float64", r_max=float(r3), num_scalar_features=16, num_tensor_features=8, mlp_width=16, readout_width=8)
    mpath = os.path.join(model_dir, "kk_strict.nequip.pth")
    allegro_torch.export_nequip_pth(mpath, cfg)
    cell, pos, types = lmp_like.diamond_si(3, a=a, jitter=0.0)
    rs = lmp_like.build_rank_system(cell, pos, types, float(r3) + 1.0)
    m = capi.Model(mpath, 0, emu_lib)
    x = np.ascontiguousarray(rs.x, dtype=np.float64)
    mt = np.zeros(rs.nall, dtype=np.int32)
    maxn = int(rs.numneigh[: rs.nlocal].max()) + 2
    counts = {}
    for layout in ("right", "left"):
        tab = np.full((rs.nlocal, maxn), 0x12345678, dtype=np.int32)
        for i in range(rs.nlocal):
            tab[i, : rs.numneigh[i]] = rs.flat[rs.offsets[i]: rs.offsets[i + 1]] | (1 << 29)
        store = np.ascontiguousarray(tab) if layout == "right" else np.asfortranarray(tab)
        sa, ss = (maxn, 1) if layout == "right" else (1, rs.nlocal)
        stream = 0
        il = np.arange(rs.nlocal, dtype=np.int32)
        nn = np.ascontiguousarray(rs.numneigh, dtype=np.int32)
        m.neigh_update_dev_table(rs.nlocal, rs.nall, il.ctypes.data, nn.ctypes.data, store.ctypes.data, sa, ss)
        for cmp_ in ("le", "lt"):
            m.set_option("cutoff_compare", cmp_)
            f = np.zeros((rs.nall, 3)); ev = np.zeros(7)
            m.compute_dev(rs.nlocal, rs.nghost, x.ctypes.data, mt.ctypes.data, f.ctypes.data, 0, ev.ctypes.data)
            counts[(layout, cmp_)] = m.nedges()
    n = rs.nlocal
    # diamond shells: 4 + 12 + 12 inside or on r3; the 12 third-shell pairs sit exactly on the cutoff up to rounding of the positions
    assert counts[("right", "le")] == counts[("left", "le")] and counts[("right", "lt")] == counts[("left", "lt")]
    assert counts[("right", "lt")] <= counts[("right", "le")] <= 28 * n and counts[("right", "lt")] >= 16 * n
    # exact case: two atoms 3.0 apart, cutoff 3.0 -> rsq == cut^2 in float64: kept by `<=` (host path), dropped by `<` (KOKKOS path)
    cfg3 = model_file.model_S(model_dtype="float64", r_max=3.0, num_scalar_features=16, num_tensor_features=8, mlp_width=16, readout_width=8)
    p3 = os.path.join(model_dir, "kk_strict3.nequip.pth")
    allegro_torch.export_nequip_pth(p3, cfg3)
    m3 = capi.Model(p3, 0, emu_lib)
    x2 = np.array([[0.0, 0.0, 0.0], [3.0, 0.0, 0.0], [0.0, 2.5, 0.0]])
    tab = np.array([[1, 2], [0, 2], [0, 1]], dtype=np.int32)
    il = np.arange(3, dtype=np.int32); nn = np.full(3, 2, dtype=np.int32); mt3 = np.zeros(3, dtype=np.int32)
    m3.neigh_update_dev_table(3, 3, il.ctypes.data, nn.ctypes.data, tab.ctypes.data, 2, 1)
    got = {}
    for cmp_ in ("le", "lt"):
        m3.set_option("cutoff_compare", cmp_)
        f = np.zeros((3, 3)); ev = np.zeros(7)
        m3.compute_dev(3, 0, x2.ctypes.data, mt3.ctypes.data, f.ctypes.data, 0, ev.ctypes.data)
        got[cmp_] = m3.nedges()
    assert got == {"le": 4, "lt": 2}                            # 0-2 (2.5) both ways always; 0-1 (3.0) only with `<=`; 1-2 is 3.9 apart


@pytest.mark.gpu
def test_kokkos_pair_style_on_device_pointers(hip_lib, tmp_path, model_dir):
    """Same driver, views in hipMalloc'ed memory, real liballegro_hip.so: the class hands genuine device pointers of x, f, type and
    of the column-major neighbor table to the `_dev` entry points (float32 model)."""
    _build("kk_hip")
    _run_and_check(os.path.join(SHIM, "_build", "driver_kk_hip"), tmp_path, model_dir, "float32", 2e-5, 1e-5)


def test_table_list_rejects_bad_input(emu_lib, model_dir):
    """ahip_neigh_update_dev_table / ahip_map_types_dev validate on the device and report through ahip_last_error: an out-of-range neighbour,
    an out-of-range centre, non-positive strides, an unmapped type."""
    cfg = model_file.model_S(model_dtype="float64", num_scalar_features=16, num_tensor_features=8, mlp_width=16, readout_width=8)
    p = os.path.join(model_dir, "kk_bad.nequip.pth")
    allegro_torch.export_nequip_pth(p, cfg)
    m = capi.Model(p, 0, emu_lib)
    il = np.arange(3, dtype=np.int32); nn = np.full(3, 2, dtype=np.int32)
    tab = np.array([[1, 2], [0, 7], [0, 1]], dtype=np.int32)            # neighbour 7 of a 3-atom system
    with pytest.raises(capi.AhipError, match="neighbour index out of range"):
        m.neigh_update_dev_table(3, 3, il.ctypes.data, nn.ctypes.data, tab.ctypes.data, 2, 1)
    bad_il = np.array([0, 1, 5], dtype=np.int32)
    tab[1, 1] = 2
    with pytest.raises(capi.AhipError, match="ilist entry out of range"):
        m.neigh_update_dev_table(3, 3, bad_il.ctypes.data, nn.ctypes.data, tab.ctypes.data, 2, 1)
    with pytest.raises(capi.AhipError, match="strides must be positive"):
        m.neigh_update_dev_table(3, 3, il.ctypes.data, nn.ctypes.data, tab.ctypes.data, 0, 1)
    m.neigh_update_dev_table(3, 3, il.ctypes.data, nn.ctypes.data, tab.ctypes.data, 2, 1)      # the corrected table installs
    assert m.nneigh() == 6
    types = np.array([1, 2, 1], dtype=np.int32); out = np.zeros(3, dtype=np.int32)
    with pytest.raises(capi.AhipError, match="not mapped"):
        m.map_types_dev(3, types.ctypes.data, np.array([0, -1], dtype=np.int32), out.ctypes.data)
    with pytest.raises(capi.AhipError, match="out of range"):
        m.map_types_dev(3, np.array([1, 3, 1], dtype=np.int32).ctypes.data, np.array([0, 0], dtype=np.int32), out.ctypes.data)
    m.map_types_dev(3, types.ctypes.data, np.array([0, 0], dtype=np.int32), out.ctypes.data)
    assert out.tolist() == [0, 0, 0]
