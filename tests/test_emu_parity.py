"""Kernel LOGIC parity on the CPU: the sync-free HIP sources compiled against the host stand-in
runtime (tests/host_emu) versus the golden vectors of the float64 torch oracle.  These tests do not
replace the GPU parity tests (tests/test_gpu_parity.py); they make the same sources debuggable in a
container without a GPU."""
import numpy as np
import pytest

import parity_cases as pc
import util


@pytest.mark.parametrize("tag", util.GOLDEN_TAGS)
@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_golden_single_rank(emu_lib, model_dir, tag, dtype):
    if tag == "Cu-cubic_r15" and dtype == "float32":
        pytest.skip("covered in float64; keeps the CPU suite short")
    res, g = pc.check_golden(emu_lib, model_dir, tag, dtype)
    assert res["info"]["path"] == ("generic_f64" if dtype == "float64" else "generic_f32")
    pc.check_edges_vs_brute_force(res, g)


@pytest.mark.parametrize("grid", [(2, 1, 1), (2, 2, 1)])
@pytest.mark.parametrize("tag", ["CuPd-cubic-big_r5", "Cu-cubic_r5"])
def test_golden_multi_rank(emu_lib, model_dir, tag, grid):
    """n_rank 2 and 4 like /root/reference/tests/test_python_repro_allegro.py:44-47,70-77:
    strictly local model => per-rank partial forces/energies/virials add up to the single-rank result."""
    res, g = pc.check_golden(emu_lib, model_dir, tag, "float64", grid=grid)
    pc.check_edges_vs_brute_force(res, g)


def test_chunked_equals_unchunked(emu_lib, model_dir):
    a, _ = pc.check_golden(emu_lib, model_dir, "Si64_r5", "float64")
    b, _ = pc.check_golden(emu_lib, model_dir, "Si64_r5", "float64", options={"chunk_edges": 100})
    np.testing.assert_allclose(a["forces"], b["forces"], atol=1e-12)
    np.testing.assert_allclose(a["virial"], b["virial"], atol=1e-10)


def test_neighbor_order_irrelevant(emu_lib, model_dir):
    pc.check_golden(emu_lib, model_dir, "Cu2AgO4_r5", "float64", shuffle_seed=3)


def test_precision_option_float64_on_float32_model(emu_lib, model_dir):
    pc.check_golden(emu_lib, model_dir, "Si64_r5", "float32", options={"precision": "float64"})


@pytest.mark.parametrize("lmax,U,UF", [(1, 16, 32), (2, 24, 32), (2, 40, 64)])
def test_zero_padded_model_is_the_same_model(emu_lib, model_dir, lmax, U, UF):
    """csrc/model_io.cpp: pad_host_model -- what the fused kernels run for a model narrower than their fixed widths (S -> 64 scalars, U -> 32 / 64 tensor features with the (l, u) columns
    moved, MLP width -> 64, read-out width -> 32).  Here on the CPU: the same file evaluated by the emulated layer-at-a-time float64 kernels as it is and after padding (a test-only hook
    of the emulation library swaps the padded copy in before the first evaluation): forces, energies and virial agree to round-off, i.e. padding adds exact zeros and nothing else."""
    import ctypes as C
    from oracle import allegro_torch
    from pair_allegro_amd import lmp_like, model_file
    from pair_allegro_amd.pair import PairAllegro, atom_from_rank_system, list_from_rank_system
    g = util.load_golden("Cu2AgO4_r5")
    names = ["Ag", "Cu", "O"]
    cfg = dict(model_file.DEFAULT_CFG, model_dtype="float64", type_names=names, l_max=lmax, num_layers=3 if lmax == 2 else 2, num_scalar_features=48, num_tensor_features=U,
               mlp_width=40, readout_width=24, avg_num_neighbors=30.0)
    w = model_file.init_weights(cfg)
    path = f"{model_dir}/pad_l{lmax}_U{U}.nequip.pth"
    allegro_torch.export_nequip_pth(path, cfg, w)
    types = np.array([names.index(s) + 1 for s in g["symbols"]], dtype=np.int32)
    rs = lmp_like.build_rank_system(g["cell"], g["pos"], types, cfg["r_max"] + 1.0)
    out = []
    for padded in (False, True):
        pair = PairAllegro(lib=emu_lib, quiet=True)
        pair.settings([])
        pair.coeff(["*", "*", path] + names, ntypes=3)
        if padded:
            emu_lib.lib.ahip_emu_pad_model.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
            assert emu_lib.lib.ahip_emu_pad_model(pair.model.h, 64, UF, 64, 32) == 0
        atom = atom_from_rank_system(rs, 3)
        pair.compute(atom, list_from_rank_system(rs))
        out.append((atom.f.copy(), pair.eatom[: rs.nlocal].copy(), pair.eng_vdwl, np.array(pair.virial)))
        pair.model.close()
    (f0, e0, pe0, v0), (f1, e1, pe1, v1) = out
    assert np.abs(f0).max() > 1e-3
    np.testing.assert_allclose(f1, f0, rtol=0, atol=1e-12 * max(1.0, np.abs(f0).max()))
    np.testing.assert_allclose(e1, e0, rtol=0, atol=1e-12 * max(1.0, np.abs(e0).max()))
    np.testing.assert_allclose(pe1, pe0, rtol=1e-13)
    np.testing.assert_allclose(v1, v0, rtol=0, atol=1e-11 * max(1.0, np.abs(v0).max()))
