"""GPU tests of the wide fused kernels (fused_lx.hip): l_max = 2 with 32 tensor features -- the model shape of the reference's
own test YAML (/root/reference/tests/test_data/test_repro_allegro.yaml:89-99), which is also the shape of the committed golden
fixtures -- and with 64 tensor features (BASELINE config 5's model L), against the float64 oracle."""
import numpy as np
import pytest

import parity_cases as pc
import util
from oracle import allegro_torch
from pair_allegro_amd import lmp_like, model_file

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag", ["Cu-cubic_r5", "Cu2AgO4_r5", "aspirin_r5", "aspirin_r15", "CuPd-cubic-big_r5", "water_192_r5"])
def test_golden_yaml_shape_on_the_fused_kernel(hip_lib, model_dir, tag):
    """The reference's four test geometries with its YAML's model shape (l_max 2, 32 tensor features, 3 layers) and the 192-atom
    water sample of BASELINE config 5 with model L (64 tensor features): the default path is the fused kernel; forces, per-atom
    energies, PE and virial against the committed float64 goldens."""
    res, g = pc.check_golden(hip_lib, model_dir, tag, "float32")
    assert res["info"]["path"] in pc.FUSED_F32EQ
    pc.check_edges_vs_brute_force(res, g)
    gen = util.run_pair(hip_lib, util.golden_model(g, model_dir, "float32")[0], g["cell"], g["pos"], *util.lammps_types(g),
                        options={"path": "generic"})
    np.testing.assert_allclose(res["forces"], gen["forces"], atol=5e-5)


@pytest.mark.parametrize("grid", [(2, 1, 1), (2, 2, 1)])
def test_golden_yaml_shape_multi_rank(hip_lib, model_dir, grid):
    res, g = pc.check_golden(hip_lib, model_dir, "CuPd-cubic-big_r5", "float32", grid=grid)
    assert res["info"]["path"] in pc.FUSED_F32EQ


def _case(model_dir, name, cfg, cell, pos, symbols):
    w = model_file.init_weights(cfg)
    path = f"{model_dir}/{name}.nequip.pth"
    allegro_torch.export_nequip_pth(path, cfg, w)
    names = sorted(set(symbols))
    types = np.array([names.index(s) + 1 for s in symbols], dtype=np.int32)
    ref = util.oracle_run(dict(cfg, model_dtype="float64"), w, cell, pos, types, names)
    return path, types, names, ref


@pytest.mark.parametrize("U,nb,p", [(32, 6, 6), (64, 10, 5)])
def test_wide_kernels_take_any_radial_basis(hip_lib, model_dir, U, nb, p):
    """`num_bessels` / `polynomial_cutoff_p` other than the YAML's 8 / 6 (test_repro_allegro.yaml:83-86) stay on the wide fused kernels: the radial
    basis only enters through the tabulated two-body embedding, built by the host from the model's own Bessel block (round 4)."""
    g = util.load_golden("Cu2AgO4_r5")
    cfg = model_file.model_L(type_names=["Cu", "Ag", "O"], num_tensor_features=U, num_bessels=nb, poly_p=p, avg_num_neighbors=30.0)
    path, types, names, ref = _case(model_dir, f"L_nb{nb}_U{U}", cfg, g["cell"], g["pos"], g["symbols"])
    res = util.run_pair(hip_lib, path, g["cell"], g["pos"], types, names, options={"path": "fused"})
    assert res["info"]["path"] in pc.FUSED_F32EQ
    util.assert_close_to(res, ref, 5e-4, what=f"U={U}, {nb} Bessel functions, p={p}")
    assert np.abs(res["forces"] - ref["forces"]).max() < pc.NORTH_STAR_DF


@pytest.mark.parametrize("U", [32, 64])
@pytest.mark.parametrize("nl", [1, 2, 3])
def test_model_L_layers_and_widths(hip_lib, model_dir, nl, U):
    """CuPd 256-atom box relabelled O/H: every layer count (the kernels are instantiated per layer count) and both widths."""
    g = util.load_golden("CuPd-cubic-big_r5")
    symbols = ["O" if s == "Cu" else "H" for s in g["symbols"]]
    nb = float(len(util.glue.brute_force_edges(g["cell"], g["pos"], 5.0)[0])) / len(g["pos"])
    cfg = model_file.model_L(avg_num_neighbors=nb, num_layers=nl, num_tensor_features=U)
    path, types, names, ref = _case(model_dir, f"cupd_L_{nl}_{U}", cfg, g["cell"], g["pos"], symbols)
    res = util.run_pair(hip_lib, path, g["cell"], g["pos"], types, names, options={"path": "fused"})
    assert res["info"]["path"] in pc.FUSED_F32EQ
    util.assert_close_to(res, ref, 5e-4, what=f"model L nl={nl} U={U} fused vs f64 oracle")
    assert np.abs(res["forces"] - ref["forces"]).max() < pc.NORTH_STAR_DF


def test_model_L_three_types_ragged(hip_lib, model_dir):
    """Cu2AgO4 (7 atoms, triclinic, 3 types, ragged degrees) with model L's widths: partial tiles, 9 type-pair tables."""
    g = util.load_golden("Cu2AgO4_r5")
    nb = float(g["nedges"]) / len(g["pos"])
    cfg = model_file.model_L(type_names=["Cu", "Ag", "O"], avg_num_neighbors=nb)
    path, types, names, ref = _case(model_dir, "cu2ago4_L", cfg, g["cell"], g["pos"], g["symbols"])
    res = util.run_pair(hip_lib, path, g["cell"], g["pos"], types, names)
    assert res["info"]["path"] in pc.FUSED_F32EQ
    util.assert_close_to(res, ref, 5e-4, what="Cu2AgO4 model L")
    assert np.abs(res["forces"] - ref["forces"]).max() < pc.NORTH_STAR_DF


def test_model_L_centres_with_more_than_64_edges(hip_lib, model_dir):
    """A water box where a few centres exceed the 64 edge slots of a tile: those centres go through the layer-at-a-time kernels
    (heavy_generic), everything else through the fused kernel; the sum must equal the all-generic evaluation, and a box where
    EVERY centre is too large falls back to the generic path as a whole."""
    cell, pos, types = lmp_like.water(14)
    cfg = model_file.model_L(avg_num_neighbors=53.6)
    path = f"{model_dir}/water_L14.ahip"
    model_file.save_ahip(path, cfg, model_file.init_weights(cfg))
    gen = util.run_pair(hip_lib, path, cell, pos, types, ["O", "H"], options={"path": "generic"})
    res = util.run_pair(hip_lib, path, cell, pos, types, ["O", "H"])
    assert res["info"]["path"] in pc.FUSED_F32EQ and res["info"]["max_degree"] > 64
    assert np.abs(res["forces"] - gen["forces"]).max() < 5e-5
    np.testing.assert_allclose(res["eatom"], gen["eatom"], atol=5e-5)
    np.testing.assert_allclose(res["pe"], gen["pe"], rtol=2e-6)
    np.testing.assert_allclose(res["virial"], gen["virial"], atol=2e-3 * len(pos) ** 0.5, rtol=1e-4)
    g = util.load_golden("Cu-cubic_r15")
    cfg15 = model_file.model_L(type_names=["Cu"], r_max=6.1, avg_num_neighbors=78.0)
    path15 = f"{model_dir}/cu61_L.ahip"
    model_file.save_ahip(path15, cfg15, model_file.init_weights(cfg15))
    reps = 3
    cell3 = g["cell"] * reps
    shifts = np.array([[i, j, k] for i in range(reps) for j in range(reps) for k in range(reps)], dtype=float)
    pos3 = np.concatenate([g["pos"] + s @ g["cell"] for s in shifts])
    all_heavy = util.run_pair(hip_lib, path15, cell3, pos3, np.ones(len(pos3), np.int32), ["Cu"])
    assert all_heavy["info"]["path"] == "generic_f32"


def test_model_L_six_species(hip_lib, model_dir):
    """Model L (l_max 2, 64 tensor features: the wave-pair kernel) with 6 model types."""
    g = util.load_golden("CuPd-cubic-big_r5")
    names = ["A", "B", "C", "D", "E", "F"]
    rng = np.random.RandomState(6)
    symbols = [names[k] for k in rng.randint(0, 6, size=len(g["pos"]))]
    cfg = model_file.model_L(type_names=names, avg_num_neighbors=42.0)
    path, types, lnames, ref = _case(model_dir, "six_species_L", cfg, g["cell"], g["pos"], symbols)
    res = util.run_pair(hip_lib, path, g["cell"], g["pos"], types, lnames)
    assert res["info"]["path"] in pc.FUSED_F32EQ
    util.assert_close_to(res, ref, 5e-4, what="6 species model L")
    assert np.abs(res["forces"] - ref["forces"]).max() < pc.NORTH_STAR_DF


@pytest.mark.parametrize("U,S,W,R", [(16, 48, 40, 24), (48, 64, 48, 32), (24, 32, 64, 16)])
def test_narrower_l2_models_run_on_the_wide_kernels_zero_padded(hip_lib, model_dir, U, S, W, R):
    """l_max = 2 models narrower than the wide kernels' fixed widths run on them zero-padded (model_io.cpp: pad_host_model): up to 32 tensor features on k_fused_lx, 33..64 on
    k_fused_lx2; any S / MLP width <= 64, read-out width <= 32 (free hyper-parameters of /root/reference/tests/test_data/test_repro_allegro.yaml:89-99).  Cu2AgO4 (3 types, ragged)."""
    g = util.load_golden("Cu2AgO4_r5")
    cfg = model_file.model_L(type_names=["Cu", "Ag", "O"], num_tensor_features=U, num_scalar_features=S, mlp_width=W, readout_width=R, avg_num_neighbors=30.0)
    path, types, names, ref = _case(model_dir, f"L_narrow_U{U}_S{S}_W{W}_R{R}", cfg, g["cell"], g["pos"], g["symbols"])
    res = util.run_pair(hip_lib, path, g["cell"], g["pos"], types, names)
    assert res["info"]["path"] == pc.FUSED_DEFAULT, res["info"]
    util.assert_close_to(res, ref, 5e-4, what=f"narrow l_max = 2 model U={U} S={S} W={W} R={R}")
    assert np.abs(res["forces"] - ref["forces"]).max() < pc.NORTH_STAR_DF
    gen = util.run_pair(hip_lib, path, g["cell"], g["pos"], types, names, options={"path": "generic"})
    np.testing.assert_allclose(res["forces"], gen["forces"], atol=2e-5)
