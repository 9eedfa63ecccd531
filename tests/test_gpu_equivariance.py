"""Oracle-INDEPENDENT parity property of the HIP kernels (-m gpu): the model is E(3)-equivariant by construction (real spherical harmonics, Clebsch-Gordan
tensor products, scalar read-out), so rotating + translating + relabelling a structure must leave the energies unchanged, rotate the forces and
conjugate the virial -- whatever the weights are.  A convention error in the angular tables, the tensor product, its hand-derived gradient or the
force / virial assembly breaks this on the device without any reference being involved; the oracle comparison of the other tests cannot see an error
that oracle and kernels share, this one can see every error that is not itself rotation-covariant.
Runs on all three fused kernels (k_fused: l_max 1; k_fused_lx: l_max 2 / 32 features = the shape of /root/reference/tests/test_data/test_repro_allegro.yaml:89-99;
k_fused_lx2: l_max 2 / 64 features) and on the layer-at-a-time kernels, on the reference's aspirin geometry (non-periodic -> 50 A box, tests/conftest.py:186-190
of the reference), float32 arithmetic: tolerance 2e-5 on forces of magnitude ~1."""
import numpy as np
import pytest

import util
import parity_cases as pc  # noqa: E402
from oracle import allegro_torch
from pair_allegro_amd import cg, model_file

pytestmark = pytest.mark.gpu


def _cfg(kind):
    tn = ["C", "H", "O"]
    if kind == "S":
        return model_file.model_S(type_names=tn, avg_num_neighbors=20.0, num_layers=3)
    if kind == "Y":
        return model_file.model_L(type_names=tn, num_tensor_features=32, avg_num_neighbors=20.0)
    return model_file.model_L(type_names=tn, avg_num_neighbors=20.0)


@pytest.mark.parametrize("path_opt", ["fused", "generic"])
@pytest.mark.parametrize("kind", ["S", "Y", "L"])
def test_rotation_translation_permutation(hip_lib, model_dir, kind, path_opt):
    g = util.load_golden("aspirin_r5")
    cfg = _cfg(kind)
    w = model_file.init_weights(cfg)
    path = f"{model_dir}/equiv_{kind}.nequip.pth"
    allegro_torch.export_nequip_pth(path, cfg, w)
    names = ["C", "H", "O"]
    types = np.array([names.index(s) + 1 for s in g["symbols"]], dtype=np.int32)
    pos, cell = np.asarray(g["pos"]), np.asarray(g["cell"])
    a = util.run_pair(hip_lib, path, cell, pos, types, names, options={"path": path_opt})
    assert a["info"]["path"] in (pc.FUSED_F32EQ if path_opt == "fused" else ("generic_f32",))
    rng = np.random.default_rng(5)
    centre = pos.mean(axis=0)
    fmax = np.abs(a["forces"]).max()
    assert fmax > 1e-2
    for trial in range(3):
        R = cg._random_rotation(rng)
        shift = rng.uniform(-8.0, 8.0, size=3)                  # stays well inside the 50 A box: no periodic image within reach
        perm = rng.permutation(len(pos))
        pos2 = ((pos - centre) @ R.T + centre + shift)[perm]
        b = util.run_pair(hip_lib, path, cell, pos2, types[perm], names, options={"path": path_opt})
        tol = 2e-5 * max(1.0, fmax)
        np.testing.assert_allclose(b["eatom"], a["eatom"][perm], atol=2e-5, err_msg=f"{kind}/{path_opt}: per-atom energies are not invariant")
        np.testing.assert_allclose(b["pe"], a["pe"], atol=1e-4)
        np.testing.assert_allclose(b["forces"], (a["forces"] @ R.T)[perm], atol=tol, err_msg=f"{kind}/{path_opt}: forces do not rotate with the structure")
        # virial in LAMMPS order xx yy zz xy xz yz -> matrix, conjugated by R
        def mat(v):
            return np.array([[v[0], v[3], v[4]], [v[3], v[1], v[5]], [v[4], v[5], v[2]]])
        np.testing.assert_allclose(mat(b["virial"]), R @ mat(a["virial"]) @ R.T, atol=2e-4, err_msg=f"{kind}/{path_opt}: virial does not transform as a tensor")
    assert abs(a["forces"].sum(axis=0)).max() < 1e-4             # no net force on an isolated molecule
