"""Angular conventions: real SH normalisation, CG tables pinned against sympy, equivariance."""
import numpy as np
import pytest

from pair_allegro_amd import cg


def test_sh_component_normalisation():
    rng = np.random.default_rng(0)
    n = rng.normal(size=(200, 3))
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    Y = cg.real_sh(n, 3)
    for l in range(4):
        np.testing.assert_allclose((Y[:, l * l:(l + 1) ** 2] ** 2).sum(1), 2 * l + 1, rtol=1e-12)


def _complex_to_real(l):
    M = np.zeros((2 * l + 1, 2 * l + 1), complex)
    for m in range(-l, l + 1):
        i = m + l
        if m < 0:
            M[i, l + m] = 1j / np.sqrt(2); M[i, l - m] = -1j * (-1) ** m / np.sqrt(2)
        elif m == 0:
            M[i, l] = 1
        else:
            M[i, l - m] = 1 / np.sqrt(2); M[i, l + m] = (-1) ** m / np.sqrt(2)
    return M


@pytest.mark.parametrize("path", cg.tp_paths(3))
def test_w3j_matches_sympy_up_to_sign(path):
    from sympy.physics.wigner import wigner_3j
    l1, l2, l3 = path
    W = np.zeros((2 * l1 + 1, 2 * l2 + 1, 2 * l3 + 1))
    for a in range(-l1, l1 + 1):
        for b in range(-l2, l2 + 1):
            for c in range(-l3, l3 + 1):
                W[a + l1, b + l2, c + l3] = float(wigner_3j(l1, l2, l3, a, b, c))
    R = np.einsum("ia,jb,kc,abc->ijk", _complex_to_real(l1).conj(), _complex_to_real(l2).conj(),
                  _complex_to_real(l3).conj(), W.astype(complex))
    R = R.real if np.abs(R.real).max() > np.abs(R.imag).max() else R.imag
    R /= np.linalg.norm(R)
    mine = cg.real_w3j(l1, l2, l3)
    assert min(np.abs(R - mine).max(), np.abs(R + mine).max()) < 1e-12


def test_tensor_product_is_equivariant():
    rng = np.random.default_rng(3)
    a, b, c = (v / np.linalg.norm(v) for v in rng.normal(size=(3, 3)))
    R = cg._random_rotation(rng)
    for (l1, l2, l3) in cg.tp_paths(3):
        C = cg.path_coeff(l1, l2, l3)
        s = lambda v, l: cg.real_sh(v[None], 3)[0, l * l:(l + 1) ** 2]
        inv = np.einsum("abc,a,b,c->", C, s(a, l1), s(b, l2), s(c, l3))
        inv_r = np.einsum("abc,a,b,c->", C, s(R @ a, l1), s(R @ b, l2), s(R @ c, l3))
        assert abs(inv - inv_r) < 1e-12


def test_generated_header_is_current():
    import os
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    txt = open(os.path.join(here, "pair_allegro_amd", "csrc", "cg_tables.h")).read()
    for lmax in (1, 2, 3):
        ent = cg.sparse_path_entries(lmax)
        assert f"#define AHIP_CG_L{lmax}_N {len(ent)}" in txt
        for (p, i1, i2, i3, c) in ent[:5] + ent[-5:]:
            assert f"{{{p}, {i1}, {i2}, {i3}, {c!r}}}" in txt
