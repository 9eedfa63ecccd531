"""Real spherical harmonics and real Clebsch-Gordan (Wigner-3j) tables for l <= 2.

This is the *specification* of the angular conventions of the allegro-hip model
(DESIGN.md "Model spec"); the HIP kernels use the header generated from it
(``tools/gen_cg.py`` -> ``csrc/cg_tables.h``) and the torch oracle imports it.

The reference repository contains none of this arithmetic: the model lives in an
opaque TorchScript file (/root/reference/pair_nequip_allegro.cpp:222,425), the only
in-repo evidence being the hyper-parameter names in
/root/reference/tests/test_data/test_repro_allegro.yaml:80-103 (``l_max`` ...).

Conventions
-----------
* Real SH, "component" normalisation (sum_m Y_lm(n)^2 = 2l+1 on the unit sphere),
  m ordered -l..l:
    l=0: 1
    l=1: sqrt(3) (y, z, x)
    l=2: sqrt(15) xy, sqrt(15) yz, sqrt(5)/2 (2z^2-x^2-y^2), sqrt(15) xz, sqrt(15)/2 (x^2-y^2)
* w3j(l1,l2,l3)[m1,m2,m3]: the unique rotation-invariant 3-tensor of the real irreps,
  Frobenius norm 1, sign fixed so that the first non-zero entry in lexicographic
  (m1,m2,m3) order is positive.  It is derived *numerically* here (null space of the
  invariance constraint under random rotations) so the derivation is independent of
  sympy; tests/test_cg.py pins it against sympy.physics.wigner up to that sign.
* Tensor-product path coefficient: C_path = sqrt(2 l3 + 1) * w3j(l1,l2,l3).
* Paths are enumerated l3-major: for l3, for l1, for l2 with |l1-l2| <= l3 <= l1+l2,
  all three <= l_max ("so3" parity setting: no parity selection rule).
"""
from __future__ import annotations

import functools
from typing import List, Tuple

import numpy as np

SQ3 = np.sqrt(3.0)
SQ5 = np.sqrt(5.0)
SQ15 = np.sqrt(15.0)

SQ7 = np.sqrt(7.0)
SQ42 = np.sqrt(42.0)
SQ70 = np.sqrt(70.0)
SQ105 = np.sqrt(105.0)

LMAX_SUPPORTED = 3


def sh_dim(lmax: int) -> int:
    return (lmax + 1) ** 2


def l_of_index(lmax: int) -> List[int]:
    """l value of every flattened (l,m) component."""
    out: List[int] = []
    for l in range(lmax + 1):
        out += [l] * (2 * l + 1)
    return out


def real_sh(n: np.ndarray, lmax: int) -> np.ndarray:
    """Y[..., (lmax+1)^2] for unit vectors n[..., 3] (x,y,z)."""
    assert 0 <= lmax <= LMAX_SUPPORTED
    x, y, z = n[..., 0], n[..., 1], n[..., 2]
    cols = [np.ones_like(x)]
    if lmax >= 1:
        cols += [SQ3 * y, SQ3 * z, SQ3 * x]
    if lmax >= 2:
        cols += [
            SQ15 * x * y,
            SQ15 * y * z,
            0.5 * SQ5 * (2 * z * z - x * x - y * y),
            SQ15 * x * z,
            0.5 * SQ15 * (x * x - y * y),
        ]
    if lmax >= 3:
        # l = 3, m = -3 .. 3, same conventions (z the polar axis, component normalisation: the mean square of every component over the sphere is 1);
        # written as homogeneous cubics (r = 1), which is the form the kernels differentiate (generic_kernels.h: sh_grad_dot)
        cols += [
            0.25 * SQ70 * y * (3 * x * x - y * y),
            SQ105 * x * y * z,
            0.25 * SQ42 * y * (4 * z * z - x * x - y * y),
            0.5 * SQ7 * z * (2 * z * z - 3 * x * x - 3 * y * y),
            0.25 * SQ42 * x * (4 * z * z - x * x - y * y),
            0.5 * SQ105 * z * (x * x - y * y),
            0.25 * SQ70 * x * (x * x - 3 * y * y),
        ]
    return np.stack(cols, axis=-1)


def _random_rotation(rng: np.random.Generator) -> np.ndarray:
    q, r = np.linalg.qr(rng.normal(size=(3, 3)))
    q = q * np.sign(np.diag(r))
    if np.linalg.det(q) < 0:
        q[:, 0] = -q[:, 0]
    return q


def wigner_D_real(l: int, R: np.ndarray) -> np.ndarray:
    """D with Y_l(R n) = D @ Y_l(n), obtained by exact least squares."""
    rng = np.random.default_rng(1234 + l)
    pts = rng.normal(size=(64, 3))
    pts /= np.linalg.norm(pts, axis=1, keepdims=True)
    sl = slice(l * l, (l + 1) * (l + 1))
    A = real_sh(pts, LMAX_SUPPORTED)[:, sl]            # [P, 2l+1]
    B = real_sh(pts @ R.T, LMAX_SUPPORTED)[:, sl]      # [P, 2l+1]
    Dt, *_ = np.linalg.lstsq(A, B, rcond=None)          # A @ Dt = B  => D = Dt.T
    return Dt.T


@functools.lru_cache(maxsize=None)
def real_w3j(l1: int, l2: int, l3: int) -> np.ndarray:
    """Rotation-invariant tensor [2l1+1, 2l2+1, 2l3+1], Frobenius norm 1."""
    assert abs(l1 - l2) <= l3 <= l1 + l2
    rng = np.random.default_rng(99)
    d1, d2, d3 = 2 * l1 + 1, 2 * l2 + 1, 2 * l3 + 1
    rows = []
    for _ in range(4):
        R = _random_rotation(rng)
        K = np.kron(np.kron(wigner_D_real(l1, R), wigner_D_real(l2, R)), wigner_D_real(l3, R))
        rows.append(K - np.eye(d1 * d2 * d3))
    M = np.concatenate(rows, axis=0)
    _, s, vt = np.linalg.svd(M)
    assert s[-1] < 1e-10 and (len(s) < 2 or s[-2] > 1e-6), (l1, l2, l3, s[-3:])
    c = vt[-1].reshape(d1, d2, d3)
    c[np.abs(c) < 1e-12] = 0.0
    c /= np.linalg.norm(c)
    first = c.flat[np.flatnonzero(c)[0]]
    if first < 0:
        c = -c
    return c


def tp_paths(lmax: int, scalar_only: bool = False) -> List[Tuple[int, int, int]]:
    """(l1, l2, l3) paths; l1 indexes the edge tensor V, l2 the centre environment."""
    paths = []
    for l3 in range(lmax + 1):
        if scalar_only and l3 > 0:
            break
        for l1 in range(lmax + 1):
            for l2 in range(lmax + 1):
                if abs(l1 - l2) <= l3 <= l1 + l2:
                    paths.append((l1, l2, l3))
    return paths


def path_coeff(l1: int, l2: int, l3: int) -> np.ndarray:
    """C_path[m1,m2,m3] = sqrt(2 l3+1) w3j."""
    return np.sqrt(2 * l3 + 1.0) * real_w3j(l1, l2, l3)


def sparse_path_entries(lmax: int, scalar_only: bool = False):
    """List of (path_index, i1, i2, i3, coeff) with i* flattened (l,m) indices."""
    out = []
    for p, (l1, l2, l3) in enumerate(tp_paths(lmax, scalar_only)):
        c = path_coeff(l1, l2, l3)
        for m1 in range(2 * l1 + 1):
            for m2 in range(2 * l2 + 1):
                for m3 in range(2 * l3 + 1):
                    if c[m1, m2, m3] != 0.0:
                        out.append((p, l1 * l1 + m1, l2 * l2 + m2, l3 * l3 + m3, float(c[m1, m2, m3])))
    return out
