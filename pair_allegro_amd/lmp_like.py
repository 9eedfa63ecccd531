"""What LAMMPS hands to a pair style, rebuilt on the host with numpy/scipy (test + driver plumbing).

LAMMPS itself is not part of the reference repository ([EXT] in SURVEY.md); the pair style only
sees its products: ``atom->x/type/tag`` for nlocal owned + nghost ghost atoms and a *full*,
skin-inflated neighbor list over the owned atoms (REQ_FULL | REQ_GHOST request,
/root/reference/pair_nequip_allegro.cpp:143-147; consumed :469-480).  This module produces exactly
those arrays for a periodic (possibly triclinic) cell and a brick decomposition over a px*py*pz
processor grid, the way LAMMPS does: ghosts are every periodic image / foreign atom whose
fractional coordinate lies within cut/height of the sub-domain brick.
"""
from __future__ import annotations

import itertools
from dataclasses import dataclass, field
from typing import List, Sequence, Tuple

import numpy as np
from scipy.spatial import cKDTree

NEIGHMASK = 0x1FFFFFFF


@dataclass
class RankSystem:
    """Per-rank LAMMPS view."""
    nlocal: int
    nghost: int
    x: np.ndarray          # [nall,3] f64, locals first
    type: np.ndarray       # [nall] i32, 1-based LAMMPS types
    tag: np.ndarray        # [nall] i32, 1-based global ids
    ilist: np.ndarray      # [inum] i32
    numneigh: np.ndarray   # [nall] i32 (by atom index; zero for ghosts)
    firstneigh: List[np.ndarray] = field(default_factory=list)   # by atom index
    offsets: np.ndarray = None     # CSR form over ilist
    flat: np.ndarray = None

    @property
    def nall(self) -> int:
        return self.nlocal + self.nghost


def cell_heights(cell: np.ndarray) -> np.ndarray:
    vol = abs(np.linalg.det(cell))
    a, b, c = cell
    return np.array([vol / np.linalg.norm(np.cross(b, c)), vol / np.linalg.norm(np.cross(c, a)),
                     vol / np.linalg.norm(np.cross(a, b))])


def wrap(cell: np.ndarray, pos: np.ndarray) -> np.ndarray:
    frac = pos @ np.linalg.inv(cell)
    frac -= np.floor(frac)
    return frac @ cell


def build_rank_system(cell: np.ndarray, pos: np.ndarray, types: Sequence[int], rc_list: float,
                      grid: Tuple[int, int, int] = (1, 1, 1), rank: Tuple[int, int, int] = (0, 0, 0),
                      sort_neighbors: bool = True) -> RankSystem:
    """cell rows are lattice vectors; pos must be wrapped into the cell; types 1-based."""
    cell = np.asarray(cell, dtype=np.float64)
    pos = np.asarray(pos, dtype=np.float64)
    types = np.asarray(types, dtype=np.int32)
    n = len(pos)
    inv = np.linalg.inv(cell)
    frac = pos @ inv
    frac = np.where(frac >= 1.0, frac - 1.0, frac)
    lo = np.array([rank[k] / grid[k] for k in range(3)])
    hi = np.array([(rank[k] + 1) / grid[k] for k in range(3)])
    own = np.all((frac >= lo) & (frac < hi), axis=1)
    own_idx = np.flatnonzero(own)
    margin = rc_list / cell_heights(cell)                      # fractional ghost shell per direction
    nimg = np.ceil(margin + (hi - lo) * 0 ).astype(int)
    shifts = list(itertools.product(*[range(-int(nimg[k]) - 1, int(nimg[k]) + 2) for k in range(3)]))
    gx, gt, gtag = [], [], []
    for s in shifts:
        sf = frac + np.asarray(s, dtype=np.float64)
        inside = np.all((sf >= lo - margin) & (sf <= hi + margin), axis=1)
        if s == (0, 0, 0):
            inside &= ~own
        idx = np.flatnonzero(inside)
        if len(idx):
            gx.append(sf[idx] @ cell)
            gt.append(types[idx])
            gtag.append(idx + 1)
    nlocal = len(own_idx)
    xl = frac[own_idx] @ cell
    if gx:
        x = np.concatenate([xl] + gx)
        ty = np.concatenate([types[own_idx]] + gt)
        tag = np.concatenate([own_idx + 1] + gtag)
    else:
        x, ty, tag = xl, types[own_idx], own_idx + 1
    nall = len(x)
    tree = cKDTree(x)
    rows = tree.query_ball_point(x[:nlocal], rc_list) if nlocal else []
    first: List[np.ndarray] = []
    numneigh = np.zeros(nall, dtype=np.int32)
    for i in range(nlocal):
        r = np.asarray(rows[i], dtype=np.int32)
        r = r[r != i]
        if sort_neighbors:
            r.sort()
        first.append(r)
        numneigh[i] = len(r)
    for _ in range(nall - nlocal):
        first.append(np.zeros(0, dtype=np.int32))
    ilist = np.arange(nlocal, dtype=np.int32)
    offsets = np.zeros(nlocal + 1, dtype=np.int64)
    if nlocal:
        offsets[1:] = np.cumsum(numneigh[:nlocal])
    flat = np.concatenate(first[:nlocal]) if nlocal and offsets[-1] > 0 else np.zeros(0, dtype=np.int32)
    return RankSystem(nlocal=nlocal, nghost=nall - nlocal, x=np.ascontiguousarray(x), type=ty.astype(np.int32),
                      tag=tag.astype(np.int32), ilist=ilist, numneigh=numneigh, firstneigh=first,
                      offsets=offsets, flat=flat.astype(np.int32))


def grid_ranks(grid: Tuple[int, int, int]):
    return list(itertools.product(range(grid[0]), range(grid[1]), range(grid[2])))


def diamond_si(ncell: int, a: float = 5.431, jitter: float = 0.05, seed: int = 0):
    """BASELINE.md section 4 workloads: Si diamond, ncell^3 cells, Gaussian jitter, wrapped."""
    basis = np.array([[0, 0, 0], [0, .5, .5], [.5, 0, .5], [.5, .5, 0],
                      [.25, .25, .25], [.25, .75, .75], [.75, .25, .75], [.75, .75, .25]])
    g = np.stack(np.meshgrid(np.arange(ncell), np.arange(ncell), np.arange(ncell), indexing="ij"), -1).reshape(-1, 3)
    pos = (g[:, None, :] + basis[None, :, :]).reshape(-1, 3) * a
    rng = np.random.RandomState(seed)
    pos = pos + rng.normal(0.0, jitter, size=pos.shape)
    cell = np.eye(3) * (ncell * a)
    return cell, wrap(cell, pos), np.ones(len(pos), dtype=np.int32)


# ---- BASELINE configs 3 and 5: synthetic but physically dense multi-species boxes -----------------------
# gamma-Li3PO4 (Pnma, Z = 4, 32 atoms): a = 10.49, b = 6.12, c = 4.93 A -> 0.1011 atoms/A^3, ~53 neighbours at 5 A.
# Fractional coordinates of the asymmetric unit (approximate literature values; the workload only needs the density,
# the 4-species topology and sane nearest-neighbour distances).
_LI3PO4_ABC = (10.49, 6.12, 4.93)
_LI3PO4_SITES = [            # (LAMMPS type, Wyckoff, x, y, z): types 1 Li, 2 P, 3 O1 (8d), 4 O2 (4c + 4c)
    (1, "8d", 0.1639, 0.5013, 0.3013),
    (1, "4c", 0.4237, 0.75, 0.2056),
    (2, "4c", 0.4115, 0.25, 0.3088),
    (3, "8d", 0.3416, 0.0426, 0.2054),
    (4, "4c", 0.0507, 0.25, 0.2937),
    (4, "4c", 0.0895, 0.75, 0.1223),
]
LI3PO4_LAMMPS_NAMES = ["Li", "P", "O", "O"]          # deck: pair_coeff * * <file> Li P O O (two LAMMPS types share model type O)
LI3PO4_MASSES = {"Li": 6.94, "P": 30.974, "O": 15.999}


def _pnma_images(w, x, y, z):
    if w == "4c":                                       # mirror plane y = 1/4, 3/4
        return [(x, y, z), (-x + 0.5, y + 0.5, z + 0.5), (-x, y + 0.5, -z), (x + 0.5, y, -z + 0.5)]
    g = [(x, y, z), (-x + 0.5, -y, z + 0.5), (-x, y + 0.5, -z), (x + 0.5, -y + 0.5, -z + 0.5)]
    return g + [(-a, -b, -c) for a, b, c in g]


def li3po4(reps=(10, 16, 20), jitter: float = 0.05, seed: int = 0):
    """BASELINE config 3: (10,16,20) cells = 102 400 atoms in a 104.9 x 97.9 x 98.6 A box, LAMMPS types Li P O1 O2."""
    frac, types = [], []
    for t, w, x, y, z in _LI3PO4_SITES:
        for p in _pnma_images(w, x, y, z):
            frac.append(p); types.append(t)
    frac = np.mod(np.asarray(frac, dtype=np.float64), 1.0)
    types = np.asarray(types, dtype=np.int32)
    abc = np.asarray(_LI3PO4_ABC)
    nx, ny, nz = reps
    g = np.stack(np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij"), -1).reshape(-1, 3)
    pos = ((g[:, None, :] + frac[None, :, :]).reshape(-1, 3)) * abc
    ty = np.tile(types, len(g))
    rng = np.random.RandomState(seed)
    pos = pos + rng.normal(0.0, jitter, size=pos.shape)
    cell = np.diag(abc * np.asarray(reps, dtype=np.float64))
    return cell, wrap(cell, pos), ty


WATER_MASSES = {"O": 15.999, "H": 1.008}


def water(m: int = 55, density: float = 0.1003, jitter: float = 0.3, seed: int = 0, min_contact: float = 1.25):
    """BASELINE config 5: m^3 rigid-geometry H2O molecules (m = 55 -> 499 125 atoms), 0.1003 atoms/A^3 (~53 neighbours at
    5 A): oxygens on a jittered simple-cubic lattice, uniformly random orientations; molecules with an intermolecular
    contact shorter than `min_contact` get a new orientation until none is left (deterministic in `seed`).  LAMMPS types
    1 = O, 2 = H; atoms are ordered molecule by molecule (O H H), as a LAMMPS data file would list them."""
    nmol = m ** 3
    box = (3.0 * nmol / density) ** (1.0 / 3.0)
    a = box / m
    rng = np.random.RandomState(seed)
    g = np.stack(np.meshgrid(np.arange(m), np.arange(m), np.arange(m), indexing="ij"), -1).reshape(-1, 3)
    O = (g + 0.5) * a + rng.normal(0.0, jitter, size=(nmol, 3))
    th, r = np.deg2rad(104.52), 0.9572
    h1 = np.array([r * np.sin(th / 2), 0.0, r * np.cos(th / 2)])
    h2 = np.array([-r * np.sin(th / 2), 0.0, r * np.cos(th / 2)])

    def rotations(k):
        q = rng.normal(size=(k, 4))
        q /= np.linalg.norm(q, axis=1, keepdims=True)
        w, x, y, z = q.T
        return np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                         2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                         2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], axis=1).reshape(k, 3, 3)

    R = rotations(nmol)
    cell = np.eye(3) * box
    for _ in range(200):
        pos = wrap(cell, np.stack([O, O + R @ h1, O + R @ h2], axis=1).reshape(-1, 3))
        pairs = cKDTree(pos, boxsize=box).query_pairs(min_contact, output_type="ndarray")
        pairs = pairs[pairs[:, 0] // 3 != pairs[:, 1] // 3]
        if len(pairs) == 0:
            break
        bad = np.unique(pairs // 3)
        R[bad] = rotations(len(bad))
        O[bad] += rng.normal(0.0, 0.05, size=(len(bad), 3))
    else:
        raise RuntimeError("water(): could not remove all close contacts")
    types = np.tile(np.array([1, 2, 2], dtype=np.int32), nmol)
    return cell, pos, types
