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
