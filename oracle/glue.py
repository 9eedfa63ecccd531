"""ORACLE (test infrastructure): numpy restatement of the reference pair style's host glue.

Follows /root/reference/pair_nequip_allegro.cpp line by line in meaning (not in text):

* ``preprocess``  -- :457-650  (pass 1 count :488-512, prefix sum :515-519, pass 2 fill :566-629)
* ``compute``     -- :333-407  (call the model :355, scatter :369-380, virial unpack :382-393)

The model is any callable obeying the TorchScript dict contract (``oracle.allegro_torch``
module or a ``torch.jit.load``-ed ``*.nequip.pth``), i.e. exactly what ``call()`` (:409-430) runs.
The edge test is the host path's ``rsq <= cut^2`` (:507,:599; the Kokkos path uses ``<``,
pair_nequip_allegro_kokkos.cpp:189 -- SURVEY.md App. D).
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch

NEIGHMASK = 0x1FFFFFFF


def preprocess(x: np.ndarray, type_: np.ndarray, nlocal: int, ilist: np.ndarray, numneigh: np.ndarray,
               firstneigh: Sequence[np.ndarray], type_mapper: np.ndarray, cutoff_matrix: np.ndarray
               ) -> Dict[str, np.ndarray]:
    """-> {pos f64 [ntotal,3], edge_index i64 [2,E], atom_types i64 [ntotal]} (:524-533,:638-641)."""
    ntotal = len(x)
    centres, neighs = [], []
    for ii in range(nlocal):                                   # :566-629, ii < nlocal branch
        i = int(ilist[ii])
        jl = np.asarray(firstneigh[i][: numneigh[i]], dtype=np.int64) & NEIGHMASK       # :586-587
        d = x[i][None, :] - x[jl]                              # :591-593
        rsq = (d * d).sum(axis=1)                              # :595
        cut = cutoff_matrix[type_[i] - 1, type_[jl] - 1]       # :597-598
        keep = rsq <= cut * cut                                # :599 (inverted continue)
        centres.append(np.full(int(keep.sum()), i, dtype=np.int64))   # :601
        neighs.append(jl[keep])                                # :602 (allegro: ghost index kept)
    ei = np.stack([np.concatenate(centres) if centres else np.zeros(0, np.int64),
                   np.concatenate(neighs) if neighs else np.zeros(0, np.int64)])
    atom_types = np.asarray(type_mapper, dtype=np.int64)[np.asarray(type_, dtype=np.int64) - 1]   # :576
    return {"pos": np.array(x[:ntotal], dtype=np.float64), "edge_index": ei, "atom_types": atom_types}


def compute(model, x: np.ndarray, type_: np.ndarray, nlocal: int, ilist: np.ndarray, numneigh: np.ndarray,
            firstneigh: Sequence[np.ndarray], type_mapper: np.ndarray, cutoff_matrix: np.ndarray,
            f: np.ndarray, eatom: Optional[np.ndarray] = None) -> Tuple[float, np.ndarray, Dict[str, np.ndarray]]:
    """One reference force evaluation; f accumulated in place; returns (eng_vdwl, virial[6], inputs)."""
    inp = preprocess(x, type_, nlocal, ilist, numneigh, firstneigh, type_mapper, cutoff_matrix)
    out = model({k: torch.from_numpy(v) for k, v in inp.items()})          # call(), :409-430
    forces = out["forces"].detach().cpu().numpy()                           # :358
    ae = out["atomic_energy"].detach().cpu().numpy()                        # :361
    ntotal = len(x)
    eng = 0.0
    for ii in range(ntotal):                                                # :372 (allegro: nforces = ntotal)
        i = int(ilist[ii]) if ii < len(ilist) else ii
        f[i] += forces[i]                                                   # :375-377
        if ii < nlocal:
            if eatom is not None:
                eatom[i] = ae[i, 0]                                         # :378
            eng += ae[i, 0]                                                 # :379
    v = out["virial"].detach().cpu().numpy()                                # :383
    virial = np.array([v[0, 0, 0], v[0, 1, 1], v[0, 2, 2], v[0, 0, 1], v[0, 0, 2], v[0, 1, 2]])   # :387-392
    return eng, virial, inp


def brute_force_edges(cell: np.ndarray, pos: np.ndarray, r_max: float):
    """Independent periodic neighbour search (the role nequip's neighbor list plays in
    /root/reference/tests/test_python_repro_allegro.py:259-286): all (i, j, shift) with
    |pos[j] + shift@cell - pos[i]| <= r_max, i != j or shift != 0.  Returns (i, j, dist)."""
    cell = np.asarray(cell, dtype=np.float64)
    vol = abs(np.linalg.det(cell))
    a, b, c = cell
    heights = np.array([vol / np.linalg.norm(np.cross(b, c)), vol / np.linalg.norm(np.cross(c, a)),
                        vol / np.linalg.norm(np.cross(a, b))])
    nmax = np.ceil(r_max / heights).astype(int) + 1
    out_i, out_j, out_d = [], [], []
    for sx in range(-nmax[0], nmax[0] + 1):
        for sy in range(-nmax[1], nmax[1] + 1):
            for sz in range(-nmax[2], nmax[2] + 1):
                sh = np.array([sx, sy, sz], dtype=np.float64) @ cell
                d = pos[None, :, :] + sh[None, None, :] - pos[:, None, :]
                dist = np.sqrt((d * d).sum(-1))
                mask = dist <= r_max
                if sx == 0 and sy == 0 and sz == 0:
                    np.fill_diagonal(mask, False)
                ii, jj = np.nonzero(mask)
                out_i.append(ii); out_j.append(jj); out_d.append(dist[ii, jj])
    return np.concatenate(out_i), np.concatenate(out_j), np.concatenate(out_d)
