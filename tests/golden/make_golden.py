"""Generates tests/golden/*.npz (run HERE, where /root/reference exists; the outputs travel).

Inputs  : frame 0 of the four geometry files the reference's tests hold
          (/root/reference/tests/test_data/{Cu-cubic,CuPd-cubic-big,Cu2AgO4,aspirin}.xyz) at the
          cutoffs of /root/reference/tests/conftest.py:54-64, prepared as conftest.py:186-194 does
          (non-periodic -> 50 A cubic box, centred; wrapped).  Plus the BASELINE config-1 64-atom Si box.
Outputs : forces / per-atom energies / PE / virial of the float64 torch oracle (oracle/allegro_torch.py)
          driven through the glue restatement (oracle/glue.py) on the LAMMPS-like single-rank system,
          ghost forces folded back onto their owners by tag (what LAMMPS' reverse_comm does).
The model is the hyper-parameter set of tests/test_data/test_repro_allegro.yaml:80-103 (l_max 2,
3 layers, 64 scalars, 32 tensor features, MLP 2x64, readout 1x32, 8 Bessels, p=6) with the build's
seeded initialiser (seed 1); avg_num_neighbors = mean edge count of the structure.
The reference stores no expected outputs for this path, so these vectors pin the build's own
oracle (regression + cross-machine), not nequip -- "parity unpinned", see DESIGN.md.
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import allegro_torch, glue  # noqa: E402
from pair_allegro_amd import lmp_like, model_file  # noqa: E402

REF = "/root/reference/tests/test_data"
HERE = os.path.dirname(os.path.abspath(__file__))

CASES = [
    ("CuPd-cubic-big.xyz", ["Cu", "Pd"], 5.0),
    ("aspirin.xyz", ["C", "H", "O"], 5.0),
    ("aspirin.xyz", ["C", "H", "O"], 15.0),
    ("Cu2AgO4.xyz", ["Cu", "Ag", "O"], 5.0),
    ("Cu-cubic.xyz", ["Cu"], 5.0),
    ("Cu-cubic.xyz", ["Cu"], 15.0),
]


def read_first_frame(path):
    with open(path) as f:
        n = int(f.readline())
        comment = f.readline()
        sym, pos = [], []
        for _ in range(n):
            t = f.readline().split()
            sym.append(t[0])
            pos.append([float(t[1]), float(t[2]), float(t[3])])
    cell = None
    if 'Lattice="' in comment:
        lat = comment.split('Lattice="')[1].split('"')[0].split()
        cell = np.array([float(v) for v in lat]).reshape(3, 3)
    periodic = 'pbc="T T T"' in comment
    return sym, np.array(pos), cell, periodic


def prepare(sym, pos, cell, periodic):
    if not periodic:                                   # conftest.py:186-190
        L = 50.0
        cell = L * np.eye(3)
        pos = pos - 0.5 * (pos.min(0) + pos.max(0)) + 0.5 * L    # ase Atoms.center()
    return cell, lmp_like.wrap(cell, pos)              # conftest.py:191-193


def run_case(name, cell, pos, symbols, model_types, r_max, cfg_over, tag):
    lmp_names = sorted(set(symbols))                   # ASE / the deck: alphabetical LAMMPS types
    types = np.array([lmp_names.index(s) + 1 for s in symbols], dtype=np.int32)
    rs = lmp_like.build_rank_system(cell, pos, types, r_max + 1.0)      # `neighbor 1.0 bin`
    nedge = len(glue.brute_force_edges(cell, pos, r_max)[0])
    cfg = model_file.DEFAULT_CFG.copy()
    cfg.update(model_dtype="float64", type_names=list(model_types), r_max=float(r_max),
               avg_num_neighbors=float(nedge) / len(pos), seed=1)
    cfg.update(cfg_over)
    w = model_file.init_weights(cfg)
    oracle = allegro_torch.build(cfg, w)
    mapper = np.array([model_types.index(s) if s in model_types else -1 for s in lmp_names], dtype=np.int32)
    cm = np.full((len(lmp_names), len(lmp_names)), float(r_max))
    f = np.zeros_like(rs.x)
    ea = np.zeros(rs.nall)
    eng, vir, inp = glue.compute(oracle, rs.x, rs.type, rs.nlocal, rs.ilist, rs.numneigh, rs.firstneigh, mapper, cm, f, ea)
    n = len(pos)
    forces = np.zeros((n, 3))
    np.add.at(forces, rs.tag - 1, f)                   # reverse comm: ghosts -> owners
    eatom = np.zeros(n)
    eatom[rs.tag[: rs.nlocal] - 1] = ea[: rs.nlocal]
    blob = model_file.dumps(cfg, w)
    out = os.path.join(HERE, f"{tag}.npz")
    np.savez_compressed(out, cell=cell, pos=pos, symbols=np.array(symbols), lmp_type_names=np.array(lmp_names),
                        cfg=json.dumps(cfg), weights_sha256=hashlib.sha256(blob).hexdigest(),
                        forces=forces, eatom=eatom, pe=eng, virial=vir, nedges=inp["edge_index"].shape[1])
    print(f"{tag}: N={n} E={inp['edge_index'].shape[1]} pe={eng:.6f} |F|max={np.abs(forces).max():.4f} -> {out}")


def main_l3():
    """Round 6: l_max = 3 and widths off every fused shape (free hyper-parameters of /root/reference/tests/test_data/test_repro_allegro.yaml:89-99): the
    layer-at-a-time kernels.  Two of the reference's geometries: the 7-atom triclinic Cu2AgO4 cell (3 types) and the 4-atom Cu cell at 5 A."""
    odd = dict(l_max=3, num_layers=2, num_scalar_features=48, num_tensor_features=16, mlp_width=40, readout_width=24)
    for fname, model_types, r_max in (("Cu2AgO4.xyz", ["Cu", "Ag", "O"], 5.0), ("Cu-cubic.xyz", ["Cu"], 5.0)):
        sym, pos, cell, periodic = read_first_frame(os.path.join(REF, fname))
        cell, pos = prepare(sym, pos, cell, periodic)
        run_case(fname, cell, pos, sym, model_types, r_max, odd, f"{fname.split('.')[0]}_r{int(r_max)}_l3")


def main():
    if "--l3" in sys.argv:          # only the round-6 additions (the older vectors stay byte for byte)
        return main_l3()
    yaml_model = dict(l_max=2, num_layers=3, num_scalar_features=64, num_tensor_features=32)
    for fname, model_types, r_max in CASES:
        sym, pos, cell, periodic = read_first_frame(os.path.join(REF, fname))
        cell, pos = prepare(sym, pos, cell, periodic)
        tag = f"{fname.split('.')[0]}_r{int(r_max)}"
        run_case(fname, cell, pos, sym, model_types, r_max, yaml_model, tag)
    # BASELINE config 1: 64-atom Si, model S
    cell, pos, _ = lmp_like.diamond_si(2)
    run_case("si64", cell, pos, ["Si"] * 64, ["Si"], 5.0, dict(l_max=1, num_layers=2, num_tensor_features=32), "Si64_r5")
    # BASELINE config 3 / config 5 samples (generators of pair_allegro_amd/lmp_like.py, seeded): 128-atom Li3PO4 with model S,
    # 192-atom water with model L (l_max 2, 64 tensor features, 3 layers)
    cell, pos, lt = lmp_like.li3po4(reps=(1, 2, 2))
    sym = [lmp_like.LI3PO4_LAMMPS_NAMES[t - 1] for t in lt]
    run_case("li3po4", cell, pos, sym, ["Li", "P", "O"], 5.0, dict(l_max=1, num_layers=2, num_tensor_features=32), "Li3PO4_128_r5")
    cell, pos, wt = lmp_like.water(m=4)
    sym = [["O", "H"][t - 1] for t in wt]
    run_case("water", cell, pos, sym, ["O", "H"], 5.0, dict(l_max=2, num_layers=3, num_tensor_features=64), "water_192_r5")


if __name__ == "__main__":
    main()
