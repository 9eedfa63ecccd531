"""`compute allegro` / `compute allegro/atom` (reference compute/compute_allegro.cpp): the Python mirror and the
C-ABI behind it on the CPU emulation, against the float64 oracle.  Same deck surface, same error texts."""
import numpy as np
import pytest

import util
from oracle import allegro_torch
from pair_allegro_amd import lmp_like, model_file
from pair_allegro_amd.compute import ComputeAllegro
from pair_allegro_amd.pair import LammpsError, PairAllegro, atom_from_rank_system, list_from_rank_system


def _setup(model_dir):
    g = util.load_golden("Cu2AgO4_r5")
    cfg = model_file.model_S(model_dtype="float64", type_names=["Cu", "Ag", "O"], num_scalar_features=16,
                             num_tensor_features=8, mlp_width=16, readout_width=8, avg_num_neighbors=37.0)
    w = model_file.init_weights(cfg)
    path = f"{model_dir}/compute.nequip.pth"
    allegro_torch.export_nequip_pth(path, cfg, w)
    types, names = util.lammps_types(g)
    ref = util.oracle_run(cfg, w, g["cell"], g["pos"], types, names)
    return g, cfg, w, path, types, names, ref


def _vir33(v6):
    xx, yy, zz, xy, xz, yz = v6
    return np.array([[xx, xy, xz], [xy, yy, yz], [xz, yz, zz]])


def _run(lib, path, g, types, names, grid, computes):
    """One force evaluation per rank with the computes attached; returns the rank-reduced results the way LAMMPS
    would produce them (all-reduce for vectors, reverse communication by tag for newton per-atom arrays)."""
    n = len(g["pos"])
    out = {}
    for r in lmp_like.grid_ranks(grid):
        pair = PairAllegro(me=0, nprocs=1, lib=lib, quiet=True)
        pair.settings([])
        pair.coeff(["*", "*", path] + list(names), ntypes=len(names))
        cs = [ComputeAllegro(a, pair) for a in computes]
        pair.init_style()
        rs = lmp_like.build_rank_system(g["cell"], g["pos"], types, pair.init_one(1, 1) + 1.0, grid=grid, rank=r)
        atom = atom_from_rank_system(rs, len(names))
        pair.compute(atom, list_from_rank_system(rs))
        for a, c in zip(computes, cs):
            key = (a[2], a[3])
            if not c.peratom:
                out[key] = out.get(key, 0) + c.compute_vector(rs.nlocal).copy()
            else:
                arr = c.compute_peratom(rs.nlocal, rs.nall)
                if c.newton:                                   # comm->reverse_comm(this)
                    buf = c.pack_reverse_comm(rs.nghost, rs.nlocal)
                    glob = out.setdefault(key, np.zeros((n, c.nperatom)))
                    np.add.at(glob, rs.tag[: rs.nlocal] - 1, arr[: rs.nlocal])
                    np.add.at(glob, rs.tag[rs.nlocal:] - 1, buf.reshape(-1, c.nperatom))
                else:
                    glob = out.setdefault(key, np.zeros((n, c.nperatom)))
                    glob[rs.tag[: rs.nlocal] - 1] = arr[: rs.nlocal]
        pair.model.close()
    return out


@pytest.mark.parametrize("grid", [(1, 1, 1), (2, 1, 1)])
def test_compute_allegro_against_oracle(emu_lib, model_dir, grid):
    g, cfg, w, path, types, names, ref = _setup(model_dir)
    computes = [["v", "all", "allegro", "virial", "9"],
                ["f", "all", "allegro/atom", "forces", "3", "1"],
                ["e", "all", "allegro/atom", "atomic_energy", "1", "0"]]
    out = _run(emu_lib, path, g, types, names, grid, computes)
    np.testing.assert_allclose(out[("allegro", "virial")].reshape(3, 3), _vir33(ref["virial"]), atol=1e-8)
    np.testing.assert_allclose(out[("allegro/atom", "forces")], ref["forces"], atol=1e-9)
    np.testing.assert_allclose(out[("allegro/atom", "atomic_energy")][:, 0], ref["eatom"], atol=1e-10)


def test_total_energy_includes_ghost_shifts(emu_lib, model_dir):
    """compute/README.md: the model's global energy also carries the shifts of the ghost atoms."""
    g, cfg, w, path, types, names, ref = _setup(model_dir)
    pair = PairAllegro(me=0, nprocs=1, lib=emu_lib, quiet=True)
    pair.settings([])
    pair.coeff(["*", "*", path] + list(names), ntypes=len(names))
    c = ComputeAllegro(["t", "all", "allegro", "total_energy", "1"], pair)
    rs = lmp_like.build_rank_system(g["cell"], g["pos"], types, 6.0)
    atom = atom_from_rank_system(rs, len(names))
    pair.compute(atom, list_from_rank_system(rs))
    mapper = np.asarray(pair.type_mapper)
    ghost_shift = w["shift"][mapper[rs.type[rs.nlocal:] - 1]].sum()
    np.testing.assert_allclose(c.compute_vector(rs.nlocal)[0], ref["pe"] + ghost_shift, rtol=1e-10)
    assert ComputeAllegro(["t", "all", "allegro", "total_energy", "1"], pair).compute_vector(0)[0] == 0.0     # empty domain
    pair.model.close()


def test_compute_allegro_deck_errors(emu_lib, model_dir):
    g, cfg, w, path, types, names, ref = _setup(model_dir)
    pair = PairAllegro(me=0, nprocs=1, lib=emu_lib, quiet=True)
    pair.settings([])
    pair.coeff(["*", "*", path] + list(names), ntypes=len(names))
    with pytest.raises(LammpsError, match="Incorrect args for compute allegro$"):
        ComputeAllegro(["c", "all", "allegro", "virial"], pair)
    with pytest.raises(LammpsError, match="Incorrect args for compute allegro/atom"):
        ComputeAllegro(["c", "all", "allegro/atom", "forces", "3"], pair)
    with pytest.raises(LammpsError, match="can only operate on group 'all'"):
        ComputeAllegro(["c", "mobile", "allegro", "virial", "9"], pair)
    with pytest.raises(LammpsError, match="Incorrect vector length!"):
        ComputeAllegro(["c", "all", "allegro", "virial", "0"], pair)
    with pytest.raises(LammpsError, match="no pair style; compute allegro must be defined after pair style"):
        ComputeAllegro(["c", "all", "allegro", "virial", "9"], None)
    rs = lmp_like.build_rank_system(g["cell"], g["pos"], types, 6.0)
    wrong = ComputeAllegro(["c", "all", "allegro", "virial", "6"], pair)
    atom = atom_from_rank_system(rs, len(names))
    pair.compute(atom, list_from_rank_system(rs))
    with pytest.raises(LammpsError, match="size 9 of quantity tensor virial does not match expected 6 on rank 0"):
        wrong.compute_vector(rs.nlocal)
    ComputeAllegro(["c", "all", "allegro", "polarization", "3"], pair)       # accepted now, fails when the model is evaluated
    with pytest.raises(LammpsError, match="model output 'polarization' not found"):
        pair.compute(atom_from_rank_system(rs, len(names)), list_from_rank_system(rs))
    pair.model.close()
