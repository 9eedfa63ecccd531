"""One pair-style object through a run's worth of different situations (-m gpu).

The parity tests create a fresh `PairAllegro` per evaluation; a LAMMPS run keeps ONE for thousands of steps while the number of ghosts, the
neighbour list, the largest degree and even the kernel family change under it (/root/reference/pair_nequip_allegro.cpp:333-407 is entered
once per step on the same object).  State that survives a call -- page-locked staging vectors that grow, the counters that are read back
lazily, the tile arrays packed by the edge build, the last edge total used as a size hint, the heavy-centre list of the wide kernels --
is exactly what such a sequence exercises.  Every evaluation of the long-lived object must equal the evaluation of a fresh object on the
same inputs (same kernels, so the tolerance is the float64 atomics' arrival order), and take the same kernel path."""
import numpy as np
import pytest

import parity_cases as pc  # noqa: E402
import util
from oracle import allegro_torch
from pair_allegro_amd import lmp_like, model_file
from pair_allegro_amd.pair import PairAllegro, atom_from_rank_system, list_from_rank_system

pytestmark = pytest.mark.gpu


def _si(ncell, scale, seed):
    """Si diamond box, uniformly compressed by `scale` (density, hence degree, grows as scale^-3)."""
    cell, pos, types = lmp_like.diamond_si(ncell, seed=seed)
    return cell * scale, pos * scale, types


def _new_pair(hip_lib, path, names, options):
    pair = PairAllegro(me=0, nprocs=1, lib=hip_lib, quiet=True)
    pair.settings([])
    pair.coeff(["*", "*", path] + list(names), ntypes=len(names))
    for k, v in options.items():
        pair.model.set_option(k, v)
    pair.init_style()
    return pair


def _evaluate(pair, rs, names, x=None, list_changed=True, lst=None):
    atom = atom_from_rank_system(rs, len(names))
    if x is not None:
        atom.x = x
    lst = lst or list_from_rank_system(rs)
    pair.compute(atom, lst, list_changed=list_changed)
    ea = pair.eatom[: rs.nlocal].copy() if rs.nlocal else np.zeros(0)
    return dict(f=atom.f.copy(), eatom=ea, pe=pair.eng_vdwl, virial=pair.virial.copy(),
                path=pair.model.last_path if rs.nlocal else "none"), lst


def _same(a, b, what):
    assert a["path"] == b["path"], f"{what}: path {a['path']} vs fresh {b['path']}"
    fs = max(np.abs(b["f"]).max(), 1e-30)
    assert np.abs(a["f"] - b["f"]).max() <= 1e-9 * fs, f"{what}: forces differ by {np.abs(a['f'] - b['f']).max():.3e}"
    np.testing.assert_allclose(a["eatom"], b["eatom"], rtol=1e-12, atol=1e-12, err_msg=what)
    np.testing.assert_allclose(a["pe"], b["pe"], rtol=1e-12, atol=1e-12, err_msg=what)
    np.testing.assert_allclose(a["virial"], b["virial"], rtol=1e-9, atol=1e-9 * max(1.0, np.abs(b["virial"]).max()), err_msg=what)


def _run_sequence(hip_lib, path, cfg, names, seq, options):
    """seq: (label, ncell, scale, expected path or None).  Every entry: list hand-over + evaluation, then the same list with moved atoms."""
    keep = _new_pair(hip_lib, path, names, options)
    seen = []
    for label, ncell, scale, expect in seq:
        if ncell == 0:                                   # an empty sub-domain between two populated ones (reference :340-341)
            rs = lmp_like.build_rank_system(np.eye(3) * 30.0, np.zeros((0, 3)), np.zeros(0, dtype=np.int32), cfg["r_max"] + 1.0)
            res, _ = _evaluate(keep, rs, names)
            assert res["pe"] == 0.0 and not res["virial"].any()
            seen.append("none")
            continue
        cell, pos, types = _si(ncell, scale, seed=ncell)
        rs = lmp_like.build_rank_system(cell, pos, types, cfg["r_max"] + 1.0)
        res, lst = _evaluate(keep, rs, names)
        fresh = _new_pair(hip_lib, path, names, options)
        ref, _ = _evaluate(fresh, rs, names)
        _same(res, ref, f"{label}: list hand-over")
        if expect is not None:
            assert res["path"] == expect, f"{label}: {res['path']}"
        if "heavy" in label or "degree" in label:         # the label's claim about the degrees, from the library's own counter
            md = keep.model.last_max_degree
            assert (md > 64) == (label in ("some heavy centres", "mostly heavy", "nearly all heavy", "list rows > 128", "degree > 64", "degrees around 64", "degree > 128", "degree > 64 again")), (label, md)
        seen.append(res["path"])
        # a later step of the same neighbour-list epoch: atoms have moved (less than half the skin), the list object is the same
        rng = np.random.RandomState(7 + ncell)
        x2 = rs.x.copy()
        disp = rng.uniform(-0.12, 0.12, size=(len(pos), 3))
        x2 += disp[rs.tag - 1]                            # ghosts move with their owners
        res2, _ = _evaluate(keep, rs, names, x=x2, list_changed=False, lst=lst)
        ref2, _ = _evaluate(fresh, rs, names, x=x2, list_changed=False)
        _same(res2, ref2, f"{label}: moved atoms, same list")
        assert np.abs(res2["f"] - res["f"]).max() > 1e-6      # it really is another configuration
        fresh.model.close()
    keep.model.close()
    return seen


def test_one_pair_object_through_growing_shrinking_and_denser_systems(hip_lib, model_dir):
    """Model S: 216 -> 1000 atoms (staging buffers grow) -> empty sub-domain -> 216 again -> compressed boxes whose degrees walk through
    <= 64 with list rows > 64 (tile shape chosen on the device), 65..128 (8-wave tiles), > 128 (two-pass edge build + layer-at-a-time
    kernels) -> back to the plain box."""
    cfg = model_file.model_S(type_names=["Si"], avg_num_neighbors=28.0)
    w = model_file.init_weights(cfg)
    path = f"{model_dir}/lifecycle_S.nequip.pth"
    allegro_torch.export_nequip_pth(path, cfg, w)
    seq = [("small", 3, 1.0, pc.FUSED_DEFAULT), ("grown", 5, 1.0, pc.FUSED_DEFAULT), ("empty", 0, 0, None), ("small again", 3, 1.0, pc.FUSED_DEFAULT),
           ("rows > 64", 3, 0.86, pc.FUSED_DEFAULT), ("degree > 64", 3, 0.74, pc.FUSED_DEFAULT), ("degrees around 64", 3, 0.748, pc.FUSED_DEFAULT), ("degree > 128", 3, 0.55, "generic_f32"),
           ("plain after fallback", 4, 1.0, pc.FUSED_DEFAULT), ("degree > 64 again", 4, 0.74, pc.FUSED_DEFAULT)]
    seen = _run_sequence(hip_lib, path, cfg, ["Si"], seq, {})
    assert seen.count(pc.FUSED_DEFAULT) == 8 and seen.count("generic_f32") == 1


def test_one_pair_object_with_heavy_centres_coming_and_going(hip_lib, model_dir):
    """Model L (wide kernel, 64-slot tiles): centres with more than 64 edges are left to the layer-at-a-time kernels, and how many there are
    is read back only when there can be any (after the wide kernel has been enqueued) -- none, some (11 % of the centres), none, most, nearly all (93 %), a list with rows
    longer than 128 entries (two-pass edge build, whole system on the layer-at-a-time kernels), an empty sub-domain, none."""
    cfg = model_file.model_L(type_names=["Si"], avg_num_neighbors=28.0)
    w = model_file.init_weights(cfg)
    path = f"{model_dir}/lifecycle_L.nequip.pth"
    allegro_torch.export_nequip_pth(path, cfg, w)
    seq = [("no heavy centres", 3, 1.0, pc.FUSED_DEFAULT), ("some heavy centres", 3, 0.748, pc.FUSED_DEFAULT), ("none again", 4, 1.0, pc.FUSED_DEFAULT),
           ("mostly heavy", 3, 0.745, pc.FUSED_DEFAULT), ("nearly all heavy", 3, 0.742, pc.FUSED_DEFAULT), ("list rows > 128", 3, 0.70, "generic_f32"), ("empty", 0, 0, None), ("none at the end", 3, 1.0, pc.FUSED_DEFAULT)]
    seen = _run_sequence(hip_lib, path, cfg, ["Si"], seq, {})
    assert seen == [pc.FUSED_DEFAULT] * 5 + ["generic_f32", "none", pc.FUSED_DEFAULT], seen
