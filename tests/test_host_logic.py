"""Command surface and error behaviour of the host mirror (same checks and texts as
/root/reference/pair_nequip_allegro.cpp:137-206,394), model-file plumbing, C-ABI argument checks."""
import io
import os
import contextlib

import numpy as np
import pytest

import util
from oracle import allegro_torch
from pair_allegro_amd import capi, lmp_like, model_file
from pair_allegro_amd.pair import LammpsError, PairAllegro, atom_from_rank_system, list_from_rank_system


@pytest.fixture(scope="module")
def si_model(model_dir):
    cfg = model_file.model_S(model_dtype="float64", num_scalar_features=16, num_tensor_features=8, mlp_width=16, readout_width=8)
    path = os.path.join(model_dir, "host_logic.nequip.pth")
    allegro_torch.export_nequip_pth(path, cfg)
    return path, cfg


def test_pair_style_takes_no_arguments(emu_lib):
    p = PairAllegro(lib=emu_lib, quiet=True)
    p.settings([])
    with pytest.raises(LammpsError, match="Illegal pair_style command, too many arguments"):
        p.settings(["foo"])


def test_pair_coeff_argument_checks(emu_lib, si_model):
    path, _ = si_model
    p = PairAllegro(lib=emu_lib, quiet=True)
    with pytest.raises(LammpsError, match="Incorrect args for pair coefficients, should be"):
        p.coeff(["*", "*", path], ntypes=1)
    with pytest.raises(LammpsError, match="Incorrect args for pair coefficients"):
        p.coeff(["1", "*", path, "Si"], ntypes=1)
    with pytest.raises(RuntimeError, match="Only accepts model paths with extension"):
        p.coeff(["*", "*", "model.pt", "Si"], ntypes=1)
    with pytest.raises(RuntimeError, match="cannot open model file"):
        p.coeff(["*", "*", "/nonexistent/x.nequip.pth", "Si"], ntypes=1)
    p.coeff(["*", "*", path, "Si"], ntypes=1)
    assert p.cutoff == 5.0 and p.type_mapper == [0] and p.setflag[1][1] == 1 and p.init_one(1, 1) == 5.0
    assert p.restartinfo == 0 and p.manybody_flag == 1


def test_type_mapping_by_name_and_unmapped_types(emu_lib, model_dir):
    cfg = model_file.model_S(model_dtype="float64", type_names=["Cu", "Ag", "O"], num_scalar_features=16,
                             num_tensor_features=8, mlp_width=16, readout_width=8)
    path = os.path.join(model_dir, "three.nequip.pth")
    allegro_torch.export_nequip_pth(path, cfg)
    p = PairAllegro(lib=emu_lib, quiet=True)
    p.coeff(["*", "*", path, "Ag", "Cu", "O", "O"], ntypes=4)          # two LAMMPS types -> one model type
    assert p.type_mapper == [1, 0, 2, 2]
    p.coeff(["*", "*", path, "Ag", "Cu", "X"], ntypes=3)
    assert p.type_mapper == [1, 0, -1] and p.setflag[1][3] == 0 and p.setflag[1][2] == 1
    out = io.StringIO()
    with contextlib.redirect_stdout(out):
        q = PairAllegro(lib=emu_lib)
        q.coeff(["*", "*", path, "Ag", "Cu", "O"], ntypes=3)
    assert "NequIP/Allegro: Loading model from" in out.getvalue() and "0 | Cu | 2 | Cu" in out.getvalue()


def test_init_style_requirements(emu_lib):
    p = PairAllegro(lib=emu_lib, quiet=True)
    with pytest.raises(LammpsError, match="Pair style Allegro requires atom IDs"):
        p.init_style(tag_enable=0)
    with pytest.raises(LammpsError, match="Pair style allegro requires newton pair on"):
        p.init_style(newton_pair=0)
    assert p.init_style() == {"full": True, "ghost": True}


def test_ranks_vs_devices(emu_lib, monkeypatch):
    with pytest.raises(LammpsError, match="mismatch between number of ranks and number of available GPUs"):
        PairAllegro(me=3, nprocs=4, lib=emu_lib, quiet=True)          # emulation exposes one device
    monkeypatch.setenv("_NEQUIP_LOG_LEVEL", "DEBUG")
    p = PairAllegro(me=3, nprocs=4, lib=emu_lib, quiet=True)           # debug mode wraps around (:104-110)
    assert p.device == 0 and p.debug_mode == 1


def test_compute_errors_and_empty_domain(emu_lib, si_model):
    path, cfg = si_model
    cell, pos, types = lmp_like.diamond_si(2)
    rs = lmp_like.build_rank_system(cell, pos, types, 6.0)
    p = PairAllegro(lib=emu_lib, quiet=True)
    p.coeff(["*", "*", path, "Si"], ntypes=1)
    atom = atom_from_rank_system(rs, 1)
    with pytest.raises(LammpsError, match="do not support per-atom virial"):
        p.compute(atom, list_from_rank_system(rs), vflag_atom=True)
    lst = list_from_rank_system(rs)
    lst.inum = 0
    p.compute(atom, lst)
    assert p.eng_vdwl == 0.0 and not atom.f.any()
    # forces are ADDED to f (pair_nequip_allegro.cpp:375-377)
    atom.f[:] = 1.0
    p.compute(atom, list_from_rank_system(rs))
    base = atom.f.copy() - 1.0
    atom.f[:] = 0.0
    p.compute(atom, list_from_rank_system(rs))
    np.testing.assert_allclose(atom.f, base, atol=1e-12)


def test_debug_edge_dump_format(emu_lib, si_model, capfd, monkeypatch):
    """`Allegro edges: i j rij` ... `end Allegro edges`, 0-based tag-1 ids, %.10g
    (pair_nequip_allegro.cpp:564,625,632; parsed by tests/test_python_repro_allegro.py:203-217)."""
    path, cfg = si_model
    monkeypatch.setenv("_NEQUIP_LOG_LEVEL", "DEBUG")
    cell, pos, types = lmp_like.diamond_si(2)
    rs = lmp_like.build_rank_system(cell, pos, types, 6.0)
    p = PairAllegro(lib=emu_lib, quiet=True)
    p.coeff(["*", "*", path, "Si"], ntypes=1)
    p.compute(atom_from_rank_system(rs, 1), list_from_rank_system(rs))
    text = capfd.readouterr().out
    body = text.split("Allegro edges: i j rij\n")[1].split("end Allegro edges")[0].strip().splitlines()
    assert len(body) == 1792
    i, j, d = zip(*[(int(a), int(b), float(c)) for a, b, c in (ln.split() for ln in body)])
    bi, bj, bd = util.glue.brute_force_edges(cell, pos, 5.0)
    assert sorted(zip(i, j)) == sorted(zip(bi.tolist(), bj.tolist()))
    np.testing.assert_allclose(np.sort(d), np.sort(bd), atol=1e-8)


def test_capi_argument_and_state_errors(emu_lib, si_model, tmp_path):
    path, cfg = si_model
    with pytest.raises(capi.AhipError) as e:
        capi.Model(str(tmp_path / "x.bin"), 0, emu_lib)
    assert e.value.code == capi.AHIP_ERR_FILE
    bad = tmp_path / "bad.nequip.pth"
    bad.write_bytes(b"not a zip at all" * 10)
    with pytest.raises(capi.AhipError, match="not a zip archive"):
        capi.Model(str(bad), 0, emu_lib)
    m = capi.Model(path, 0, emu_lib)
    x = np.zeros((2, 3)); f = np.zeros((2, 3))
    with pytest.raises(capi.AhipError) as e:
        m.compute(2, 0, x, np.ones(2, np.int32), np.zeros(1, np.int32), np.full((1, 1), 5.0), f)
    assert e.value.code == capi.AHIP_ERR_STATE
    with pytest.raises(capi.AhipError, match="unknown option"):
        m.set_option("nope", "1")
    with pytest.raises(capi.AhipError, match="neighbour index out of range"):
        m.neigh_update_csr(2, np.array([0, 1]), np.array([0, 1, 2]), np.array([1, 7]))
    m.neigh_update_csr(2, np.array([0, 1]), np.array([0, 1, 2]), np.array([1, 0]))
    with pytest.raises(capi.AhipError, match="not mapped"):
        m.compute(2, 0, x, np.ones(2, np.int32), np.array([-1], np.int32), np.full((1, 1), 5.0), f)
    assert m.type_names == ["Si"] and m.r_max == 5.0 and m.model_dtype == "float64"
    m.close()


def test_model_file_roundtrip_and_per_edge_type_cutoff(emu_lib, tmp_path):
    cfg = model_file.model_S(type_names=["A", "B"], per_edge_type_cutoff=[[5.0, 4.0], [4.0, 3.5]], model_dtype="float64",
                             num_scalar_features=16, num_tensor_features=8, mlp_width=16, readout_width=8)
    w = model_file.init_weights(cfg)
    c2, w2 = model_file.loads(model_file.dumps(cfg, w))
    assert c2["per_edge_type_cutoff"] == cfg["per_edge_type_cutoff"] and c2["type_names"] == ["A", "B"]
    for k in w:
        assert np.array_equal(w[k], w2[k])
    path = str(tmp_path / "pc.nequip.pth")
    allegro_torch.export_nequip_pth(path, cfg, w)
    p = PairAllegro(lib=emu_lib, quiet=True)
    p.coeff(["*", "*", path, "B", "A"], ntypes=2)
    np.testing.assert_array_equal(p.cutoff_matrix, [[3.5, 4.0], [4.0, 5.0]])
    # parity with per-pair cutoffs (filter + envelope both use them)
    rng = np.random.RandomState(0)
    cell = np.eye(3) * 9.0
    pos = rng.uniform(0, 9, size=(30, 3))
    types = rng.randint(1, 3, size=30).astype(np.int32)
    res = util.run_pair(emu_lib, path, cell, pos, types, ["B", "A"])
    ref = util.oracle_run(cfg, w, cell, pos, types, ["B", "A"])
    util.assert_close_to(res, ref, 1e-8, what="per-edge-type cutoff")


def test_stage_timings_accumulate_between_reads(emu_lib, model_dir):
    """Round 4: `timing=1` records an event pair per stage and call and nobody waits for them inside ahip_compute*; `ahip_get_timings` returns, per stage, the SUM over the
    calls since the previous read and `ahip_get_timing_counts` the number of launches behind it (bench.py reads them once behind its timed region; read after every call
    they are that call's timings, as before)."""
    import util
    from pair_allegro_amd import capi, lmp_like
    g = util.load_golden("Si64_r5")
    path, cfg, w = util.golden_model(g, model_dir, "float64")
    types, names = util.lammps_types(g)
    rs = lmp_like.build_rank_system(g["cell"], g["pos"], types, cfg["r_max"] + 1.0)
    m = capi.Model(path, 0, emu_lib)
    m.set_option("timing", "1")
    m.neigh_update_csr(rs.nall, rs.ilist, rs.offsets, rs.flat)
    mapper = np.zeros(1, np.int32); cm = np.full((1, 1), cfg["r_max"])
    def one():
        f = np.zeros_like(rs.x); e = np.zeros(rs.nall)
        m.compute(rs.nlocal, rs.nghost, rs.x, rs.type, mapper, cm, f, e)
    one()
    t1, c1 = m.timings_and_counts()
    assert "model_generic" in t1 and c1["model_generic"] == 1 and t1["model_generic"] > 0
    assert m.timings() == {}                                   # nothing recorded since the last read
    for _ in range(3):
        one()
    t3, c3 = m.timings_and_counts()
    assert c3["model_generic"] == 3 and set(t3) == set(t1)
    assert t3["model_generic"] > 1.5 * t1["model_generic"] * 0.5          # a sum over three calls, not the last call's value (loose: host timers)
    m.close()


def test_host_list_larger_than_the_staging_buffer_arrives_intact(emu_lib, si_model):
    """Pageable host arrays reach the device in chunks through the library's page-locked staging buffer (csrc/engine.h: copy_h2d, 8 MiB): a row whose only
    in-range neighbour sits behind the first chunk boundary must still produce its edge (and nothing else)."""
    path, cfg = si_model
    m = capi.Model(path, 0, emu_lib)
    n_far = 2_300_000                                     # 9.2 MB of int32 indices: two chunks
    x = np.array([[0.0, 0.0, 0.0], [1.5, 0.0, 0.0], [100.0, 0.0, 0.0]])
    row = np.full(n_far + 1, 2, dtype=np.int32)
    row[-1] = 1
    m.neigh_update_csr(3, np.array([0], dtype=np.int32), np.array([0, len(row)], dtype=np.int64), row)
    f = np.zeros_like(x)
    eatom = np.zeros(3)
    m.compute(1, 2, x, np.ones(3, dtype=np.int32), np.array([0], dtype=np.int32), np.full((1, 1), cfg["r_max"]), f, eatom)
    ei, r = m.get_edges()
    assert ei.shape == (2, 1) and ei[0, 0] == 0 and ei[1, 0] == 1
    np.testing.assert_allclose(r, [1.5], rtol=1e-12)
    np.testing.assert_allclose(f[0], -f[1], atol=1e-12)       # the pair force; the far atom feels nothing
    assert not f[2].any()
    m.close()
