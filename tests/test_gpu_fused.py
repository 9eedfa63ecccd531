"""GPU tests of the fused MFMA path: the register-chain primitive, then the whole kernel against
the float64 oracle goldens, the float32 generic path and size-independent properties."""
import os

import numpy as np
import pytest

import parity_cases as pc
import util
from oracle import allegro_torch
from pair_allegro_amd import cg, lmp_like, model_file

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("arith", ["f32", "f16x2", "bf16x3", "tf32eq"])
@pytest.mark.parametrize("K,N", [(8, 64), (32, 32), (32, 64), (64, 32), (64, 64), (96, 64), (64, 96), (64, 8)])
def test_mfma_linear_primitive(hip_lib, K, N, arith, monkeypatch):
    """The streamed register-chain linear, on the f32-input MFMA, on the f16x2 split (two float16 terms per operand, three f16 MFMA products) and on
    the bf16x3 split (six bf16 MFMA terms): all must reproduce x @ W to float32 accuracy."""
    monkeypatch.setenv("AHIP_FUSED_ARITH", arith)
    rng = np.random.RandomState(K * 100 + N)
    W = rng.normal(size=(K, N))                    # asymmetric: catches transposed fragments
    x = rng.normal(size=(32, K)).astype(np.float32)
    out = hip_lib.debug_fused_linear(W, x)
    ref = x.astype(np.float64) @ W
    # tf32eq: two-term bf16 split, dropped terms <= 3 * 2^-17 per product (TF32 itself: 2^-11 per operand)
    tol = 1e-4 if arith == "tf32eq" else 2e-5
    np.testing.assert_allclose(out, ref, atol=tol * np.abs(ref).max(), rtol=tol / 2)


@pytest.mark.parametrize("tag", ["Si64_r5", "Li3PO4_128_r5"])
def test_fused_golden_si64(hip_lib, model_dir, tag):
    """Committed float64 goldens of the model-S workloads (BASELINE config 1's 64-atom Si box, a 128-atom sample of config 3's Li3PO4)."""
    res, g = pc.check_golden(hip_lib, model_dir, tag, "float32", options={"path": "fused"})
    assert res["info"]["path"] in pc.FUSED_F32EQ
    pc.check_edges_vs_brute_force(res, g)


def _model_S_case(model_dir, name, type_names, symbols, cell, pos, nl=2, seed=1, **over):
    cfg = model_file.model_S(type_names=list(type_names), num_layers=nl, seed=seed,
                             avg_num_neighbors=float(len(util.glue.brute_force_edges(cell, pos, 5.0)[0])) / len(pos), **over)
    w = model_file.init_weights(cfg)
    path = f"{model_dir}/{name}.nequip.pth"
    allegro_torch.export_nequip_pth(path, cfg, w)
    names = sorted(set(symbols))
    types = np.array([names.index(s) + 1 for s in symbols], dtype=np.int32)
    cfg64 = dict(cfg, model_dtype="float64")
    ref = util.oracle_run(cfg64, w, cell, pos, types, names)
    return path, cfg, types, names, ref


@pytest.mark.parametrize("nl", [1, 2, 3])
def test_fused_vs_oracle_two_types(hip_lib, model_dir, nl):
    """CuPd 256-atom box (reference geometry), model-S shape with 2 types and 1..3 layers."""
    g = util.load_golden("CuPd-cubic-big_r5")
    path, cfg, types, names, ref = _model_S_case(model_dir, f"cupd_S_nl{nl}", ["Cu", "Pd"], g["symbols"], g["cell"], g["pos"], nl=nl)
    fused = util.run_pair(hip_lib, path, g["cell"], g["pos"], types, names, options={"path": "fused"})
    assert fused["info"]["path"] in pc.FUSED_F32EQ
    gen = util.run_pair(hip_lib, path, g["cell"], g["pos"], types, names, options={"path": "generic"})
    util.assert_close_to(fused, ref, 5e-4, what=f"fused vs f64 oracle nl={nl}")
    assert np.abs(fused["forces"] - ref["forces"]).max() < pc.NORTH_STAR_DF
    np.testing.assert_allclose(fused["forces"], gen["forces"], atol=2e-5)


@pytest.mark.parametrize("nb,p", [(5, 6), (12, 4)])
def test_fused_any_radial_basis_with_the_two_body_table(hip_lib, model_dir, nb, p):
    """`num_bessels` and `polynomial_cutoff_p` are free hyper-parameters of the reference's YAML
    (/root/reference/tests/test_data/test_repro_allegro.yaml:83-86).  The fused kernels see the radial basis only through the tabulated two-body
    embedding, which the host builds from the model's own Bessel block: 5 or 12 functions, p = 4 or 6, stay on the fused path (round 4; before:
    anything but 8 dropped to the layer-at-a-time kernels); fused_tb=mlp, whose first linear is laid out for 8, declines."""
    g = util.load_golden("CuPd-cubic-big_r5")
    path, cfg, types, names, ref = _model_S_case(model_dir, f"nb{nb}_S", ["Cu", "Pd"], g["symbols"], g["cell"], g["pos"], num_bessels=nb, poly_p=p)
    res = util.run_pair(hip_lib, path, g["cell"], g["pos"], types, names, options={"path": "fused"})
    assert res["info"]["path"] in pc.FUSED_F32EQ
    util.assert_close_to(res, ref, 5e-4, what=f"{nb} Bessel functions, p = {p}")
    assert np.abs(res["forces"] - ref["forces"]).max() < pc.NORTH_STAR_DF
    with pytest.raises(Exception, match="8 Bessel"):               # the Pair mirror re-raises the library's error as its LammpsError
        util.run_pair(hip_lib, path, g["cell"], g["pos"], types, names, options={"path": "fused", "fused_tb": "mlp"})


@pytest.mark.parametrize("ntypes", [8, 16])
def test_fused_many_species(hip_lib, model_dir, ntypes):
    """More than four model types (the fused kernels took at most 4 until round 3; real Allegro models routinely carry more): the 256-atom
    CuPd box relabelled into 8 and 16 species, per-edge-type cutoffs between 4.2 and 5 A, on the fused kernel against the float64 oracle
    (64 and 256 two-body spline tables, cutoff table read from memory instead of LDS)."""
    g = util.load_golden("CuPd-cubic-big_r5")
    names = [f"X{k:02d}" for k in range(ntypes)]
    rng = np.random.RandomState(ntypes)
    symbols = [names[k] for k in rng.randint(0, ntypes, size=len(g["pos"]))]
    for k in range(ntypes):
        symbols[k] = names[k]                                     # every species occurs
    pc_ = 5.0 - 0.8 * rng.rand(ntypes, ntypes)
    pc_ = 0.5 * (pc_ + pc_.T)
    cfg = model_file.model_S(type_names=names, per_edge_type_cutoff=pc_.tolist(), avg_num_neighbors=40.0)
    w = model_file.init_weights(cfg)
    path = f"{model_dir}/species{ntypes}.nequip.pth"
    allegro_torch.export_nequip_pth(path, cfg, w)
    types = np.array([names.index(s_) + 1 for s_ in symbols], dtype=np.int32)
    ref = util.oracle_run(dict(cfg, model_dtype="float64"), w, g["cell"], g["pos"], types, names)
    fused = util.run_pair(hip_lib, path, g["cell"], g["pos"], types, names, options={"path": "fused"})
    assert fused["info"]["path"] in pc.FUSED_F32EQ
    util.assert_close_to(fused, ref, 5e-4, what=f"{ntypes} species, fused vs f64 oracle")
    assert np.abs(fused["forces"] - ref["forces"]).max() < pc.NORTH_STAR_DF


def test_allow_tf32_selects_the_two_term_split(hip_lib, model_dir):
    """A model file with allow_tf32 = 1 (the reference hands the key to libtorch, pair_nequip_allegro.cpp:267-270) runs the fused kernel's
    linears on the bf16 matrix cores with the two-term split ("fused_tf32eq"); its error against the float64 oracle must not exceed that of
    a TF32 run of the torch oracle (emulated: operands of every matmul rounded to 10 mantissa bits, util.tf32_emulation); option
    fused_arith=f32 keeps the exact float32 arithmetic; a file with allow_tf32 = 0 never leaves it."""
    g = util.load_golden("CuPd-cubic-big_r5")
    nb = float(len(util.glue.brute_force_edges(g["cell"], g["pos"], 5.0)[0])) / len(g["pos"])
    cfg = model_file.model_S(type_names=["Cu", "Pd"], avg_num_neighbors=nb, allow_tf32=1)
    w = model_file.init_weights(cfg)
    path = f"{model_dir}/tf32.nequip.pth"
    allegro_torch.export_nequip_pth(path, cfg, w)
    names = ["Cu", "Pd"]
    types = np.array([names.index(s_) + 1 for s_ in g["symbols"]], dtype=np.int32)
    ref = util.oracle_run(dict(cfg, model_dtype="float64"), w, g["cell"], g["pos"], types, names)
    with util.tf32_emulation():
        tf = util.oracle_run(dict(cfg, model_dtype="float32"), w, g["cell"], g["pos"], types, names)
    err_tf32 = np.abs(tf["forces"] - ref["forces"]).max()
    res = util.run_pair(hip_lib, path, g["cell"], g["pos"], types, names)
    assert res["info"]["path"] == "fused_tf32eq"
    err = np.abs(res["forces"] - ref["forces"]).max()
    print(f"max|dF| vs f64 oracle: tf32eq kernel {err:.3e}, TF32-emulated torch oracle {err_tf32:.3e}")
    assert err <= err_tf32 and err < 5e-4
    util.assert_close_to(res, ref, 5e-4, what="tf32eq vs f64 oracle (reference tolerance)")
    exact = util.run_pair(hip_lib, path, g["cell"], g["pos"], types, names, options={"fused_arith": "f32"})
    assert exact["info"]["path"] in pc.FUSED_F32EQ and np.abs(exact["forces"] - ref["forces"]).max() < pc.NORTH_STAR_DF
    cfg0 = dict(cfg, allow_tf32=0)
    path0 = f"{model_dir}/notf32.nequip.pth"
    allegro_torch.export_nequip_pth(path0, cfg0, w)
    assert util.run_pair(hip_lib, path0, g["cell"], g["pos"], types, names)["info"]["path"] in pc.FUSED_F32EQ


def test_fused_bf16x3_arithmetic_matches_f32(hip_lib, model_dir):
    """Option fused_arith=bf16x3 (exact 3-way bf16 split, six MFMA terms, f32 accumulate) is float32-equivalent:
    same distance to the float64 oracle as the f32-input MFMA kernel."""
    g = util.load_golden("CuPd-cubic-big_r5")
    path, cfg, types, names, ref = _model_S_case(model_dir, "cupd_S_b3", ["Cu", "Pd"], g["symbols"], g["cell"], g["pos"])
    f32 = util.run_pair(hip_lib, path, g["cell"], g["pos"], types, names, options={"path": "fused"})
    b3 = util.run_pair(hip_lib, path, g["cell"], g["pos"], types, names, options={"path": "fused", "fused_arith": "bf16x3"})
    util.assert_close_to(b3, ref, 5e-4, what="fused bf16x3 vs f64 oracle")
    e32 = np.abs(f32["forces"] - ref["forces"]).max()
    eb3 = np.abs(b3["forces"] - ref["forces"]).max()
    assert eb3 < max(2.0 * e32, 1e-5), (eb3, e32)
    np.testing.assert_allclose(b3["forces"], f32["forces"], atol=1e-5)
    np.testing.assert_allclose(b3["pe"], f32["pe"], rtol=2e-6)


def test_fused_f16x2_arithmetic_is_float32_equivalent(hip_lib, model_dir):
    """fused_arith=f16x2 (csrc/fused_h.h: two float16 terms per operand, the remainder scaled by 2^11; three f16-MFMA products, f32 accumulate; the
    backward pass scaled by a power of two) sits at the same distance from the float64 oracle as the f32-input MFMA kernel -- forces, per-atom
    energies, virial -- on the 256-atom CuPd box (2 types) and on a model whose energy scale is 1e-4 / 1e+4 of the usual one (the backward scale at work)."""
    g = util.load_golden("CuPd-cubic-big_r5")
    path, cfg, types, names, ref = _model_S_case(model_dir, "cupd_S_h2", ["Cu", "Pd"], g["symbols"], g["cell"], g["pos"])
    f32 = util.run_pair(hip_lib, path, g["cell"], g["pos"], types, names, options={"path": "fused", "fused_arith": "f32"})
    h2 = util.run_pair(hip_lib, path, g["cell"], g["pos"], types, names, options={"path": "fused", "fused_arith": "f16x2"})
    assert f32["info"]["path"] == "fused_f32" and h2["info"]["path"] == "fused_f16x2"
    util.assert_close_to(h2, ref, 5e-4, what="fused f16x2 vs f64 oracle")
    e32 = np.abs(f32["forces"] - ref["forces"]).max()
    eh2 = np.abs(h2["forces"] - ref["forces"]).max()
    print(f"max|dF| vs f64 oracle: f32 {e32:.3e}, f16x2 {eh2:.3e}")
    assert eh2 < max(1.5 * e32, pc.F32EQ_DF), (eh2, e32)
    np.testing.assert_allclose(h2["forces"], f32["forces"], atol=1e-5)
    np.testing.assert_allclose(h2["pe"], f32["pe"], rtol=2e-6)
    for sc in (1e-4, 1e4):
        w = model_file.init_weights(cfg)
        w["scale"] = np.asarray(w["scale"]) * sc
        p2 = f"{model_dir}/cupd_S_h2_scale{sc:g}.nequip.pth"
        allegro_torch.export_nequip_pth(p2, cfg, w)
        r64 = util.oracle_run(dict(cfg, model_dtype="float64"), w, g["cell"], g["pos"], types, names)
        a = util.run_pair(hip_lib, p2, g["cell"], g["pos"], types, names, options={"path": "fused", "fused_arith": "f32"})
        b = util.run_pair(hip_lib, p2, g["cell"], g["pos"], types, names, options={"path": "fused", "fused_arith": "f16x2"})
        fm = np.abs(r64["forces"]).max()
        ea, eb = np.abs(a["forces"] - r64["forces"]).max() / fm, np.abs(b["forces"] - r64["forces"]).max() / fm
        print(f"energy scale x{sc:g}: max|dF| / max|F| f32 {ea:.3e}, f16x2 {eb:.3e}")
        assert eb < max(1.5 * ea, 2e-5), (sc, ea, eb)


def test_fused_arith_auto_selection(hip_lib, model_dir):
    """fused_arith=auto: f16x2 for a model file with allow_tf32 = 0 (float32-equivalent, named in the path), exact f32 fmaf chains on request or when the
    two-body embedding is evaluated in the kernel (fused_tb=mlp: the f16x2 instances exist with the table only); tf32eq only for allow_tf32 = 1
    (test_allow_tf32_selects_the_two_term_split)."""
    g = util.load_golden("CuPd-cubic-big_r5")
    path, cfg, types, names, ref = _model_S_case(model_dir, "cupd_S_auto", ["Cu", "Pd"], g["symbols"], g["cell"], g["pos"])
    run = lambda opts: util.run_pair(hip_lib, path, g["cell"], g["pos"], types, names, options=opts)["info"]["path"]
    assert run({}) == pc.FUSED_DEFAULT == "fused_f16x2"
    assert run({"fused_arith": "f32"}) == "fused_f32"
    assert run({"fused_tb": "mlp"}) == "fused_f32"
    assert run({"fused_arith": "bf16x3"}) == "fused_bf16x3"
    assert run({"path": "generic"}) == "generic_f32"


def _blown_up_model(model_dir):
    """two latent linears blown up by 1e3 each: hidden activations ~1e3 and ~1e6, the second exceeds float16's 65504"""
    g = util.load_golden("CuPd-cubic-big_r5")
    cfg = model_file.model_S(type_names=["Cu", "Pd"], avg_num_neighbors=40.0)
    w = model_file.init_weights(cfg)
    w["l1.lat.w0"] = np.asarray(w["l1.lat.w0"]) * 1e3
    w["l1.lat.w1"] = np.asarray(w["l1.lat.w1"]) * 1e3
    path = f"{model_dir}/h2_overflow.nequip.pth"
    allegro_torch.export_nequip_pth(path, cfg, w)
    names = ["Cu", "Pd"]
    types = np.array([names.index(s_) + 1 for s_ in g["symbols"]], dtype=np.int32)
    return g, cfg, w, path, names, types


def test_f16x2_range_alarm(hip_lib, model_dir):
    """float16 has no exponent range to spare: when an operand leaves it the edge gradient comes out non-finite and the kernel raises a flag in
    host-mapped memory.  With an EXPLICIT fused_arith=f16x2 the evaluation reports AHIP_ERR_STATE naming the remedy: the host-pointer call, which waits for the
    kernel anyway, in the same call and before it touches f.  fused_arith=f32 evaluates the same file."""
    g, cfg, w, path, names, types = _blown_up_model(model_dir)
    from pair_allegro_amd.pair import PairAllegro, atom_from_rank_system, list_from_rank_system
    rs = lmp_like.build_rank_system(g["cell"], g["pos"], types, cfg["r_max"] + 1.0)
    pair = PairAllegro(lib=hip_lib, quiet=True)
    pair.settings([])
    pair.coeff(["*", "*", path] + names, ntypes=2)
    pair.model.set_option("path", "fused"); pair.model.set_option("fused_arith", "f16x2")
    atom = atom_from_rank_system(rs, 2)
    lst = list_from_rank_system(rs)
    from pair_allegro_amd.pair import LammpsError
    with pytest.raises(LammpsError) as ei:              # the pair style turns AHIP_ERR_STATE into error->all, as it does with every library error
        pair.compute(atom, lst)
    assert "fused_arith=f32" in str(ei.value)
    assert not atom.f.any()                       # reported before the scatter
    pair.model.set_option("fused_arith", "f32")
    atom.f[:] = 0.0
    pair.compute(atom, lst)
    assert pair.model.last_path == "fused_f32" and np.isfinite(atom.f).all()
    pair.model.close()


def test_auto_is_never_less_robust_than_float32(hip_lib, model_dir):
    """VERDICT r05 #3 / ADVICE r05: the reference only ever RELAXES precision when the file says so (/root/reference/pair_nequip_allegro.cpp:267-270); the default
    arithmetic must therefore evaluate every model float32 evaluates.  Under fused_arith=auto (the default)
      (a) an activation that leaves float16's range      -> the same call re-evaluates on the f32 instance: correct forces, path fused_f32, nothing raised;
      (b) a weight beyond 32768                           -> the f32 weight stream is built instead (explicit f16x2: UnsupportedError);
      (c) a linear whose weights all sit below 2^-10      -> f32 as well (the split would keep < 26 bits of it);
    and the decision is readable (ahip_arith_note)."""
    g, cfg, w, path, names, types = _blown_up_model(model_dir)
    ref32 = util.run_pair(hip_lib, path, g["cell"], g["pos"], types, names, options={"path": "fused", "fused_arith": "f32"})
    for opts in ({}, {"AHIP_NO_ARITH_SELFCHECK": "1"}):            # (a) found by the first-evaluation self-check, and -- without it -- by the alarm of the evaluation itself
        if opts:
            os.environ["AHIP_NO_ARITH_SELFCHECK"] = "1"
        try:
            auto = util.run_pair(hip_lib, path, g["cell"], g["pos"], types, names, options={"path": "fused"})
        finally:
            os.environ.pop("AHIP_NO_ARITH_SELFCHECK", None)
        assert auto["info"]["path"] == "fused_f32" and "float32 instance" in auto["info"]["arith_note"], auto["info"]
        np.testing.assert_allclose(auto["forces"], ref32["forces"], rtol=1e-6, atol=1e-6 * np.abs(ref32["forces"]).max())
        np.testing.assert_allclose(auto["pe"], ref32["pe"], rtol=1e-6)
    # (b), (c): a sane model whose one linear is scaled up by 1e6 and the next one down by 1e6
    g2 = util.load_golden("CuPd-cubic-big_r5")
    cfg2 = model_file.model_S(type_names=["Cu", "Pd"], avg_num_neighbors=40.0)
    for what, edit in (("weight beyond float16", {"l2.env": 1e6}), ("tiny linear", {"l2.env": 1e-5})):
        w2 = model_file.init_weights(cfg2)
        for k, fct in edit.items():
            w2[k] = np.asarray(w2[k]) * fct
        p2 = f"{model_dir}/h2_{what.split()[0]}.nequip.pth"
        allegro_torch.export_nequip_pth(p2, cfg2, w2)
        r64 = util.oracle_run(dict(cfg2, model_dtype="float64"), w2, g2["cell"], g2["pos"], types, names)
        auto = util.run_pair(hip_lib, p2, g2["cell"], g2["pos"], types, names, options={"path": "fused"})
        assert auto["info"]["path"] == "fused_f32" and "float32 instance" in auto["info"]["arith_note"], (what, auto["info"])
        fm = np.abs(r64["forces"]).max()
        f32x = util.run_pair(hip_lib, p2, g2["cell"], g2["pos"], types, names, options={"path": "fused", "fused_arith": "f32"})
        np.testing.assert_allclose(auto["forces"], f32x["forces"], rtol=0, atol=1e-6 * fm)          # auto IS the float32 instance now ...
        assert np.abs(auto["forces"] - r64["forces"]).max() < 2e-4 * fm, what                       # ... and float32 carries the model (forces ~1e6 with the blown-up linear)
        if "beyond" in what:
            with pytest.raises(Exception, match="float16"):
                util.run_pair(hip_lib, p2, g2["cell"], g2["pos"], types, names, options={"path": "fused", "fused_arith": "f16x2"})
    # a sane model keeps f16x2 and says so
    path3, cfg3, types3, names3, ref3 = _model_S_case(model_dir, "cupd_S_note", ["Cu", "Pd"], g2["symbols"], g2["cell"], g2["pos"])
    ok = util.run_pair(hip_lib, path3, g2["cell"], g2["pos"], types3, names3)
    assert ok["info"]["path"] == "fused_f16x2" and "f16x2 kept" in ok["info"]["arith_note"], ok["info"]


def test_f16x2_backward_scale_is_per_centre_type(hip_lib, model_dir):
    """Energy scales that differ by six orders of magnitude between species (VERDICT r05 #3b): the backward pass of the f16x2 arithmetic runs scaled by a power of
    two PER CENTRE TYPE.  One Pd atom (scale 1e+3) in a Cu box (scale 1e-3): the forces on the Cu atoms farther than r_max from it come from Cu-centred edges
    only, and must be as close to the float64 oracle -- relative to THEIR size -- as the f32 instance's (one global power of two taken from the largest scale left
    those gradients in float16's subnormals: ~1e-2 relative)."""
    g = util.load_golden("CuPd-cubic-big_r5")
    symbols = ["Cu"] * len(g["symbols"]); symbols[0] = "Pd"
    cfg = model_file.model_S(type_names=["Cu", "Pd"], avg_num_neighbors=40.0)
    w = model_file.init_weights(cfg)
    w["scale"] = np.array([1e-3, 1e3])
    path = f"{model_dir}/h2_hetero_scale.nequip.pth"
    allegro_torch.export_nequip_pth(path, cfg, w)
    names = ["Cu", "Pd"]
    types = np.array([names.index(s_) + 1 for s_ in symbols], dtype=np.int32)
    r64 = util.oracle_run(dict(cfg, model_dtype="float64"), w, g["cell"], g["pos"], types, names)
    cell = np.diag(np.asarray(g["cell"])) if np.asarray(g["cell"]).ndim == 2 else np.asarray(g["cell"])
    d = np.asarray(g["pos"]) - np.asarray(g["pos"])[0]
    d -= cell * np.round(d / cell)
    far = np.linalg.norm(d, axis=1) > cfg["r_max"] + 0.1
    assert far.sum() > 50
    a = util.run_pair(hip_lib, path, g["cell"], g["pos"], types, names, options={"path": "fused", "fused_arith": "f32"})
    b = util.run_pair(hip_lib, path, g["cell"], g["pos"], types, names, options={"path": "fused", "fused_arith": "f16x2"})
    fm = np.abs(r64["forces"][far]).max()
    ea, eb = np.abs(a["forces"][far] - r64["forces"][far]).max() / fm, np.abs(b["forces"][far] - r64["forces"][far]).max() / fm
    print(f"Cu atoms beyond r_max of the Pd atom: max|dF| / max|F| f32 {ea:.3e}, f16x2 {eb:.3e} (max|F| there {fm:.3e}, overall {np.abs(r64['forces']).max():.3e})")
    assert eb < max(1.5 * ea, 2e-5), (ea, eb)
    c = util.run_pair(hip_lib, path, g["cell"], g["pos"], types, names)          # and auto keeps f16x2 on it
    assert c["info"]["path"] == "fused_f16x2", c["info"]


@pytest.mark.parametrize("depth", [1, 3])
def test_fused_latent_mlp_depth_1_and_3(hip_lib, model_dir, depth):
    """The fused family widened by one axis: allegro_mlp_hidden_layers_depth 1 and 3 (2 in /root/reference/tests/test_data/test_repro_allegro.yaml:94) --
    the two-body MLP behind the spline table and the latent MLP of every layer -- select the fused path (template parameter MD of k_fused, f16x2 instances) and
    match the float64 oracle like depth 2 does: 256-atom CuPd box, 2 types, 2 and 3 layers, against the layer-at-a-time kernels too; fused_arith=f32 (no such instance)
    falls back to them."""
    g = util.load_golden("CuPd-cubic-big_r5")
    for nl in (2, 3):
        path, cfg, types, names, ref = _model_S_case(model_dir, f"cupd_S_md{depth}_nl{nl}", ["Cu", "Pd"], g["symbols"], g["cell"], g["pos"], nl=nl, mlp_depth=depth)
        fused = util.run_pair(hip_lib, path, g["cell"], g["pos"], types, names)
        assert fused["info"]["path"] == "fused_f16x2"
        util.assert_close_to(fused, ref, 5e-4, what=f"fused, MLP depth {depth}, {nl} layers vs f64 oracle")
        err = np.abs(fused["forces"] - ref["forces"]).max()
        gen = util.run_pair(hip_lib, path, g["cell"], g["pos"], types, names, options={"path": "generic"})
        egen = np.abs(gen["forces"] - ref["forces"]).max()
        print(f"MLP depth {depth}, {nl} layers: max|dF| vs f64 oracle fused {err:.3e}, layer-at-a-time f32 {egen:.3e}")
        assert err < pc.F32EQ_DF * 2 and err < max(3.0 * egen, 1e-5)
    exact = util.run_pair(hip_lib, path, g["cell"], g["pos"], types, names, options={"fused_arith": "f32"})
    assert exact["info"]["path"] == "generic_f32"


def test_fused_two_body_table_matches_mlp(hip_lib, model_dir):
    """Default: the two-body embedding x0(d; type pair) comes from a per-pair cubic spline table built from the float64 MLP
    (512 intervals); option fused_tb=mlp evaluates the three linears in the kernel.  Both must sit at the same distance
    from the float64 oracle (3 types, 9 pair tables, ragged tiles; and the 2-type 256-atom box)."""
    for tag, tn in (("Cu2AgO4_r5", ["Cu", "Ag", "O"]), ("CuPd-cubic-big_r5", ["Cu", "Pd"])):
        g = util.load_golden(tag)
        path, cfg, types, names, ref = _model_S_case(model_dir, f"tb_{tag}", tn, g["symbols"], g["cell"], g["pos"])
        tab = util.run_pair(hip_lib, path, g["cell"], g["pos"], types, names, options={"path": "fused", "fused_tb": "table"})
        mlp = util.run_pair(hip_lib, path, g["cell"], g["pos"], types, names, options={"path": "fused", "fused_tb": "mlp"})
        util.assert_close_to(tab, ref, 5e-4, what=f"{tag} two-body table vs f64 oracle")
        et, em = np.abs(tab["forces"] - ref["forces"]).max(), np.abs(mlp["forces"] - ref["forces"]).max()
        assert et < max(2.0 * em, 1e-5), (et, em)
        np.testing.assert_allclose(tab["forces"], mlp["forces"], atol=1e-5)
        np.testing.assert_allclose(tab["eatom"], mlp["eatom"], atol=2e-5)


def test_fused_two_lammps_types_share_a_model_type(hip_lib, model_dir):
    """BASELINE config 3 style deck (`pair_coeff * * f Li P O O`): two LAMMPS types mapped onto one model type
    (pair_nequip_allegro.cpp:284-294) must give the forces of the plain 3-type run."""
    g = util.load_golden("Cu2AgO4_r5")
    path, cfg, types, names, ref = _model_S_case(model_dir, "cu2ago4_S_4t", ["Cu", "Ag", "O"], g["symbols"], g["cell"], g["pos"])
    base = util.run_pair(hip_lib, path, g["cell"], g["pos"], types, names, options={"path": "fused"})
    o_type = names.index("O") + 1
    types4 = types.copy()
    o_atoms = np.where(types == o_type)[0]
    types4[o_atoms[::2]] = len(names) + 1                    # every second oxygen becomes LAMMPS type 4, also named O
    split = util.run_pair(hip_lib, path, g["cell"], g["pos"], types4, names + ["O"], options={"path": "fused"})
    assert split["info"]["path"] in pc.FUSED_F32EQ
    np.testing.assert_allclose(split["forces"], base["forces"], atol=1e-6)
    np.testing.assert_allclose(split["eatom"], base["eatom"], atol=1e-6)
    np.testing.assert_allclose(split["pe"], base["pe"], rtol=1e-7)
    assert np.abs(split["forces"] - ref["forces"]).max() < pc.NORTH_STAR_DF


def test_fused_wide_tiles_65_to_128_neighbours(hip_lib, model_dir):
    """fcc Cu with r_max 6.1 A has 78 neighbours per atom: more than the 64-slot tile of the default 4-wave workgroup,
    so the 8-wave / 128-slot kernel instance runs; it must agree with the oracle like the narrow one."""
    g = util.load_golden("Cu-cubic_r15")
    reps = 3
    cell = g["cell"] * reps
    shifts = np.array([[i, j, k] for i in range(reps) for j in range(reps) for k in range(reps)], dtype=float)
    pos = np.concatenate([g["pos"] + s @ g["cell"] for s in shifts])
    rng = np.random.RandomState(3)
    pos = pos + rng.uniform(-0.05, 0.05, size=pos.shape)
    symbols = ["Cu"] * len(pos)
    nb = float(len(util.glue.brute_force_edges(cell, pos, 6.1)[0])) / len(pos)
    cfg = model_file.model_S(type_names=["Cu"], r_max=6.1, avg_num_neighbors=nb)
    w = model_file.init_weights(cfg)
    path = f"{model_dir}/cu61_S.nequip.pth"
    allegro_torch.export_nequip_pth(path, cfg, w)
    types = np.ones(len(pos), dtype=np.int32)
    ref = util.oracle_run(dict(cfg, model_dtype="float64"), w, cell, pos, types, ["Cu"])
    res = util.run_pair(hip_lib, path, cell, pos, types, ["Cu"], options={"path": "fused"})
    assert res["info"]["path"] in pc.FUSED_F32EQ
    assert 64 < res["info"]["max_degree"] <= 128
    util.assert_close_to(res, ref, 5e-4, what="wide-tile fused vs f64 oracle")
    assert np.abs(res["forces"] - ref["forces"]).max() < pc.NORTH_STAR_DF


def test_compute_allegro_outputs_on_the_fused_path(hip_lib, model_dir):
    """`compute allegro/atom forces 3 1` and `atomic_energy 1 0` served by the fused kernel's results."""
    from pair_allegro_amd.compute import ComputeAllegro
    from pair_allegro_amd.pair import PairAllegro, atom_from_rank_system, list_from_rank_system
    g = util.load_golden("CuPd-cubic-big_r5")
    path, cfg, types, names, ref = _model_S_case(model_dir, "cupd_S_cmp", ["Cu", "Pd"], g["symbols"], g["cell"], g["pos"])
    pair = PairAllegro(me=0, nprocs=1, lib=hip_lib, quiet=True)
    pair.settings([])
    pair.coeff(["*", "*", path] + list(names), ntypes=len(names))
    cf = ComputeAllegro(["f", "all", "allegro/atom", "forces", "3", "1"], pair)
    ce = ComputeAllegro(["e", "all", "allegro/atom", "atomic_energy", "1", "0"], pair)
    cv = ComputeAllegro(["v", "all", "allegro", "virial", "9"], pair)
    pair.model.set_option("path", "fused")
    rs = lmp_like.build_rank_system(g["cell"], g["pos"], types, 6.0)
    atom = atom_from_rank_system(rs, len(names))
    pair.compute(atom, list_from_rank_system(rs))
    assert pair.model.last_path in pc.FUSED_F32EQ
    arr = cf.compute_peratom(rs.nlocal, rs.nall).copy()
    np.add.at(arr, rs.tag[rs.nlocal:] - 1, cf.pack_reverse_comm(rs.nghost, rs.nlocal).reshape(-1, 3))
    forces = np.zeros_like(ref["forces"])
    forces[rs.tag[: rs.nlocal] - 1] = arr[: rs.nlocal]
    assert np.abs(forces - ref["forces"]).max() < pc.NORTH_STAR_DF
    e = ce.compute_peratom(rs.nlocal, rs.nall)[: rs.nlocal, 0]
    np.testing.assert_allclose(e, ref["eatom"][rs.tag[: rs.nlocal] - 1], atol=5e-4)
    xx, yy, zz, xy, xz, yz = ref["virial"]
    np.testing.assert_allclose(cv.compute_vector(rs.nlocal).reshape(3, 3), [[xx, xy, xz], [xy, yy, yz], [xz, yz, zz]],
                               atol=20 * 5e-4 * len(g["pos"]) ** 0.5)
    pair.model.close()


def test_fused_multi_rank_and_ragged_tiles(hip_lib, model_dir):
    """2x2x1 ranks; Cu2AgO4 (7 atoms, ragged degrees, 3 types, triclinic) exercises partial tiles."""
    g = util.load_golden("Cu2AgO4_r5")
    path, cfg, types, names, ref = _model_S_case(model_dir, "cu2ago4_S", ["Cu", "Ag", "O"], g["symbols"], g["cell"], g["pos"])
    fused = util.run_pair(hip_lib, path, g["cell"], g["pos"], types, names, options={"path": "fused"})
    util.assert_close_to(fused, ref, 5e-4, what="Cu2AgO4 fused")
    g2 = util.load_golden("CuPd-cubic-big_r5")
    path, cfg, types, names, ref = _model_S_case(model_dir, "cupd_S_mr", ["Cu", "Pd"], g2["symbols"], g2["cell"], g2["pos"])
    fused = util.run_pair(hip_lib, path, g2["cell"], g2["pos"], types, names, grid=(2, 2, 1), options={"path": "fused"})
    util.assert_close_to(fused, ref, 5e-4, what="CuPd fused 2x2x1")


def test_fused_falls_back_when_degree_exceeds_tile(hip_lib, model_dir):
    """r_max 15 A on the 4-atom Cu cell: ~1200 edges per atom > 128 slots -> auto picks the generic path,
    path=fused is a clean error."""
    g = util.load_golden("Cu-cubic_r15")
    cfg = model_file.model_S(type_names=["Cu"], r_max=15.0, avg_num_neighbors=1204.0)
    w = model_file.init_weights(cfg)
    path = f"{model_dir}/cu15_S.nequip.pth"
    allegro_torch.export_nequip_pth(path, cfg, w)
    types = np.ones(4, dtype=np.int32)
    res = util.run_pair(hip_lib, path, g["cell"], g["pos"], types, ["Cu"])
    assert res["info"]["path"] == "generic_f32"
    with pytest.raises(Exception, match="fused path unavailable"):
        util.run_pair(hip_lib, path, g["cell"], g["pos"], types, ["Cu"], options={"path": "fused"})


def test_full_size_properties_10k(hip_lib, model_dir):
    """BASELINE configs[1] (10 648-atom Si) at full size through size-independent properties:
    net force = 0 (Newton's third law), PE = sum of per-atom energies, fused == generic,
    rigid rotation leaves energies invariant and rotates forces, translation invariance."""
    cfg = model_file.model_S()
    w = model_file.init_weights(cfg)
    path = f"{model_dir}/si_S.nequip.pth"
    allegro_torch.export_nequip_pth(path, cfg, w)
    cell, pos, types = lmp_like.diamond_si(11)
    a = util.run_pair(hip_lib, path, cell, pos, types, ["Si"], options={"path": "fused"})
    b = util.run_pair(hip_lib, path, cell, pos, types, ["Si"], options={"path": "generic"})
    assert a["info"]["path"] in pc.FUSED_F32EQ and b["info"]["path"] == "generic_f32"
    assert np.abs(a["forces"].sum(0)).max() < 1e-6 * len(pos) ** 0.5
    np.testing.assert_allclose(a["eatom"].sum(), a["pe"], rtol=1e-10)
    assert np.abs(a["forces"] - b["forces"]).max() < 5e-5
    np.testing.assert_allclose(a["pe"], b["pe"], rtol=1e-6)
    np.testing.assert_allclose(a["virial"], b["virial"], atol=2e-3 * len(pos) ** 0.5, rtol=1e-4)
    shifted = lmp_like.wrap(cell, pos + np.array([1.234, -0.77, 3.1]))
    c = util.run_pair(hip_lib, path, cell, shifted, types, ["Si"], options={"path": "fused"})
    np.testing.assert_allclose(c["pe"], a["pe"], rtol=1e-6)
    assert np.abs(c["forces"] - a["forces"]).max() < 5e-5


def test_narrower_models_run_fused_zero_padded(hip_lib, model_dir):
    """VERDICT r05 missing #3: the widths are free hyper-parameters of /root/reference/tests/test_data/test_repro_allegro.yaml:89-99 and the reference executes any archive; here everything but
    S = W = 64, U = 32, R = 32 fell to the layer-at-a-time kernels at ~12 x the time.  A model NARROWER than a fused kernel's fixed widths now runs on it zero-padded (csrc/model_io.cpp:
    pad_host_model: padded features are exact zeros through every linear, SiLU and tensor product): 48 scalars, 16 tensor features, MLP width 40, read-out width 24 -- 2 and 3 layers, MLP depth 2 and 3 --
    takes the fused path under the default options and matches the float64 oracle and the layer-at-a-time kernels like the full-width model does."""
    g = util.load_golden("CuPd-cubic-big_r5")
    for nl, depth in ((2, 2), (3, 2), (2, 3)):
        path, cfg, types, names, ref = _model_S_case(model_dir, f"cupd_narrow_nl{nl}_md{depth}", ["Cu", "Pd"], g["symbols"], g["cell"], g["pos"], nl=nl, mlp_depth=depth,
                                                     num_scalar_features=48, num_tensor_features=16, mlp_width=40, readout_width=24)
        fused = util.run_pair(hip_lib, path, g["cell"], g["pos"], types, names)
        assert fused["info"]["path"] == pc.FUSED_DEFAULT, fused["info"]
        util.assert_close_to(fused, ref, 5e-4, what=f"narrow model, {nl} layers, depth {depth}: fused vs f64 oracle")
        gen = util.run_pair(hip_lib, path, g["cell"], g["pos"], types, names, options={"path": "generic"})
        err, egen = np.abs(fused["forces"] - ref["forces"]).max(), np.abs(gen["forces"] - ref["forces"]).max()
        print(f"narrow model nl={nl} depth={depth}: max|dF| vs f64 oracle fused {err:.3e}, layer-at-a-time f32 {egen:.3e}")
        assert err < max(3.0 * egen, 1e-5)
    # wider than the kernel: still the layer-at-a-time path
    path, cfg, types, names, ref = _model_S_case(model_dir, "cupd_too_wide", ["Cu", "Pd"], g["symbols"], g["cell"], g["pos"], num_scalar_features=80)
    res = util.run_pair(hip_lib, path, g["cell"], g["pos"], types, names)
    assert res["info"]["path"] == "generic_f32"
    util.assert_close_to(res, ref, 5e-4, what="S = 80: layer-at-a-time kernels")


def test_auto_on_the_device_resident_path_reports_once_and_continues_on_float32(hip_lib, model_dir, monkeypatch):
    """The `_dev` entry points never wait for the device, so a float16-range alarm of one evaluation is met at the NEXT one.  Under fused_arith=auto: that next
    call returns AHIP_ERR_STATE once -- the earlier forces were invalid and have been handed out --, switches the model to the f32 instance, and every call after it evaluates
    normally (the reference's float32 evaluates such a model, /root/reference/pair_nequip_allegro.cpp:267-270 only ever relaxes precision).  The first-evaluation self-check
    is switched off here; with it on the very first call already lands on the f32 instance and nothing is ever reported (second half)."""
    import torch
    from pair_allegro_amd import capi, md
    g, cfg, w, path, names, types = _blown_up_model(model_dir)
    ref32 = util.run_pair(hip_lib, path, g["cell"], g["pos"], types, names, options={"path": "fused", "fused_arith": "f32"})
    dev = torch.device("cuda", 0)
    box = np.diag(np.asarray(g["cell"]))
    for selfcheck in (False, True):
        if selfcheck:
            monkeypatch.delenv("AHIP_NO_ARITH_SELFCHECK", raising=False)
        else:
            monkeypatch.setenv("AHIP_NO_ARITH_SELFCHECK", "1")
        model = capi.Model(path, 0, hip_lib)
        sim = md.Simulation(md.HipBackend(model, [63.5, 106.4]), box, cfg["r_max"], 1.0, np.asarray(g["pos"]), (types - 1).astype(np.int32),
                            np.zeros_like(np.asarray(g["pos"])), dev, grid=(1, 1, 1), rank=0, dist=None, dt=0.001)
        sim.rebuild()
        ev = torch.zeros(7, dtype=torch.float64, device=dev)

        def evaluate():
            sim.f.zero_()
            sim.backend.compute(sim.x, sim.mtype, sim.f, sim.nlocal, ev)
            torch.cuda.synchronize()
            f = torch.zeros((sim.nlocal, 3), dtype=torch.float64, device=dev)
            f.index_add_(0, torch.arange(sim.nlocal, device=dev), sim.f[: sim.nlocal])
            if sim.nall > sim.nlocal:
                f.index_add_(0, sim._ghost_src, sim.f[sim.nlocal:])
            out = np.zeros((sim.nlocal, 3)); out[sim.tag[: sim.nlocal].cpu().numpy()] = f.cpu().numpy()
            return out
        if not selfcheck:
            f1 = evaluate()                                       # f16x2: overflows, non-finite forces, alarm raised -- nobody has looked yet
            assert not np.isfinite(f1).all() and model.last_path == "fused_f16x2"
            with pytest.raises(capi.AhipError) as ei:             # the next call reports the EARLIER evaluation, once
                evaluate()
            assert "EARLIER evaluation" in str(ei.value.msg) and "float32 instance" in model.arith_note
        f2 = evaluate()
        assert model.last_path == "fused_f32" and np.isfinite(f2).all()
        # (forces of ~1e6 from float32 sums in another neighbour order than the host-path reference: a few 1e-4 of max|F| apart)
        np.testing.assert_allclose(f2, ref32["forces"], rtol=0, atol=2e-3 * np.abs(ref32["forces"]).max())
        f3 = evaluate()                                           # and stays there
        assert model.last_path == "fused_f32"
        np.testing.assert_allclose(f3, f2, rtol=0, atol=1e-9 * np.abs(f2).max())
        model.close()
