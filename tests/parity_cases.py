"""Parity checks shared by the CPU-emulation tests (-m "not gpu") and the GPU tests (-m gpu).
Every check drives the library through PairAllegro -> C-ABI, like the reference's test_repro."""
import numpy as np

from oracle import glue

import util

TOL = {"float32": 5e-4, "float64": 1e-8}          # /root/reference/tests/conftest.py:113
NORTH_STAR_DF = 1e-4                               # BASELINE.json: max|dF| < 1e-4 eV/A (float32 compute)
F32EQ_DF = 5e-6                                    # a split arithmetic (f16x2, bf16x3) is float32-EQUIVALENT: max|dF| vs the float64 oracle on the 10 648-atom Si box within 5e-6 (f32 fmaf chains: 1.6e-6 .. 3.3e-6)
# kernel paths whose arithmetic is float32 or float32-equivalent: exact fmaf chains; two float16 terms per operand (csrc/fused_h.h); three bf16 terms -- everything but tf32eq
FUSED_F32EQ = ("fused_f32", "fused_f16x2", "fused_bf16x3")
FUSED_DEFAULT = "fused_f16x2"                      # what fused_arith=auto selects on every fused kernel (k_fused, k_fused_lx2, k_fused_lx) when the model file has allow_tf32 = 0


def check_golden(lib, model_dir, tag, dtype, grid=(1, 1, 1), options=None, shuffle_seed=None):
    g = util.load_golden(tag)
    path, cfg, w = util.golden_model(g, model_dir, dtype)
    types, names = util.lammps_types(g)
    res = util.run_pair(lib, path, g["cell"], g["pos"], types, names, grid=grid, options=options, shuffle_seed=shuffle_seed)
    tol = TOL[dtype]
    if tag.startswith("aspirin"):
        tol *= 23                                  # conftest.py:114-116 (kcal/mol-scale forces)
    util.assert_close_to(res, g, tol, what=f"{tag} {dtype} grid={grid}")
    if dtype == "float32" and not tag.startswith("aspirin"):
        assert np.abs(res["forces"] - g["forces"]).max() < NORTH_STAR_DF
    return res, g


def check_edges_vs_brute_force(res, g):
    """Edge multiset == brute-force periodic search at r_max
    (/root/reference/tests/test_python_repro_allegro.py:259-286)."""
    bi, bj, bd = glue.brute_force_edges(g["cell"], g["pos"], g["cfg"]["r_max"])
    i, j, d = res["edges"]
    assert len(i) == len(bi) == int(g["nedges"])
    mine = sorted(zip(i.tolist(), j.tolist()))
    ref = sorted(zip(bi.tolist(), bj.tolist()))
    assert mine == ref
    np.testing.assert_allclose(np.sort(d), np.sort(bd), rtol=0, atol=1e-5)
