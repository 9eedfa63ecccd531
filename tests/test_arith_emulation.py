"""CPU: the split arithmetics of the fused kernel (f16x2: csrc/fused_h.h; bf16x3: csrc/fused.hip linear_b), emulated operand by operand on the
float32 torch oracle (util.split_emulation), sit at the same distance from the float64 oracle as plain float32 -- the claim behind the name
"float32-equivalent" and behind `fused_arith=auto` choosing f16x2.  The kernels themselves are compared with the oracle in the gpu suite
(test_gpu_fused.py, test_gpu_soak.py, pair_allegro_amd/tools/arith_check.py)."""
import numpy as np
import pytest

import util
from pair_allegro_amd import lmp_like, model_file


def _errors(cfg, w, cell, pos, types, names, modes):
    ref = util.oracle_run(dict(cfg, model_dtype="float64"), w, cell, pos, types, names)
    out = {}
    for mode in modes:
        with util.split_emulation(mode):
            r = util.oracle_run(dict(cfg, model_dtype="float32"), w, cell, pos, types, names)
        out[mode] = (np.abs(r["forces"] - ref["forces"]).max(), np.abs(r["eatom"] - ref["eatom"]).max())
    return out, np.abs(ref["forces"]).max()


def test_split_arithmetics_are_float32_equivalent_on_si():
    """512-atom Si box, model S: max|dF| of f16x2 and of bf16x3 within 1.5 x that of float32 (10 648 atoms, same script: 2.87e-6 / 2.81e-6 / 2.81e-6)."""
    cfg = model_file.model_S()
    w = model_file.init_weights(cfg)
    cell, pos, types = lmp_like.diamond_si(4)
    err, fmax = _errors(cfg, w, cell, pos, types, ["Si"], ("f32", "f16x2", "bf16x3"))
    print({k: f"{v[0]:.3e}" for k, v in err.items()}, f"max|F| {fmax:.3f}")
    assert err["f16x2"][0] < 1.5 * err["f32"][0] and err["bf16x3"][0] < 1.5 * err["f32"][0]
    assert err["f16x2"][0] < 5e-6 and err["f16x2"][1] < 5e-6


@pytest.mark.parametrize("scale", [1e-3, 1e3])
def test_f16x2_needs_no_care_for_the_energy_scale_in_this_range(scale):
    """The backward pass is linear in scale[type] / sqrt(avg_num_neighbors); unscaled, the f16x2 split of the gradients keeps float32 accuracy for
    energy scales 1e-3 .. 1e3 of the usual one (below ~1e-5 it degrades: 1.6e-4 relative at 1e-5) -- the kernel does not rely on that: it scales the
    backward pass by a power of two that brings the upstream gradient to O(1) (FusedArgs::bscale)."""
    cfg = model_file.model_S()
    w = model_file.init_weights(cfg)
    w["scale"] = np.asarray(w["scale"]) * scale
    cell, pos, types = lmp_like.diamond_si(3)
    err, fmax = _errors(cfg, w, cell, pos, types, ["Si"], ("f32", "f16x2"))
    assert err["f16x2"][0] / fmax < max(1.5 * err["f32"][0] / fmax, 2e-5)
