"""Kernel LOGIC parity on the CPU: the sync-free HIP sources compiled against the host stand-in
runtime (tests/host_emu) versus the golden vectors of the float64 torch oracle.  These tests do not
replace the GPU parity tests (tests/test_gpu_parity.py); they make the same sources debuggable in a
container without a GPU."""
import numpy as np
import pytest

import parity_cases as pc
import util


@pytest.mark.parametrize("tag", util.GOLDEN_TAGS)
@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_golden_single_rank(emu_lib, model_dir, tag, dtype):
    if tag == "Cu-cubic_r15" and dtype == "float32":
        pytest.skip("covered in float64; keeps the CPU suite short")
    res, g = pc.check_golden(emu_lib, model_dir, tag, dtype)
    assert res["info"]["path"] == ("generic_f64" if dtype == "float64" else "generic_f32")
    pc.check_edges_vs_brute_force(res, g)


@pytest.mark.parametrize("grid", [(2, 1, 1), (2, 2, 1)])
@pytest.mark.parametrize("tag", ["CuPd-cubic-big_r5", "Cu-cubic_r5"])
def test_golden_multi_rank(emu_lib, model_dir, tag, grid):
    """n_rank 2 and 4 like /root/reference/tests/test_python_repro_allegro.py:44-47,70-77:
    strictly local model => per-rank partial forces/energies/virials add up to the single-rank result."""
    res, g = pc.check_golden(emu_lib, model_dir, tag, "float64", grid=grid)
    pc.check_edges_vs_brute_force(res, g)


def test_chunked_equals_unchunked(emu_lib, model_dir):
    a, _ = pc.check_golden(emu_lib, model_dir, "Si64_r5", "float64")
    b, _ = pc.check_golden(emu_lib, model_dir, "Si64_r5", "float64", options={"chunk_edges": 100})
    np.testing.assert_allclose(a["forces"], b["forces"], atol=1e-12)
    np.testing.assert_allclose(a["virial"], b["virial"], atol=1e-10)


def test_neighbor_order_irrelevant(emu_lib, model_dir):
    pc.check_golden(emu_lib, model_dir, "Cu2AgO4_r5", "float64", shuffle_seed=3)


def test_precision_option_float64_on_float32_model(emu_lib, model_dir):
    pc.check_golden(emu_lib, model_dir, "Si64_r5", "float32", options={"precision": "float64"})
