"""GPU parity tests proper: liballegro_hip.so on a real MI355X, through the C-ABI, against the
golden vectors of the float64 oracle (committed) and, at full size, against size-independent
properties."""
import numpy as np
import pytest

import parity_cases as pc
import util

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag", util.GOLDEN_TAGS)
@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_golden_generic(hip_lib, model_dir, tag, dtype):
    res, g = pc.check_golden(hip_lib, model_dir, tag, dtype, options={"path": "generic"})
    assert res["info"]["path"] == ("generic_f64" if dtype == "float64" else "generic_f32")
    pc.check_edges_vs_brute_force(res, g)


@pytest.mark.parametrize("grid", [(2, 1, 1), (2, 2, 1)])
@pytest.mark.parametrize("tag", ["CuPd-cubic-big_r5", "Cu-cubic_r5"])
def test_golden_multi_rank(hip_lib, model_dir, tag, grid):
    res, g = pc.check_golden(hip_lib, model_dir, tag, "float32", grid=grid)
    pc.check_edges_vs_brute_force(res, g)


def test_chunked_equals_unchunked(hip_lib, model_dir):
    a, _ = pc.check_golden(hip_lib, model_dir, "CuPd-cubic-big_r5", "float64", options={"path": "generic"})
    b, _ = pc.check_golden(hip_lib, model_dir, "CuPd-cubic-big_r5", "float64", options={"path": "generic", "chunk_edges": 1500})
    np.testing.assert_allclose(a["forces"], b["forces"], atol=1e-12)
