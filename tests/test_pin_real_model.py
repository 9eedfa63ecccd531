"""The pinning procedure for a genuine nequip archive (tests/pin_real_model.py), exercised on the oracle's own export: the
archive's TorchScript graph through the libtorch harness (the reference's call sequence) against the converted weights on the host
emulation of the kernels; and a deliberately mis-converted file is diagnosed by the convention table."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import util
from oracle import allegro_torch
from pair_allegro_amd import model_file
from pair_allegro_amd.tools import convert_nequip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _xyz(path, g):
    with open(path, "w") as f:
        f.write(f"{len(g['pos'])}\n")
        f.write('Lattice="' + " ".join(repr(float(v)) for v in np.asarray(g["cell"]).reshape(9)) + '" Properties=species:S:1:pos:R:3\n')
        for s, p in zip(g["symbols"], g["pos"]):
            f.write(f"{s} {float(p[0])!r} {float(p[1])!r} {float(p[2])!r}\n")


@pytest.fixture(scope="module")
def case(tmp_path_factory):
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "cpu_baseline"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        pytest.skip("libtorch harness not buildable here")
    d = tmp_path_factory.mktemp("pin")
    g = util.load_golden("Cu2AgO4_r5")
    cfg = model_file.model_S(model_dtype="float64", type_names=["Cu", "Ag", "O"], num_scalar_features=16, num_tensor_features=8, mlp_width=16,
                             readout_width=8, l_max=2, num_layers=2, avg_num_neighbors=30.0)
    w = model_file.init_weights(cfg)
    pth = str(d / "own.nequip.pth")
    allegro_torch.export_nequip_pth(pth, cfg, w)
    xyz = str(d / "s.xyz")
    _xyz(xyz, g)
    return d, cfg, w, pth, xyz


def _run(args):
    return subprocess.run([sys.executable, os.path.join(ROOT, "tests", "pin_real_model.py")] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=ROOT)


def test_own_export_is_pinned(case):
    d, cfg, w, pth, xyz = case
    out = str(d / "r.json")
    r = _run([pth, xyz, "--emu", "--json", out])
    assert r.returncode == 0, r.stdout.decode() + r.stderr.decode()[-2000:]
    assert b"PINNED" in r.stdout
    res = json.load(open(out))
    assert res["pinned"] and res["rows"][0]["max_dF"] < 1e-9


def test_wrong_residual_convention_is_diagnosed(case):
    """The archive's graph keeps the true residual coefficients; the weight section is written as if the OTHER residual form had been
    meant -- the as-converted evaluation disagrees and the table names item 8."""
    d, cfg, w, pth, xyz = case
    w2 = {k: v.copy() for k, v in w.items()}
    for k in range(1, cfg["num_layers"] + 1):
        a, b = w[f"l{k}.res"]
        s = b * b / (a * a + b * b)
        w2[f"l{k}.res"] = np.array([1.0, s]) / np.sqrt(1.0 + s * s)
        np.testing.assert_allclose([np.sqrt(1 - s), np.sqrt(s)], np.array([a, b]) / np.hypot(a, b), atol=1e-12)
    bad = str(d / "bad.nequip.pth")
    convert_nequip.write_with_blob(pth, bad, cfg, w2)
    r = _run([bad, xyz, "--emu"])
    assert r.returncode == 1
    txt = r.stdout.decode()
    assert "MISMATCH" in txt and "-> convention '8: residual update" in txt, txt
