"""SURVEY 8f-1 on the device: the one-command pinning procedure tests/pin_real_model.py run WITHOUT --emu, i.e. the archive's
own TorchScript graph through the libtorch harness (the reference's call sequence, /root/reference/pair_nequip_allegro.cpp:214-232, 409-430)
against the converted weights on liballegro_hip.so -- converter -> model_io.cpp -> fused kernels:
  (i)  the oracle's own export (it already carries the allegro_hip.bin section), model S on the fused kernel k_fused;
  (ii) an archive WITHOUT that section whose parameters carry allegro-style names (the synthetic tree of tests/test_convert_nequip.py) and whose
       graph is the oracle module with the same weights: pin_real_model converts it by the name rules and compares the two evaluations, on the
       reference YAML's shape (l_max = 2, 32 tensor features, 3 layers -> k_fused_lx).
Parity stays UNPINNED against a genuine nequip-compile archive (none exists here); what this pins is the route every such archive will take."""
import json
import os
import subprocess
import sys
from typing import Dict

import numpy as np
import pytest
import torch

import util
import parity_cases as pc  # noqa: E402
from oracle import allegro_torch
from pair_allegro_amd import model_file
from test_convert_nequip import _nequip_style_archive, _Box          # the allegro-named parameter tree

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _xyz(path, g):
    with open(path, "w") as f:
        f.write(f"{len(g['pos'])}\n")
        f.write('Lattice="' + " ".join(repr(float(v)) for v in np.asarray(g["cell"]).reshape(9)) + '" Properties=species:S:1:pos:R:3\n')
        for s, p in zip(g["symbols"], g["pos"]):
            f.write(f"{s} {float(p[0])!r} {float(p[1])!r} {float(p[2])!r}\n")


def _pin(args):
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "cpu_baseline"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        pytest.skip("libtorch harness not buildable here")
    return subprocess.run([sys.executable, os.path.join(ROOT, "tests", "pin_real_model.py")] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=ROOT)


def test_own_export_is_pinned_on_the_fused_kernel(hip_lib, tmp_path):
    g = util.load_golden("CuPd-cubic-big_r5")
    nb = float(len(util.glue.brute_force_edges(g["cell"], g["pos"], 5.0)[0])) / len(g["pos"])
    cfg = model_file.model_S(type_names=["Cu", "Pd"], avg_num_neighbors=nb, model_dtype="float32")
    w = model_file.init_weights(cfg)
    pth, xyz, out = str(tmp_path / "own.nequip.pth"), str(tmp_path / "s.xyz"), str(tmp_path / "r.json")
    allegro_torch.export_nequip_pth(pth, cfg, w)
    _xyz(xyz, g)
    r = _pin([pth, xyz, "--json", out])
    assert r.returncode == 0, r.stdout.decode() + r.stderr.decode()[-2000:]
    res = json.load(open(out))
    assert res["pinned"] and res["rows"][0]["kernel_path"] in pc.FUSED_F32EQ
    assert res["rows"][0]["max_dF"] < 1e-4 and res["rows"][0]["max_dEi"] < 5e-4


class _TopWithGraph(torch.nn.Module):
    """allegro-named parameters (what the converter reads) + the oracle module (what forward runs): an archive shaped like a
    nequip-compile output -- a graph, named parameters, the five metadata members, no allegro_hip.bin."""

    def __init__(self, model, graph):
        super().__init__()
        self.model = model
        self.graph = graph

    def forward(self, data: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        return self.graph(data)


def test_converted_allegro_named_archive_is_pinned_on_the_wide_fused_kernel(hip_lib, tmp_path):
    g = util.load_golden("Cu2AgO4_r5")
    cfg = model_file.model_L(type_names=["Cu", "Ag", "O"], num_tensor_features=32, avg_num_neighbors=30.0, model_dtype="float32")     # the reference YAML's shape
    w = {k: v.astype(np.float32).astype(np.float64) for k, v in model_file.init_weights(cfg).items()}
    for k in range(1, cfg["num_layers"] + 1):                 # residual coefficients the upstream parametrisation can express
        sgm = 0.3 + 0.2 * k
        w[f"l{k}.res"] = (np.array([1.0, sgm]) / np.sqrt(1.0 + sgm * sgm)).astype(np.float32).astype(np.float64)
    tree_only = str(tmp_path / "tree.pth")
    _nequip_style_archive(tree_only, cfg, w)                  # builds the named tree; re-read below to wrap it around a real graph
    tree = torch.jit.load(tree_only).model
    src, xyz, out, rules = str(tmp_path / "real.nequip.pth"), str(tmp_path / "s.xyz"), str(tmp_path / "r.json"), str(tmp_path / "rules.json")
    # float32 storage rounds the residual parameters through a logit/sigmoid pair: the graph gets the weights the converter will recover
    from pair_allegro_amd.tools import convert_nequip
    cfg_c, w_c, _ = convert_nequip.convert(tree_only, None, None, None)
    graph = torch.jit.script(allegro_torch.build(dict(cfg, model_dtype="float32"), w_c).eval())
    torch.jit.save(torch.jit.script(_TopWithGraph(tree, graph)), src, _extra_files=model_file.reference_metadata(cfg))
    json.dump({"ignore": convert_nequip.DEFAULT_IGNORE + [r"^graph\..*"]}, open(rules, "w"))
    _xyz(xyz, g)
    from pair_allegro_amd import capi
    with pytest.raises(capi.AhipError, match="convert_nequip"):           # no weight section: the library names the converter
        capi.Model(src, 0, hip_lib)
    r = _pin([src, xyz, "--map", rules, "--json", out])
    assert r.returncode == 0, r.stdout.decode() + r.stderr.decode()[-2000:]
    res = json.load(open(out))
    assert res["pinned"] and res["rows"][0]["kernel_path"].startswith("fused")
    assert res["rows"][0]["max_dF"] < 1e-4
