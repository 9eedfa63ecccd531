"""SURVEY 8f-1 groundwork: the converter from a `nequip-compile`-style TorchScript archive (no `allegro_hip.bin` member) to a
file liballegro_hip.so loads.  A synthetic archive -- `torch.jit.save` of a module tree carrying allegro-style parameter names,
with the five metadata members the reference reads (pair_nequip_allegro.cpp:214-220) -- goes through
`python -m pair_allegro_amd.tools.convert_nequip`; the converted file must evaluate exactly like the natively exported model,
and an unknown tensor must stop the conversion with its name."""
import os
import subprocess
import sys
from typing import Dict

import numpy as np
import pytest
import torch

import util
from oracle import allegro_torch
from pair_allegro_amd import lmp_like, model_file
from pair_allegro_amd.tools import convert_nequip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _Box(torch.nn.Module):
    """A module that only carries named tensors (parameters / buffers / children)."""

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return x


class _Top(torch.nn.Module):
    def __init__(self, model):
        super().__init__()
        self.model = model

    def forward(self, data: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        return data


def _nequip_style_archive(path, cfg, w, extra_tensor=False, with_avg=True):
    t = lambda a: torch.nn.Parameter(torch.tensor(np.asarray(a), dtype=torch.float32), requires_grad=False)
    model = _Box()
    se = _Box(); se.mlp = _Box(); se.bessel = _Box()
    for k in range(cfg["mlp_depth"] + 1):
        se.mlp.register_parameter(f"_weight_{k}", t(w[f"tb.w{k}"]))
    se.bessel.register_buffer("bessel_weights", torch.arange(1, cfg["num_bessels"] + 1, dtype=torch.float32) * 3.14159)
    model.scalar_embed = se
    te = _Box(); te.env_embed = _Box(); te.env_embed.register_parameter("weight", t(w["emb.w"]))
    model.tensor_embed = te
    al = _Box()
    NL = cfg["num_layers"]
    al.env_embed_mlps = torch.nn.ModuleList(); al.tps = torch.nn.ModuleList(); al.latents = torch.nn.ModuleList(); al.linears = torch.nn.ModuleList()
    for k in range(1, NL + 1):
        b = _Box(); b.register_parameter("_weight_0", t(w[f"l{k}.env"])); al.env_embed_mlps.append(b)
        b = _Box(); b.register_parameter("path_weights", t(w[f"l{k}.tp"])); al.tps.append(b)
        b = _Box()
        for j in range(cfg["mlp_depth"] + 1):
            b.register_parameter(f"_weight_{j}", t(w[f"l{k}.lat.w{j}"]))
        al.latents.append(b)
        if k < NL:
            b = _Box(); b.register_parameter("weight", t(w[f"l{k}.mix"])); al.linears.append(b)
    sg = np.array([w[f"l{k}.res"][1] / w[f"l{k}.res"][0] for k in range(1, NL + 1)])      # upstream form: (alpha, beta) = (1, s) / sqrt(1 + s^2), s = sigmoid(p)
    al.register_parameter("_latent_resnet_update_params", t(np.log(sg / (1.0 - sg))))
    if with_avg:
        al.register_buffer("avg_num_neighbors", torch.tensor(float(cfg["avg_num_neighbors"])))
    model.allegro = al
    ro = _Box()
    for k in range(cfg["readout_depth"] + 1):
        ro.register_parameter(f"_weight_{k}", t(w[f"out.w{k}"]))
    model.edge_readout = ro
    ss = _Box(); ss.register_parameter("scales", t(w["scale"])); ss.register_parameter("shifts", t(w["shift"]))
    model.per_type_energy_scale_shift = ss
    if extra_tensor:
        model.register_parameter("mystery_gate", t(np.ones(3)))
    torch.jit.save(torch.jit.script(_Top(model)), path, _extra_files=model_file.reference_metadata(cfg))


def test_convert_synthetic_archive_and_load_it(emu_lib, tmp_path):
    cfg = model_file.model_S(type_names=["Cu", "Ag", "O"], per_edge_type_cutoff=[[5.0, 4.5, 4.0], [4.5, 5.0, 4.2], [4.0, 4.2, 4.8]],
                             num_scalar_features=16, num_tensor_features=8, mlp_width=16, readout_width=8, num_layers=3, l_max=2,
                             avg_num_neighbors=37.0)
    w = {k: v.astype(np.float32).astype(np.float64) for k, v in model_file.init_weights(cfg).items()}     # what float32 storage keeps
    for k in range(1, cfg["num_layers"] + 1):                 # residual coefficients the upstream parametrisation can express
        sgm = 0.3 + 0.2 * k
        w[f"l{k}.res"] = np.array([1.0, sgm]) / np.sqrt(1.0 + sgm * sgm)
    src, dst = str(tmp_path / "real.nequip.pth"), str(tmp_path / "converted.nequip.pth")
    _nequip_style_archive(src, cfg, w)
    # the library refuses the unconverted file with a message that names the converter
    from pair_allegro_amd import capi
    with pytest.raises(capi.AhipError, match="convert_nequip"):
        capi.Model(src, 0, emu_lib)
    r = subprocess.run([sys.executable, "-m", "pair_allegro_amd.tools.convert_nequip", src, dst],          # avg_num_neighbors: the archive's buffer
                       cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()
    assert b"PARITY UNPINNED" in r.stderr
    assert "converted, %d tensors mapped" % len(model_file.tensor_shapes(cfg)) in r.stdout.decode()
    cfg2, w2 = model_file.load(dst)
    for key in ("l_max", "num_layers", "num_scalar_features", "num_tensor_features", "mlp_depth", "mlp_width", "readout_width", "type_names"):
        assert cfg2[key] == cfg[key], key
    assert np.allclose(np.asarray(cfg2["per_edge_type_cutoff"]), np.asarray(cfg["per_edge_type_cutoff"]))
    assert abs(cfg2["avg_num_neighbors"] - 37.0) < 1e-6
    for k in w:
        np.testing.assert_allclose(w2[k], w[k], rtol=1e-6, atol=1e-7, err_msg=k)
    # the converted file evaluates like the natively exported model (float64 emulation of the same kernels)
    g = util.load_golden("Cu2AgO4_r5")
    types, names = util.lammps_types(g)
    native = str(tmp_path / "native.nequip.pth")
    allegro_torch.export_nequip_pth(native, dict(cfg, model_dtype="float32"), w)
    a = util.run_pair(emu_lib, dst, g["cell"], g["pos"], types, names, options={"precision": "float64"})
    b = util.run_pair(emu_lib, native, g["cell"], g["pos"], types, names, options={"precision": "float64"})
    np.testing.assert_allclose(a["forces"], b["forces"], atol=1e-6)
    np.testing.assert_allclose(a["pe"], b["pe"], rtol=1e-6)


def test_unknown_tensor_stops_the_conversion(tmp_path):
    cfg = model_file.model_S(num_scalar_features=16, num_tensor_features=8, mlp_width=16, readout_width=8)
    w = model_file.init_weights(cfg)
    src = str(tmp_path / "odd.nequip.pth")
    _nequip_style_archive(src, cfg, w, extra_tensor=True)
    with pytest.raises(convert_nequip.ConversionError, match="tensor 'model.mystery_gate' .* has no counterpart"):
        convert_nequip.convert(src)
    r = subprocess.run([sys.executable, "-m", "pair_allegro_amd.tools.convert_nequip", src, "--dry-run"], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 1 and b"has no counterpart" in r.stderr
    # a rule file can adopt it (here: declare it arithmetic-free)
    rules = tmp_path / "rules.json"
    import json
    rules.write_text(json.dumps({"ignore": convert_nequip.DEFAULT_IGNORE + [r".*mystery_gate$"]}))
    r = subprocess.run([sys.executable, "-m", "pair_allegro_amd.tools.convert_nequip", src, "--dry-run", "--map", str(rules)], cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0 and b"converted, " in r.stdout


def test_avg_num_neighbors_is_never_guessed(tmp_path):
    """No buffer in the archive and no flag: an error, not a silent 1.0; a flag that contradicts the buffer: an error."""
    cfg = model_file.model_S(num_scalar_features=16, num_tensor_features=8, mlp_width=16, readout_width=8, avg_num_neighbors=21.5)
    w = model_file.init_weights(cfg)
    for k in range(1, cfg["num_layers"] + 1):
        w[f"l{k}.res"] = np.array([1.0, 0.5]) / np.sqrt(1.25)
    src = str(tmp_path / "noavg.nequip.pth")
    _nequip_style_archive(src, cfg, w, with_avg=False)
    with pytest.raises(convert_nequip.ConversionError, match="avg_num_neighbors"):
        convert_nequip.convert(src)
    cfg2, _, _ = convert_nequip.convert(src, avg_num_neighbors=21.5)
    assert cfg2["avg_num_neighbors"] == 21.5
    src2 = str(tmp_path / "avg.nequip.pth")
    _nequip_style_archive(src2, cfg, w)
    with pytest.raises(convert_nequip.ConversionError, match="contradicts"):
        convert_nequip.convert(src2, avg_num_neighbors=30.0)


def test_missing_tensor_is_named(tmp_path):
    cfg = model_file.model_S(num_scalar_features=16, num_tensor_features=8, mlp_width=16, readout_width=8)
    w = model_file.init_weights(cfg)
    src = str(tmp_path / "m.nequip.pth")
    _nequip_style_archive(src, cfg, w)
    rules = [r for r in convert_nequip.DEFAULT_RULES if "scales" not in r[0]]
    with pytest.raises(convert_nequip.ConversionError, match="scales"):
        convert_nequip.convert(src, rules=rules)
