"""Converter: genuine `nequip-compile` TorchScript archive (`*.nequip.pth`) -> the same archive + the `allegro_hip.bin`
weight section that liballegro_hip.so reads (reference load path: /root/reference/pair_nequip_allegro.cpp:214-232).

    python -m pair_allegro_amd.tools.convert_nequip model.nequip.pth out.nequip.pth [--map my_rules.json] [--dry-run]

What it does
  1. opens the archive with `torch.jit.load` on the CPU -- a TorchScript archive is self-contained (its `code/` member holds
     the graph), so neither `nequip` nor `allegro` has to be importable -- and reads the five metadata members the reference
     reads (`r_max`, `type_names`, `num_types`, `per_edge_type_cutoff`, `allow_tf32`; layout probed in SURVEY.md App. B);
  2. takes `state_dict()` and maps every tensor name onto the allegro-hip tensor names of `model_file.tensor_shapes` with the
     RULES table below (regular expression -> target name, orientation).  Every tensor must either map or match IGNORE;
     anything else stops the conversion with "tensor X has no counterpart", and every allegro-hip tensor must receive exactly
     one source (shape-checked), else "allegro-hip tensor Y has no source";
  3. infers the hyper-parameters (l_max, widths, depths, layer count) from the mapped shapes, writes the AHIP blob and stores
     it as an extra STORED member next to the originals.

PARITY UNPINNED: no real `nequip-compile` output exists in this environment and the `nequip`/`allegro`
sources are not available, so the RULES table encodes the parameter naming of the public allegro >= 0.7 code base *as
recalled*, and the numerical conventions listed in docs/MODEL_SPEC.md ("Conventions a converted model must satisfy") are
assumptions until a genuine file can be run through the oracle.  The table is data: `--map rules.json` replaces it, so a naming
difference is a configuration change, not a code change.  The test (tests/test_convert_nequip.py) drives the whole path with a
synthetic archive written by `torch.jit.save` of a module carrying these names.
"""
from __future__ import annotations

import argparse
import json
import re
import sys
import zipfile
from typing import Dict, List, Optional, Tuple

import numpy as np

from .. import cg, model_file

METADATA_KEYS = ["r_max", "per_edge_type_cutoff", "type_names", "num_types", "allow_tf32"]

# (regular expression on the state-dict name, allegro-hip tensor name template, orientation)
#   orientation "io"  : stored [in, out]  -> used as is          (x @ W)
#               "oi"  : stored [out, in]  -> transposed
#               "vec" : 1-D, used as is
# {k} in a template is replaced by 1 + the first captured group (layers are 1-based in the AHIP names), {j} by the second.
DEFAULT_RULES: List[Tuple[str, str, str]] = [
    (r".*scalar_embed.*(?:mlp|embed_mlp)\._weight_(\d+)$", "tb.w{j0}", "io"),
    (r".*tensor_embed.*(?:env_embed|_env_weighter|weights?_linear).*weight$", "emb.w", "io"),
    (r".*allegro\.env_embed_mlps\.(\d+)\._weight_0$", "l{k}.env", "io"),
    (r".*allegro\.tps\.(\d+)\.(?:path_weights?|weights?)$", "l{k}.tp", "io"),
    (r".*allegro\.latents\.(\d+)\._weight_(\d+)$", "l{k}.lat.w{j}", "io"),
    (r".*allegro\.linears\.(\d+)\.(?:weight|weights)$", "l{k}.mix", "mix"),
    (r".*allegro\._latent_resnet_update_params$", "__resnet__", "vec"),
    (r".*avg_num_neighbors$", "__avg_num_neighbors__", "vec"),        # the archive's own normalisation constant (a buffer)
    (r".*(?:edge_readout|readout|edge_eng).*\._weight_(\d+)$", "out.w{j0}", "io"),
    (r".*per_type_energy_scale_shift\.scales$", "scale", "vec"),
    (r".*per_type_energy_scale_shift\.shifts$", "shift", "vec"),
]
# tensors that carry no learned arithmetic of this model spec (constants the kernels regenerate, bookkeeping buffers)
DEFAULT_IGNORE = [r".*bessel\.bessel_weights$", r".*_zero$", r".*\.cg$", r".*w3j.*", r".*_dummy.*", r".*\.num_batches_tracked$"]


class ConversionError(RuntimeError):
    pass


def read_archive(path: str):
    import torch
    extra = {k: "" for k in METADATA_KEYS}
    mod = torch.jit.load(path, map_location="cpu", _extra_files=extra)
    meta = {k: (v.decode() if isinstance(v, bytes) else v) for k, v in extra.items()}
    sd = {k: v.detach().cpu().to(torch.float64).numpy() for k, v in mod.state_dict().items() if v.is_floating_point()}
    return meta, sd


def map_state_dict(sd: Dict[str, np.ndarray], rules=None, ignore=None) -> Dict[str, np.ndarray]:
    rules = DEFAULT_RULES if rules is None else rules
    ignore = DEFAULT_IGNORE if ignore is None else ignore
    out: Dict[str, np.ndarray] = {}
    src: Dict[str, str] = {}
    for name, arr in sd.items():
        hit = None
        for pat, tmpl, orient in rules:
            m = re.match(pat, name)
            if m:
                hit = (m, tmpl, orient)
                break
        if hit is None:
            if any(re.match(p, name) for p in ignore):
                continue
            raise ConversionError(f"tensor '{name}' {tuple(arr.shape)} has no counterpart in the allegro-hip model spec "
                                  f"(add a rule with --map, or an ignore pattern if it carries no learned arithmetic)")
        m, tmpl, orient = hit
        gr = m.groups()
        target = tmpl
        if "{k}" in target:
            target = target.replace("{k}", str(int(gr[0]) + 1))
        if "{j}" in target:
            target = target.replace("{j}", str(int(gr[1])))
        if "{j0}" in target:
            target = target.replace("{j0}", str(int(gr[0])))
        a = np.asarray(arr, dtype=np.float64)
        if orient == "oi":
            a = a.T
        elif orient == "mix":                                   # [l][u_in][u_out] expected; accept [l][u_out][u_in] via --map "mix_oi"
            if a.ndim != 3:
                raise ConversionError(f"tensor '{name}': channel-mixing weights must be 3-D [l_max+1][U][U], got {a.shape}")
        elif orient == "mix_oi":
            a = np.transpose(a, (0, 2, 1))
        if target in out:
            raise ConversionError(f"allegro-hip tensor '{target}' has two sources: '{src[target]}' and '{name}'")
        out[target] = np.ascontiguousarray(a)
        src[target] = name
    return out


def infer_cfg(meta: Dict[str, str], w: Dict[str, np.ndarray], avg_num_neighbors: Optional[float]) -> dict:
    type_names = meta["type_names"].split()
    T = len(type_names)
    if meta.get("num_types") and int(float(meta["num_types"])) != T:
        raise ConversionError("metadata: num_types does not match type_names")
    def need(n):
        if n not in w:
            raise ConversionError(f"allegro-hip tensor '{n}' has no source in the archive")
        return w[n]

    tb0 = need("tb.w0")
    depth = 0
    while f"tb.w{depth + 1}" in w:
        depth += 1
    S = w[f"tb.w{depth}"].shape[1]
    B = tb0.shape[0] - 2 * T
    if B <= 0:
        raise ConversionError(f"two-body MLP input width {tb0.shape[0]} is not 2*num_types + num_bessels")
    NL = 0
    while f"l{NL + 1}.env" in w:
        NL += 1
    if NL == 0:
        raise ConversionError("no Allegro layer found (allegro-hip tensor 'l1.env' has no source in the archive)")
    env = w["l1.env"]
    lat0 = need("l1.lat.w0")
    U = lat0.shape[0] - S
    if U <= 0 or env.shape[1] % U:
        raise ConversionError("cannot infer num_tensor_features from l1.lat.w0 / l1.env shapes")
    L = env.shape[1] // U - 1
    rd = 0
    while f"out.w{rd + 1}" in w:
        rd += 1
    pc = meta.get("per_edge_type_cutoff", "").split()
    cfg = dict(model_file.DEFAULT_CFG, type_names=type_names, r_max=float(meta["r_max"]),
               per_edge_type_cutoff=(np.array([float(v) for v in pc]).reshape(T, T).tolist() if pc else None),
               num_bessels=int(B), l_max=int(L), num_layers=NL, num_scalar_features=int(S), num_tensor_features=int(U),
               mlp_depth=depth, mlp_width=int(tb0.shape[1]), readout_depth=rd, readout_width=int(w["out.w0"].shape[1]) if rd else 1,
               avg_num_neighbors=float(avg_num_neighbors), model_dtype="float32")
    return cfg


def convert(path: str, rules=None, ignore=None, avg_num_neighbors: Optional[float] = None):
    meta, sd = read_archive(path)
    w = map_state_dict(sd, rules, ignore)
    if "__resnet__" in w:
        # one learned parameter p per layer -> (alpha, beta) of x <- alpha x + beta u, the upstream allegro form (as recalled):
        # s = sigmoid(p), coefficient_old = rsqrt(s^2 + 1), coefficient_new = s * coefficient_old
        sg = 1.0 / (1.0 + np.exp(-np.atleast_1d(w.pop("__resnet__"))))
        for k in range(len(sg)):
            a_old = 1.0 / np.sqrt(sg[k] * sg[k] + 1.0)
            w[f"l{k + 1}.res"] = np.array([a_old, sg[k] * a_old])
    # environment normalisation 1/sqrt(avg_num_neighbors): the archive's own buffer, else the command line; never a silent default
    # (a wrong constant gives wrong energies and forces with no error)
    file_avg = w.pop("__avg_num_neighbors__", None)
    if avg_num_neighbors is None:
        if file_avg is None:
            raise ConversionError("the archive carries no avg_num_neighbors buffer: pass --avg-num-neighbors (the value the model was trained with)")
        avg_num_neighbors = float(np.ravel(file_avg)[0])
    elif file_avg is not None and abs(float(np.ravel(file_avg)[0]) - avg_num_neighbors) > 1e-6 * abs(avg_num_neighbors):
        raise ConversionError(f"--avg-num-neighbors {avg_num_neighbors} contradicts the archive's buffer {float(np.ravel(file_avg)[0])}")
    if not avg_num_neighbors > 0:
        raise ConversionError("avg_num_neighbors must be positive")
    cfg = infer_cfg(meta, w, avg_num_neighbors)
    want = dict(model_file.tensor_shapes(cfg))
    for name, shape in want.items():
        if name not in w:
            raise ConversionError(f"allegro-hip tensor '{name}' {shape} has no source in the archive")
        if tuple(w[name].shape) != tuple(shape):
            raise ConversionError(f"allegro-hip tensor '{name}': expected shape {shape}, the archive gives {tuple(w[name].shape)}")
    extra_t = sorted(set(w) - set(want))
    if extra_t:
        raise ConversionError(f"mapped tensors without a place in the model spec: {extra_t}")
    return cfg, {k: w[k] for k in want}, len(want)


def write_with_blob(src: str, dst: str, cfg: dict, w: Dict[str, np.ndarray]) -> None:
    """Copy the archive member by member (STORED, as TorchScript writes them) and add <root>/extra/allegro_hip.bin."""
    with zipfile.ZipFile(src) as zin, zipfile.ZipFile(dst, "w", compression=zipfile.ZIP_STORED) as zout:
        root = zin.namelist()[0].split("/")[0]
        for item in zin.infolist():
            if item.filename.endswith("extra/" + model_file.BLOB_NAME):
                continue
            zout.writestr(item.filename, zin.read(item.filename), compress_type=zipfile.ZIP_STORED)
        zout.writestr(f"{root}/extra/{model_file.BLOB_NAME}", model_file.dumps(cfg, w), compress_type=zipfile.ZIP_STORED)


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("src")
    ap.add_argument("dst", nargs="?")
    ap.add_argument("--map", help="JSON file {\"rules\": [[regex, target, orientation], ...], \"ignore\": [regex, ...]}")
    ap.add_argument("--avg-num-neighbors", type=float, default=None, help="normalisation constant if the archive does not carry it")
    ap.add_argument("--dry-run", action="store_true")
    a = ap.parse_args(argv)
    rules = ignore = None
    if a.map:
        j = json.load(open(a.map))
        rules = [tuple(r) for r in j.get("rules", DEFAULT_RULES)]
        ignore = j.get("ignore", DEFAULT_IGNORE)
    try:
        cfg, w, n = convert(a.src, rules, ignore, a.avg_num_neighbors)
    except ConversionError as e:
        print(f"convert_nequip: {e}", file=sys.stderr)
        return 1
    print("convert_nequip: PARITY UNPINNED -- the name rules and numerical conventions (docs/MODEL_SPEC.md) have never met a genuine "
          "nequip-compile archive; check the result with `python tests/pin_real_model.py <model> <structure.xyz>` before trusting it", file=sys.stderr)
    print(f"converted, {n} tensors mapped: l_max={cfg['l_max']} U={cfg['num_tensor_features']} S={cfg['num_scalar_features']} "
          f"layers={cfg['num_layers']} types={' '.join(cfg['type_names'])} r_max={cfg['r_max']}")
    if not a.dry_run:
        if not a.dst:
            print("convert_nequip: no output path given", file=sys.stderr)
            return 2
        write_with_blob(a.src, a.dst, cfg, w)
    return 0


if __name__ == "__main__":
    sys.exit(main())
