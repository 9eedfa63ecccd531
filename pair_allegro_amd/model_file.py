"""allegro-hip model file: hyper-parameters, weight initialiser, reader/writer.

A model is a plain dict ``cfg`` of hyper-parameters plus a dict ``weights`` of float64
numpy arrays (names/shapes in :func:`tensor_shapes`, semantics in DESIGN.md "Model spec").

On disk it is the *AHIP blob*:

    line 1 : ``AHIPMDL1 <header_bytes:08d>\n``   (header_bytes = offset of the binary section)
    lines  : ``key value...\n`` hyper-parameters, then ``tensor <name> <ndim> <dims..> <offset>\n``
             (offset in bytes from the start of the binary section), then ``end\n``,
             zero padding up to header_bytes
    binary : little-endian float64 tensors, each 64-byte aligned

The blob is stored either as a bare file (``*.ahip``) or -- the drop-in form -- as the
extra file ``allegro_hip.bin`` inside a TorchScript archive ``*.nequip.pth`` next to the five
metadata keys the reference pair style reads (``r_max``, ``per_edge_type_cutoff``,
``type_names``, ``num_types``, ``allow_tf32``;
/root/reference/pair_nequip_allegro.cpp:214-220), so the *same file* can be handed to the
reference ``pair_style allegro`` (libtorch) and to this one (HIP).  TorchScript archives are
ZIP files with STORED (uncompressed) members, which is what the C++ loader
(csrc/model_io.cpp) relies on.
"""
from __future__ import annotations

import io
import zipfile
from typing import Dict, List, Tuple

import numpy as np

from . import cg

MAGIC = b"AHIPMDL1"
BLOB_NAME = "allegro_hip.bin"

DEFAULT_CFG = dict(
    model_dtype="float32",
    type_names=["Si"],
    r_max=5.0,
    per_edge_type_cutoff=None,      # None or [T][T] nested list, model-type index
    num_bessels=8,
    poly_p=6,
    l_max=1,
    num_layers=2,
    num_scalar_features=64,
    num_tensor_features=32,
    mlp_depth=2,                    # hidden layers of two-body and latent MLPs
    mlp_width=64,
    readout_depth=1,
    readout_width=32,
    avg_num_neighbors=28.0,
    seed=1,
)


def model_S(**over) -> dict:
    """BASELINE config-2/4 model: l_max=1, 32 tensor features, 64 scalars, 2 layers."""
    c = dict(DEFAULT_CFG)
    c.update(over)
    return c


def model_L(**over) -> dict:
    """BASELINE config-5 model: l_max=2, 64 tensor features, 3 layers (water O/H)."""
    c = dict(DEFAULT_CFG, type_names=["O", "H"], l_max=2, num_layers=3,
             num_tensor_features=64, avg_num_neighbors=52.5)
    c.update(over)
    return c


def n_paths(cfg: dict, layer: int) -> int:
    last = layer == cfg["num_layers"]
    return len(cg.tp_paths(cfg["l_max"], scalar_only=last))


def tensor_shapes(cfg: dict) -> List[Tuple[str, Tuple[int, ...]]]:
    T = len(cfg["type_names"])
    B, S, U = cfg["num_bessels"], cfg["num_scalar_features"], cfg["num_tensor_features"]
    L, W, NL = cfg["l_max"], cfg["mlp_width"], cfg["num_layers"]
    R = cfg["readout_width"]
    out: List[Tuple[str, Tuple[int, ...]]] = []

    def mlp(prefix, din, depth, width, dout):
        dims = [din] + [width] * depth + [dout]
        for k in range(len(dims) - 1):
            out.append((f"{prefix}.w{k}", (dims[k], dims[k + 1])))

    mlp("tb", 2 * T + B, cfg["mlp_depth"], W, S)
    out.append(("emb.w", (S, U * (L + 1))))
    for k in range(1, NL + 1):
        out.append((f"l{k}.env", (S, U * (L + 1))))
        out.append((f"l{k}.tp", (n_paths(cfg, k), U)))
        mlp(f"l{k}.lat", S + U, cfg["mlp_depth"], W, S)
        out.append((f"l{k}.res", (2,)))
        if k < NL:
            out.append((f"l{k}.mix", (L + 1, U, U)))
    mlp("out", S, cfg["readout_depth"], R, 1)
    out.append(("scale", (T,)))
    out.append(("shift", (T,)))
    return out


SILU_GAIN = 1.6790                   # 1/sqrt(E[silu(z)^2]), z ~ N(0,1)


def init_weights(cfg: dict) -> Dict[str, np.ndarray]:
    """Seeded, variance-preserving random initialisation (float64)."""
    rng = np.random.RandomState(cfg["seed"])
    w: Dict[str, np.ndarray] = {}
    for name, shape in tensor_shapes(cfg):
        leaf = name.split(".")[-1]
        if name in ("scale",):
            w[name] = 8.0 + 1.0 * np.arange(shape[0], dtype=np.float64)   # force constants ~10 eV/A^2
        elif name in ("shift",):
            w[name] = -5.0 - 0.3 * np.arange(shape[0], dtype=np.float64)
        elif leaf == "res":
            c = 0.8
            a = 1.0 / np.sqrt(1.0 + c * c)
            w[name] = np.array([a, c * a])
        elif leaf == "tp":
            # ~1/sqrt(#paths feeding an output irrep)
            w[name] = rng.normal(size=shape) / np.sqrt(max(1, shape[0] / (cfg["l_max"] + 1)))
        elif leaf == "mix":
            w[name] = rng.normal(size=shape) / np.sqrt(shape[1])
        else:
            fan_in = shape[0]
            gain = 1.0 if leaf == "w0" or leaf in ("w", "env") else SILU_GAIN
            w[name] = rng.normal(size=shape) * gain / np.sqrt(fan_in)
    return w


# ----------------------------------------------------------------------------- blob I/O

def _fmt(v) -> str:
    if isinstance(v, float):
        return repr(float(v))
    return str(v)


def dumps(cfg: dict, weights: Dict[str, np.ndarray]) -> bytes:
    T = len(cfg["type_names"])
    lines = ["version 1"]
    for k in ("model_dtype", "r_max", "num_bessels", "poly_p", "l_max", "num_layers",
              "num_scalar_features", "num_tensor_features", "mlp_depth", "mlp_width",
              "readout_depth", "readout_width", "avg_num_neighbors", "seed"):
        lines.append(f"{k} {_fmt(cfg[k])}")
    lines.append(f"num_types {T}")
    if int(cfg.get("allow_tf32", 0)):
        lines.append("allow_tf32 1")
    lines.append("type_names " + " ".join(cfg["type_names"]))
    pc = cfg.get("per_edge_type_cutoff")
    if pc is not None:
        flat = np.asarray(pc, dtype=np.float64).reshape(T * T)
        lines.append("per_edge_type_cutoff " + " ".join(repr(float(x)) for x in flat))
    off = 0
    chunks = []
    for name, shape in tensor_shapes(cfg):
        a = np.ascontiguousarray(weights[name], dtype="<f8")
        assert tuple(a.shape) == tuple(shape), (name, a.shape, shape)
        lines.append(f"tensor {name} {len(shape)} " + " ".join(str(s) for s in shape) + f" {off}")
        b = a.tobytes()
        pad = (-len(b)) % 64
        chunks.append(b + b"\0" * pad)
        off += len(b) + pad
    lines.append("end")
    body = ("\n".join(lines) + "\n").encode()
    first_len = len(MAGIC) + 1 + 8 + 1
    header_bytes = ((first_len + len(body) + 63) // 64) * 64
    head = MAGIC + b" " + f"{header_bytes:08d}".encode() + b"\n" + body
    head += b"\0" * (header_bytes - len(head))
    return head + b"".join(chunks)


def loads(blob: bytes) -> Tuple[dict, Dict[str, np.ndarray]]:
    if blob[:8] != MAGIC:
        raise ValueError("not an AHIP model blob")
    header_bytes = int(blob[9:17])
    text = blob[18:header_bytes].split(b"\0", 1)[0].decode()
    cfg: dict = {"per_edge_type_cutoff": None}
    tensors = []
    for line in text.splitlines():
        tok = line.split()
        if not tok:
            continue
        if tok[0] == "end":
            break
        if tok[0] == "tensor":
            nd = int(tok[2])
            shape = tuple(int(t) for t in tok[3:3 + nd])
            tensors.append((tok[1], shape, int(tok[3 + nd])))
        elif tok[0] == "type_names":
            cfg["type_names"] = tok[1:]
        elif tok[0] == "per_edge_type_cutoff":
            cfg["per_edge_type_cutoff"] = [float(t) for t in tok[1:]]
        elif tok[0] == "model_dtype":
            cfg["model_dtype"] = tok[1]
        elif tok[0] in ("r_max", "avg_num_neighbors"):
            cfg[tok[0]] = float(tok[1])
        else:
            cfg[tok[0]] = int(tok[1])
    T = len(cfg["type_names"])
    if cfg["per_edge_type_cutoff"] is not None:
        cfg["per_edge_type_cutoff"] = np.asarray(cfg["per_edge_type_cutoff"]).reshape(T, T).tolist()
    w = {}
    for name, shape, off in tensors:
        n = int(np.prod(shape))
        w[name] = np.frombuffer(blob, dtype="<f8", count=n, offset=header_bytes + off).reshape(shape).copy()
    return cfg, w


def reference_metadata(cfg: dict) -> Dict[str, str]:
    """The five keys the reference reads (pair_nequip_allegro.cpp:214-220, 267-328)."""
    T = len(cfg["type_names"])
    pc = cfg.get("per_edge_type_cutoff")
    return {
        "r_max": repr(float(cfg["r_max"])),
        "per_edge_type_cutoff": "" if pc is None else " ".join(
            repr(float(x)) for x in np.asarray(pc, dtype=np.float64).reshape(T * T)),
        "type_names": " ".join(cfg["type_names"]),
        "num_types": str(T),
        "allow_tf32": "1" if int(cfg.get("allow_tf32", 0)) else "0",
    }


def save_ahip(path: str, cfg: dict, weights: Dict[str, np.ndarray]) -> None:
    with open(path, "wb") as f:
        f.write(dumps(cfg, weights))


def load(path: str) -> Tuple[dict, Dict[str, np.ndarray]]:
    """Read a bare ``.ahip`` blob or the blob embedded in a ``.nequip.pth`` archive."""
    with open(path, "rb") as f:
        head = f.read(8)
    if head == MAGIC:
        with open(path, "rb") as f:
            return loads(f.read())
    with zipfile.ZipFile(path) as z:
        for n in z.namelist():
            if n.endswith("extra/" + BLOB_NAME):
                return loads(z.read(n))
    raise ValueError(f"{path}: no {BLOB_NAME} section (not an allegro-hip model file)")
