"""ORACLE (test infrastructure, never shipped, never on the product path).

Torch restatement of the model that the reference executes through libtorch:
``PairNequIPAllegro<false>::call`` hands a dict {pos f64 [N,3], edge_index i64 [2,E],
atom_types i64 [N]} to a TorchScript module and reads back {atomic_energy f64 [N,1],
forces f64 [N,3], virial f64 [1,3,3]} (/root/reference/pair_nequip_allegro.cpp:409-430,
358-393).  The module here obeys exactly that contract, computes forces/virial with
*autograd* (the hand-written HIP backward is therefore checked against an independent
derivation) and can be exported with ``export_nequip_pth`` as a ``*.nequip.pth`` archive
carrying the five metadata keys of pair_nequip_allegro.cpp:214-220, i.e. a file the
reference pair style itself can load and run.

PARITY UNPINNED w.r.t. nequip/allegro: the Allegro arithmetic is not in /root/reference
(it lives in the un-vendored, unpinned ``nequip``/``allegro`` packages, tests.yml:41-42) and
the reference stores no golden vectors for it.  The arithmetic below follows the build's own
frozen spec (DESIGN.md "Model spec"; hyper-parameter vocabulary from
tests/test_data/test_repro_allegro.yaml:80-103).  What *is* pinned by the reference --
dict keys, dtypes, edge semantics, energy-sum-over-locals, virial order/sign -- is checked
in tests/.
"""
from __future__ import annotations

import math
import os
import sys
from typing import Dict, List, Optional

import numpy as np
import torch

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from pair_allegro_amd import cg, model_file  # noqa: E402  (format + angular conventions only)


class _Lin(torch.nn.Module):
    def __init__(self, w: torch.Tensor, act: bool):
        super().__init__()
        self.register_buffer("w", w)
        self.act = act

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x = x @ self.w
        if self.act:
            x = torch.nn.functional.silu(x)
        return x


class _MLP(torch.nn.Module):
    """x -> silu(x W0) -> ... -> x Wn   (no biases, SiLU on hidden layers only)."""

    def __init__(self, ws: List[torch.Tensor]):
        super().__init__()
        self.lins = torch.nn.ModuleList([_Lin(w, k < len(ws) - 1) for k, w in enumerate(ws)])

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        for lin in self.lins:
            x = lin(x)
        return x


class _Layer(torch.nn.Module):
    def __init__(self, env, tp, lat: _MLP, res, mix: Optional[torch.Tensor], last: bool):
        super().__init__()
        self.register_buffer("env", env)
        self.register_buffer("tp", tp)
        self.lat = lat
        self.register_buffer("res", res)
        self.last = last
        self.register_buffer("mix", mix if mix is not None else torch.zeros(0, dtype=env.dtype))


class AllegroOracle(torch.nn.Module):
    def __init__(self, cfg: dict, weights: Dict[str, np.ndarray]):
        super().__init__()
        dt = {"float32": torch.float32, "float64": torch.float64}[cfg["model_dtype"]]
        self.dt = dt
        T = len(cfg["type_names"])
        self.T = T
        self.B: int = cfg["num_bessels"]
        self.S: int = cfg["num_scalar_features"]
        self.U: int = cfg["num_tensor_features"]
        self.L: int = cfg["l_max"]
        self.D: int = (self.L + 1) ** 2
        self.NL: int = cfg["num_layers"]
        self.p: int = cfg["poly_p"]
        self.r_max: float = float(cfg["r_max"])
        self.inv_sqrt_nn: float = 1.0 / math.sqrt(float(cfg["avg_num_neighbors"]))

        def t(name):
            return torch.tensor(np.asarray(weights[name]), dtype=dt)

        def mlp(prefix, depth):
            return _MLP([t(f"{prefix}.w{k}") for k in range(depth + 1)])

        pc = cfg.get("per_edge_type_cutoff")
        rc = np.full((T, T), self.r_max) if pc is None else np.asarray(pc, dtype=np.float64).reshape(T, T)
        self.register_buffer("rcut", torch.tensor(rc, dtype=torch.float64))
        self.register_buffer("bessel_n", math.pi * torch.arange(1, self.B + 1, dtype=dt))
        self.tb = mlp("tb", cfg["mlp_depth"])
        self.register_buffer("emb", t("emb.w"))
        layers = []
        for k in range(1, self.NL + 1):
            last = k == self.NL
            layers.append(_Layer(t(f"l{k}.env"), t(f"l{k}.tp"), mlp(f"l{k}.lat", cfg["mlp_depth"]),
                                 t(f"l{k}.res"), None if last else t(f"l{k}.mix"), last))
        self.layers = torch.nn.ModuleList(layers)
        self.out = mlp("out", cfg["readout_depth"])
        self.register_buffer("scale", t("scale"))
        self.register_buffer("shift", t("shift"))

        # angular tables
        self.l_of: List[int] = cg.l_of_index(self.L)
        self.register_buffer("l_index", torch.tensor(self.l_of, dtype=torch.long))
        full = cg.tp_paths(self.L, False)
        self.path_l1: List[int] = [p[0] for p in full]
        self.path_l2: List[int] = [p[1] for p in full]
        self.path_l3: List[int] = [p[2] for p in full]
        self.n_scalar_paths: int = len(cg.tp_paths(self.L, True))
        cmax = 2 * self.L + 1
        ctab = np.zeros((len(full), cmax, cmax, cmax))
        for i, (l1, l2, l3) in enumerate(full):
            ctab[i, :2 * l1 + 1, :2 * l2 + 1, :2 * l3 + 1] = cg.path_coeff(l1, l2, l3)
        self.register_buffer("ctab", torch.tensor(ctab, dtype=dt))

    # ---- pieces -------------------------------------------------------------------
    def _sh(self, n: torch.Tensor) -> torch.Tensor:
        x, y, z = n[:, 0], n[:, 1], n[:, 2]
        cols = [torch.ones_like(x)]
        if self.L >= 1:
            s3 = math.sqrt(3.0)
            cols += [s3 * y, s3 * z, s3 * x]
        if self.L >= 2:
            s15 = math.sqrt(15.0)
            s5 = math.sqrt(5.0)
            cols += [s15 * x * y, s15 * y * z, 0.5 * s5 * (2 * z * z - x * x - y * y),
                     s15 * x * z, 0.5 * s15 * (x * x - y * y)]
        if self.L >= 3:                     # the l = 3 block of pair_allegro_amd/cg.py: real_sh (homogeneous cubics, component normalisation)
            s70, s105, s42, s7 = math.sqrt(70.0), math.sqrt(105.0), math.sqrt(42.0), math.sqrt(7.0)
            cols += [0.25 * s70 * y * (3 * x * x - y * y), s105 * x * y * z, 0.25 * s42 * y * (4 * z * z - x * x - y * y),
                     0.5 * s7 * z * (2 * z * z - 3 * x * x - 3 * y * y), 0.25 * s42 * x * (4 * z * z - x * x - y * y),
                     0.5 * s105 * z * (x * x - y * y), 0.25 * s70 * x * (x * x - 3 * y * y)]
        return torch.stack(cols, dim=1)

    def _tp(self, V: torch.Tensor, env_e: torch.Tensor, pw: torch.Tensor, scalar_only: bool) -> torch.Tensor:
        """V, env_e: [E, D, U]; pw [P, U] -> [E, D or 1, U]."""
        E = V.shape[0]
        Dout = 1 if scalar_only else self.D
        out = torch.zeros((E, Dout, self.U), dtype=V.dtype, device=V.device)
        npaths = self.n_scalar_paths if scalar_only else len(self.path_l1)
        for p in range(npaths):
            l1 = self.path_l1[p]
            l2 = self.path_l2[p]
            l3 = self.path_l3[p]
            a = V[:, l1 * l1:(l1 + 1) * (l1 + 1), :]
            b = env_e[:, l2 * l2:(l2 + 1) * (l2 + 1), :]
            c = self.ctab[p, :2 * l1 + 1, :2 * l2 + 1, :2 * l3 + 1]
            contrib = torch.einsum("eau,ebu,abc->ecu", a, b, c) * pw[p].unsqueeze(0).unsqueeze(0)
            out[:, l3 * l3:(l3 + 1) * (l3 + 1), :] = out[:, l3 * l3:(l3 + 1) * (l3 + 1), :] + contrib
        return out

    def edge_energy(self, rvec: torch.Tensor, ti: torch.Tensor, tj: torch.Tensor,
                    centre: torch.Tensor, natoms: int) -> torch.Tensor:
        """Per-edge energies eps_e [E] from edge vectors (model dtype)."""
        dt = rvec.dtype
        d = torch.sqrt((rvec * rvec).sum(dim=1))
        n = rvec / d.unsqueeze(1)
        rc = self.rcut[ti, tj].to(dt)
        x = d / rc
        p = float(self.p)
        fc = 1.0 - 0.5 * (p + 1.0) * (p + 2.0) * torch.pow(x, self.p) \
            + p * (p + 2.0) * torch.pow(x, self.p + 1) - 0.5 * p * (p + 1.0) * torch.pow(x, self.p + 2)
        fc = torch.where(x < 1.0, fc, torch.zeros_like(fc))
        bes = (2.0 / rc).unsqueeze(1) * torch.sin(self.bessel_n.unsqueeze(0) * x.unsqueeze(1)) / d.unsqueeze(1)
        bf = bes * fc.unsqueeze(1)
        oh_i = torch.nn.functional.one_hot(ti, self.T).to(dt)
        oh_j = torch.nn.functional.one_hot(tj, self.T).to(dt)
        xl = self.tb(torch.cat([oh_i, oh_j, bf], dim=1)) * fc.unsqueeze(1)          # x^0 [E,S]
        Y = self._sh(n)                                                              # [E,D]
        w0 = (xl @ self.emb).reshape(-1, self.L + 1, self.U)                          # [E,L+1,U]
        V = w0[:, self.l_index, :] * Y.unsqueeze(2)                                  # [E,D,U]
        for layer in self.layers:
            om = (xl @ layer.env).reshape(-1, self.L + 1, self.U)
            A = om[:, self.l_index, :] * Y.unsqueeze(2)                              # [E,D,U]
            env = torch.zeros((natoms, self.D, self.U), dtype=dt, device=rvec.device)
            env = env.index_add(0, centre, A) * self.inv_sqrt_nn
            Vp = self._tp(V, env[centre], layer.tp, layer.last)
            s = Vp[:, 0, :]
            u = layer.lat(torch.cat([xl, s], dim=1))
            xl = layer.res[0] * xl + layer.res[1] * fc.unsqueeze(1) * u
            if not layer.last:
                V = torch.einsum("edu,duv->edv", Vp, layer.mix[self.l_index])
        return self.out(xl).squeeze(1)

    def forward(self, data: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        pos = data["pos"]
        edge_index = data["edge_index"]
        types = data["atom_types"]
        centre = edge_index[0]
        neigh = edge_index[1]
        natoms = pos.shape[0]
        rvec64 = pos[neigh] - pos[centre]                       # neighbour - centre, f64
        rvec = rvec64.to(self.dt).detach().requires_grad_(True)
        ti = types[centre]
        tj = types[neigh]
        eps = self.edge_energy(rvec, ti, tj, centre, natoms)
        esum = torch.zeros(natoms, dtype=eps.dtype, device=pos.device).index_add(0, centre, eps)
        e_atom = self.scale[types] * (esum * self.inv_sqrt_nn) + self.shift[types]
        grads = torch.autograd.grad([e_atom.sum()], [rvec])
        g = grads[0]
        assert g is not None
        g64 = g.to(torch.float64)
        forces = torch.zeros((natoms, 3), dtype=torch.float64, device=pos.device)
        forces = forces.index_add(0, centre, g64).index_add(0, neigh, -g64)
        vir = -(rvec64.transpose(0, 1) @ g64)
        vir = 0.5 * (vir + vir.transpose(0, 1))
        return {
            "atomic_energy": e_atom.to(torch.float64).unsqueeze(1).detach(),
            "forces": forces.detach(),
            "virial": vir.unsqueeze(0).detach(),
        }


def build(cfg: dict, weights: Optional[Dict[str, np.ndarray]] = None) -> AllegroOracle:
    if weights is None:
        weights = model_file.init_weights(cfg)
    return AllegroOracle(cfg, weights)


def export_nequip_pth(path: str, cfg: dict, weights: Optional[Dict[str, np.ndarray]] = None) -> None:
    """Write ``*.nequip.pth``: TorchScript module + reference metadata + the AHIP blob."""
    assert path.endswith(".nequip.pth")
    if weights is None:
        weights = model_file.init_weights(cfg)
    mod = torch.jit.script(AllegroOracle(cfg, weights).eval())
    extra = dict(model_file.reference_metadata(cfg))
    extra[model_file.BLOB_NAME] = model_file.dumps(cfg, weights)
    torch.jit.save(mod, path, _extra_files=extra)
