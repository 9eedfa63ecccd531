"""Force error of every arithmetic of k_fused against the float64 oracle on the 10 648-atom Si box (BASELINE configs[1] geometry, model S).
   python pair_allegro_amd/tools/arith_check.py [ncell]          (GPU box; the oracle leg is test infrastructure)"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_PINNED_MIN_XFER_SIZE", "4095")
import util  # noqa: E402
from oracle import allegro_torch  # noqa: E402
from pair_allegro_amd import capi, lmp_like, model_file  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 11
lib = capi.Library()
for layers in (2, 3):
    cfg = model_file.model_S(num_layers=layers)
    w = model_file.init_weights(cfg)
    cell, pos, types = lmp_like.diamond_si(n)
    ref = util.oracle_run(dict(cfg, model_dtype="float64"), w, cell, pos, types, ["Si"])
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "m.nequip.pth")
        allegro_torch.export_nequip_pth(path, cfg, w)
        for arith in ("f32", "f16x2", "bf16x3", "tf32eq"):
            r = util.run_pair(lib, path, cell, pos, types, ["Si"], options={"path": "fused", "fused_arith": arith})
            df = np.abs(r["forces"] - ref["forces"])
            print(f"{len(pos)} atoms, {layers} layers, {arith:9s} {r['info']['path']:13s} max|dF| {df.max():.3e} rms {np.sqrt((df ** 2).mean()):.3e} "
                  f"max|dE_i| {np.abs(r['eatom'] - ref['eatom']).max():.3e} |dPE|/N {abs(r['pe'] - ref['pe']) / len(pos):.3e} max|dV| {np.abs(r['virial'] - ref['virial']).max():.3e}", flush=True)
