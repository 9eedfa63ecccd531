"""NVE energy drift of the fused kernel's arithmetics on the same trajectory start (10 648-atom Si box of BASELINE config 2, 300 K, dt = 1 fs):
total energy every `--every` steps for fused_arith = f16x2 (the default), f32 (f32-input MFMA) and, as the yardstick, the float64 generic kernels.
A float32-equivalent arithmetic must show the drift and the fluctuation of the f32 form, not more.

    python pair_allegro_amd/tools/nve_drift.py [--steps 4000] [--every 100] [--out gpurun_out/nve_drift.txt]
"""
import argparse
import os
import sys
import tempfile

os.environ.setdefault("GPU_PINNED_MIN_XFER_SIZE", "4095")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from pair_allegro_amd import capi, md, model_file  # noqa: E402


def run(lib, wl, vel, nsteps, every, opts, cfg_over=None):
    cfg = dict(wl["cfg"], **(cfg_over or {}))
    tmp = tempfile.mkdtemp(prefix="ahip_drift_")
    path = os.path.join(tmp, "m.ahip")
    model_file.save_ahip(path, cfg, model_file.init_weights(wl["cfg"]))      # same weights for every arithmetic
    model = capi.Model(path, 0, lib)
    for k, v in opts.items():
        model.set_option(k, v)
    dev = torch.device("cuda", 0)
    sim = md.Simulation(md.HipBackend(model, wl["masses"]), np.diag(wl["cell"]), cfg["r_max"], 1.0, wl["pos"], wl["mtype"], vel, dev, dt=0.001)
    sim.setup()
    n = len(wl["pos"])
    out = []
    for k in range(nsteps + 1):
        if k % every == 0:
            th = sim.thermo(wl["masses"])
            out.append((k, th["pe"] / n, th["ke"] / n))
        if k < nsteps:
            sim.step()
    used = model.last_path
    torch.cuda.synchronize()
    model.close()
    return used, np.array(out), sim.nrebuild


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=4000)
    ap.add_argument("--every", type=int, default=100)
    ap.add_argument("--f64-steps", type=int, default=1000)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    wl = bench.workload(2, 0)
    n = len(wl["pos"])
    mass = np.asarray(wl["masses"], dtype=np.float64)[wl["mtype"]]
    vel = md.maxwell_boltzmann(n, mass, 300.0, 12345)
    lib = capi.Library()
    lines = [f"# {wl['name']}; NVE, dt = 1 fs, 300 K start; energies in eV per atom; drift = least-squares slope of E_tot(t), fluctuation = rms of the residual"]
    cases = [("f16x2", a.steps, {"fused_arith": "f16x2"}, None), ("f32", a.steps, {"fused_arith": "f32"}, None),
             ("float64 generic", a.f64_steps, {"path": "generic", "precision": "float64"}, None)]
    for name, ns, opts, over in cases:
        try:
            used, e, nreb = run(lib, wl, vel, ns, a.every, opts, over)
        except Exception as ex:                    # an option this build does not know: say so, keep the other cases
            lines.append(f"{name}: not run ({ex})")
            continue
        t = e[:, 0] * 1e-3                          # ps
        etot = e[:, 1] + e[:, 2]
        slope, icpt = np.polyfit(t, etot, 1)
        rms = float(np.sqrt(np.mean((etot - (slope * t + icpt)) ** 2)))
        w = e[:, 0] <= a.f64_steps                  # the window every case covers: the integrator's own error dominates the start of the run
        slope_w = np.polyfit(t[w], etot[w], 1)[0]
        lines.append(f"{name:16s} path {used:12s} steps {ns:5d} rebuilds {nreb:4d}  E_tot(0) {etot[0]:+.9f}  E_tot(end) {etot[-1]:+.9f}  "
                     f"drift {slope:+.3e} eV/atom/ps (first {a.f64_steps} steps: {slope_w:+.3e})  rms fluctuation {rms:.3e} eV/atom  (ke/atom {e[0, 2]:.5f} -> {e[-1, 2]:.5f})")
    txt = "\n".join(lines)
    print(txt)
    if a.out:
        with open(a.out, "w") as f:
            f.write(txt + "\n")


if __name__ == "__main__":
    main()
