#!/usr/bin/env python3
"""bench.py -- atom-steps/s of the MI355X-native `pair_style allegro` hot path.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json metric / configs[3]): 1 000 000-atom bulk Si (50^3 diamond cells, a = 5.431 A,
Gaussian jitter 0.05 A seed 0), model S (l_max = 1, 32 tensor features, 64 scalars, 2 layers,
seeded random weights), r_max 5 A, skin 1 A, NVE, dt 1 fs, velocities 300 K.  STRONG scaling: the
same 1 M atoms are brick-decomposed over N GPUs (one process per GPU, ghost exchange via
torch.distributed == RCCL over xGMI).  A "step" is one full MD step: integrate, ghost forward comm
(or re-neighbor when an atom moved > skin/2), force evaluation through the C-ABI, ghost reverse
comm, integrate.  Inputs are resident in HBM before the timed region.

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel (library HIP-event timing on
the launch stream); `cpu_baseline` is the torch oracle (port of the libtorch reference path) timed
on this box's host cores on a bounded sample (BASELINE configs[1], 10 648 atoms).
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SI_MASS = 28.0855

# Algorithmic work per edge of model S (DESIGN.md "Roofline accounting"): MACs of every dense
# contraction, forward; the backward pass needs input gradients only, i.e. the same MAC count again.
def model_macs_per_edge(cfg, two_body_tabulated=False):
    """Dense multiply-accumulates per edge of one forward pass.  With the fused kernel's tabulated two-body embedding
    (default) the two-body MLP is not executed per edge and is left out of the count."""
    T = len(cfg["type_names"]); B = cfg["num_bessels"]; S = cfg["num_scalar_features"]
    U = cfg["num_tensor_features"]; L = cfg["l_max"]; W = cfg["mlp_width"]; R = cfg["readout_width"]
    NL = cfg["num_layers"]; D = (L + 1) ** 2; dep = cfg["mlp_depth"]
    def mlp(din, depth, width, dout):
        dims = [din] + [width] * depth + [dout]
        return sum(a * b for a, b in zip(dims[:-1], dims[1:]))
    fwd = (0 if two_body_tabulated else mlp(2 * T + B, dep, W, S)) + S * U * (L + 1)
    for k in range(1, NL + 1):
        fwd += S * U * (L + 1) + mlp(S + U, dep, W, S)
        if k < NL:
            fwd += U * U * D
    fwd += mlp(S, cfg["readout_depth"], R, 1)
    return fwd


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--ncell", type=int, default=50, help="diamond cells per box edge (50 -> 1M atoms)")
    ap.add_argument("--path", default="auto", choices=["auto", "fused", "generic"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-ncell", type=int, default=11)
    args = ap.parse_args()

    import torch
    from pair_allegro_amd import capi, lmp_like, md, model_file

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        if world == 1 and args.gpus > 1:
            sys.exit(2)
    dist = None
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: allegro-hip has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist_mod.init_process_group(backend="nccl", device_id=device)
        dist = dist_mod

    cfg = model_file.model_S()
    weights = model_file.init_weights(cfg)
    tmpdir = tempfile.mkdtemp(prefix="ahip_bench_")
    model_path = os.path.join(tmpdir, f"modelS_{rank}.ahip")
    model_file.save_ahip(model_path, cfg, weights)

    lib = capi.Library()
    model = capi.Model(model_path, local_rank, lib)
    model.set_option("path", args.path)
    model.set_option("timing", "1")

    cell, pos, _ = lmp_like.diamond_si(args.ncell)
    natoms = len(pos)
    box = np.diag(cell)
    vel = md.maxwell_boltzmann(natoms, np.full(natoms, SI_MASS), 300.0, 12345)
    grid = md.choose_grid(world)
    backend = md.HipBackend(model, [SI_MASS])
    sim = md.Simulation(backend, box, cfg["r_max"], 1.0, pos, np.zeros(natoms, dtype=np.int32), vel, device,
                        grid=grid, rank=rank, dist=dist, dt=0.001)
    sim.setup()
    for _ in range(args.warmup):
        sim.step()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    stage_ms = {}
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sim.step()
        for k, v in model.timings().items():
            stage_ms.setdefault(k, []).append(v)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    th = sim.thermo([SI_MASS])

    if rank == 0:
        ms_per_step = 1e3 * dt / args.steps
        value = natoms * args.steps / dt
        used_path = model.last_path
        # ---- roofline of the dominant kernel -------------------------------------------------
        tb_tab = used_path == "fused_f32" and os.environ.get("AHIP_FUSED_TB", "table") != "mlp"
        # ALGORITHMIC flops = the model's dense contractions (SURVEY 8d / DESIGN 4.2), whatever the kernel does with them;
        # the fused kernel's tabulated two-body embedding executes fewer: reported next to it as executed_flops_per_edge.
        flops_per_edge = 2.0 * model_macs_per_edge(cfg) * 2.0       # 2 flop per MAC x (forward + input-gradient backward)
        executed_flops_per_edge = 2.0 * model_macs_per_edge(cfg, two_body_tabulated=tb_tab) * 2.0
        import ctypes as C
        ne = C.c_longlong(0)
        lib.check(lib.lib.ahip_get_edges(model.h, C.byref(ne), None, None))
        edges_rank0 = ne.value
        stage_avg = {k: float(np.mean(v)) for k, v in stage_ms.items()}
        dom = max((k for k in stage_avg if k.startswith("model")), key=lambda k: stage_avg[k], default=None)
        roof = None
        traffic = None
        tj = os.path.join(ROOT, "profiles", "r01_traffic.json")      # measured in a separate PMC run (cannot be collected live)
        if os.path.exists(tj) and used_path == "fused_f32" and natoms == 1000000 and world == 1:
            traffic = json.load(open(tj))["traffic_bytes_per_launch"]
        if dom is not None:
            ach = flops_per_edge * edges_rank0 / (stage_avg[dom] * 1e-3) / 1e12
            roof = {"bound": "mfma", "achieved": round(ach, 3), "peak": 157.3, "unit": "TFLOP/s",
                    "frac": round(ach / 157.3, 4), "traffic": traffic, "kernel": dom,
                    "avg_ms": round(stage_avg[dom], 3), "edges_per_launch": edges_rank0,
                    "flops_per_edge": flops_per_edge, "executed_flops_per_edge": executed_flops_per_edge,
                    "two_body": "table" if tb_tab else "mlp"}
        # ---- CPU baseline + max|dF| on a bounded sample --------------------------------------
        cpu = None
        max_df = None
        parity = None
        if not args.no_cpu_baseline and world == 1:       # reported baseline: rank 0 at N = 1 only
            cpu, parity = cpu_baseline_and_parity(lib, cfg, weights, model_path, local_rank, args.cpu_sample_ncell, args.path)
            max_df = parity["max_abs_dF"]
        out = {
            "metric": "atom_steps_per_sec", "value": round(value, 1), "unit": "atom-steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{natoms}-atom bulk Si ({args.ncell}^3 diamond cells), model S (l_max=1, U=32, S=64, 2 layers), "
                                   f"r_max 5.0 A + skin 1.0 A, NVE dt=1 fs", "grid": "x".join(map(str, grid)),
                       "kernel_path": used_path, "rebuilds": sim.nrebuild, "stage_ms_rank0": {k: round(v, 3) for k, v in stage_avg.items()},
                       "pe_per_atom": th["pe"] / natoms},
            "max_abs_dF_vs_oracle": max_df,
            "parity_vs_oracle": parity,
            "roofline": roof,
            "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    model.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline_and_parity(lib, cfg, weights, model_path, device_index, ncell, path):
    """Times the oracle (torch CPU, all host cores; a port of the reference's libtorch path) on the
    config-2 box and measures max|dF| of the HIP path against it on the same configuration."""
    import torch
    from oracle import allegro_torch, glue
    from pair_allegro_amd import capi, lmp_like
    cell, pos, types = lmp_like.diamond_si(ncell)
    rs = lmp_like.build_rank_system(cell, pos, types, cfg["r_max"] + 1.0)
    mapper = np.array([0], dtype=np.int32)
    cm = np.array([[cfg["r_max"]]])
    oracle = torch.jit.script(allegro_torch.build(cfg, weights).eval())
    inp = glue.preprocess(rs.x, rs.type, rs.nlocal, rs.ilist, rs.numneigh, rs.firstneigh, mapper, cm)
    tin = {k: torch.from_numpy(v) for k, v in inp.items()}
    oracle(tin)                                                       # warm-up (JIT profiling runs)
    oracle(tin)
    reps, t0 = 0, time.perf_counter()
    while reps < 3 or (time.perf_counter() - t0 < 10.0 and reps < 20):
        out = oracle(tin)
        reps += 1
    t_eval = (time.perf_counter() - t0) / reps
    f_ref = np.zeros_like(rs.x)
    e_ref = np.zeros(len(rs.x))
    pe_ref, vir_ref, _ = glue.compute(oracle, rs.x, rs.type, rs.nlocal, rs.ilist, rs.numneigh, rs.firstneigh, mapper, cm, f_ref, e_ref)
    m = capi.Model(model_path, device_index, lib)
    m.set_option("path", path)
    m.neigh_update_csr(rs.nall, rs.ilist, rs.offsets, rs.flat)
    f = np.zeros_like(rs.x)
    e = np.zeros(len(rs.x))
    pe, vir = m.compute(rs.nlocal, rs.nghost, rs.x, rs.type, mapper, cm, f, e)
    m.close()
    max_df = float(np.abs(f - f_ref).max())
    # the other observables of the reference's own comparison: per-atom energy, PE per atom, virial per atom
    parity = {"max_abs_dF": max_df, "max_abs_dEatom": float(np.abs(e[: rs.nlocal] - e_ref[: rs.nlocal]).max()),
              "abs_dPE_per_atom": float(abs(pe - pe_ref) / rs.nlocal),
              "max_abs_dvirial_per_atom": float(np.abs(vir - vir_ref).max() / rs.nlocal), "atoms": int(rs.nlocal)}
    cpu = {"value": round(rs.nlocal / t_eval, 1), "unit": "atom-steps/s", "cores": torch.get_num_threads(),
           "kind": "port", "sample": f"{rs.nlocal}-atom bulk Si (config 2), {reps} force evaluations of the TorchScript "
                                     f"oracle (float32 model, autograd forces), {t_eval*1e3:.0f} ms each; glue excluded"}
    return cpu, parity


if __name__ == "__main__":
    main()
