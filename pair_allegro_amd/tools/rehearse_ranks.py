"""Rehearsal of the multi-GPU run on ONE GPU at full size: the real decomposition of a BASELINE workload -- e.g. the 1 M-atom Si
box on a 2x2x2 brick grid, 125 000 atoms + ~36 000 ghosts per rank -- with every rank a process on cuda:0, the real kernels, the library's ghost
exchange (pack / unpack kernels, messages staged through gloo by md.HostStagedDist because RCCL refuses two ranks on one device), both schedules
(overlapped three-range, serial), compared atom by atom with the single-rank evaluation of the same box:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29771 \
        pair_allegro_amd/tools/rehearse_ranks.py --config 4 [--ncell N] [--steps 3] [--out gpurun_out/rehearse_config4.json]

Rank 0 prints ONE JSON line: per-rank nlocal / nghost, both schedules' max|dF| by tag and |d pe|/N against the single rank (after set-up and after
`steps` NVE steps: positions too), the exchange's transport and device time.  It asserts: no atom lost, forces by tag within 5e-6 eV/A, pe per atom
within 1e-7 eV, positions after the steps within 1e-7 A.  What it cannot cover is RCCL between devices (the driver's 8-GPU run is the first)."""
import argparse
import json
import os
import sys
import tempfile
import time

os.environ.setdefault("GPU_PINNED_MIN_XFER_SIZE", "4095")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import bench  # noqa: E402
from pair_allegro_amd import capi, md, model_file  # noqa: E402

TOL_F, TOL_PE, TOL_X = 5e-6, 1e-7, 1e-7


def run(lib, path, wl, vel, grid, rank, d, nsteps, overlap, one_device_shared):
    dev = torch.device("cuda", 0)
    model = capi.Model(path, 0, lib)
    if one_device_shared:
        model.set_option("edge_schedule", "dynamic")      # several processes share this GPU: no resident-grid assumption
    model.set_option("timing", "1")
    cfg = wl["cfg"]
    sim = md.Simulation(md.HipBackend(model, wl["masses"]), np.diag(wl["cell"]), cfg["r_max"], 1.0, wl["pos"], wl["mtype"], vel, dev,
                        grid=grid, rank=rank, dist=d, dt=0.001, overlap=overlap)
    sim.setup()
    n = len(wl["pos"])
    out = {"nlocal": int(sim.nlocal), "nghost": int(sim.nall - sim.nlocal), "n_int": int(getattr(sim, "n_int", 0))}
    out["f0"] = sim.gather_forces()
    out["pe0"] = sim.thermo(wl["masses"])["pe"] / n
    sim.time_comm = True
    sim.comm_ms()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(nsteps):
        sim.step()
    torch.cuda.synchronize()
    out["step_ms_wall"] = 1e3 * (time.perf_counter() - t0) / max(nsteps, 1)
    out["comm_ms"] = sim.comm_ms() / max(nsteps, 1)
    x = torch.zeros((n, 3), dtype=torch.float64, device=dev)
    x[sim.tag[: sim.nlocal]] = sim.x[: sim.nlocal]
    if d is not None and sim.nranks > 1:
        d.all_reduce(x)
    box = np.diag(wl["cell"])
    xs = x.cpu().numpy()
    out["x"] = xs - np.floor(xs / box) * box
    out["f1"] = sim.gather_forces()
    th = sim.thermo(wl["masses"])
    out["pe1"] = th["pe"] / n
    out["virial1"] = np.array(th["virial"])
    out["path"] = model.last_path
    out["rebuilds"] = int(sim.nrebuild)
    out["lib_borders"] = bool(getattr(sim, "_lib_plan", False))
    # cost of one re-neighboring per rank (migration + borders + plan + neighbor list), twice: inside the library (ahip_comm_migrate / ahip_comm_borders, round 6)
    # and by the torch swap chain (AHIP_LIB_BORDERS=0); several ranks share this GPU, so the numbers are upper bounds of what a rank of its own sees
    out["rebuild_ms"] = {}
    if sim.nranks > 1:
        for label, flag in (("library", "1"), ("torch_swap_chain", "0")):
            os.environ["AHIP_LIB_BORDERS"] = flag
            sim.rebuild()                              # (first one after a switch: allocations)
            if d is not None:
                d.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            sim.rebuild()
            torch.cuda.synchronize()
            out["rebuild_ms"][label] = 1e3 * (time.perf_counter() - t0)
        os.environ["AHIP_LIB_BORDERS"] = "1"
    ci = getattr(sim, "comm_info", None) or {"transport": "none"}
    out["transport"] = ci.get("transport", "none")
    torch.cuda.synchronize()
    model.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=4)
    ap.add_argument("--ncell", type=int, default=0)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    staged = md.HostStagedDist(dist)
    grid = md.choose_grid(world)
    wl = bench.workload(args.config, args.ncell)
    cfg = wl["cfg"]
    n = len(wl["pos"])
    lib = capi.Library()
    tmp = tempfile.mkdtemp(prefix="ahip_rehearse_")
    path = os.path.join(tmp, f"m{rank}.ahip")
    model_file.save_ahip(path, cfg, model_file.init_weights(cfg))
    mass_by_mtype = np.asarray(wl["masses"], dtype=np.float64)
    vel = md.maxwell_boltzmann(n, mass_by_mtype[wl["mtype"]], 300.0, 12345)
    res = {}
    for name, ov in (("overlapped", True), ("serial", False)):
        r = run(lib, path, wl, vel, grid, rank, staged, args.steps, ov, True)
        counts = torch.tensor([[r["nlocal"], r["nghost"], r["n_int"]]], dtype=torch.int64)
        allc = [torch.zeros_like(counts) for _ in range(world)]
        dist.all_gather(allc, counts)
        r["counts"] = [c.view(-1).tolist() for c in allc]
        tt = torch.tensor([r["step_ms_wall"], r["comm_ms"], r["rebuild_ms"].get("library", 0.0), r["rebuild_ms"].get("torch_swap_chain", 0.0)], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        r["step_ms_wall"], r["comm_ms"] = float(tt[0]), float(tt[1])
        r["rebuild_ms"] = {"library": float(tt[2]), "torch_swap_chain": float(tt[3])}
        res[name] = r
    dist.barrier()
    ok = True
    if rank == 0:
        ref = run(lib, path, wl, vel, (1, 1, 1), 0, None, args.steps, False, False)
        line = {"workload": wl["name"], "ranks": world, "grid": "x".join(map(str, grid)), "steps": args.steps, "kernel_path": ref["path"],
                "single_rank": {"nlocal": ref["nlocal"], "nghost": ref["nghost"], "pe_per_atom": ref["pe0"], "step_ms_wall": round(ref["step_ms_wall"], 3)},
                "tolerances": {"max_abs_dF": TOL_F, "pe_per_atom": TOL_PE, "positions": TOL_X}}
        for name, r in res.items():
            nl = [c[0] for c in r["counts"]]
            e = {"nlocal": nl, "nghost": [c[1] for c in r["counts"]], "n_interior": [c[2] for c in r["counts"]],
                 "max_abs_dF_setup": float(np.abs(r["f0"] - ref["f0"]).max()), "max_abs_dF_after_steps": float(np.abs(r["f1"] - ref["f1"]).max()),
                 "d_pe_per_atom_setup": float(abs(r["pe0"] - ref["pe0"])), "d_pe_per_atom_after_steps": float(abs(r["pe1"] - ref["pe1"])),
                 "max_abs_dx_after_steps": float(np.abs(r["x"] - ref["x"]).max()), "max_abs_dvirial": float(np.abs(r["virial1"] - ref["virial1"]).max()),
                 "comm_transport": r["transport"], "comm_ms_per_step_max_over_ranks": round(r["comm_ms"], 4),
                 "step_ms_wall_max_over_ranks_shared_gpu": round(r["step_ms_wall"], 3), "rebuilds": r["rebuilds"], "kernel_path": r["path"],
                 "re_neighboring_in_the_library": r["lib_borders"], "rebuild_ms_max_over_ranks_shared_gpu": {k: round(v, 3) for k, v in r["rebuild_ms"].items()}}
            good = (sum(nl) == n and e["max_abs_dF_setup"] < TOL_F and e["max_abs_dF_after_steps"] < TOL_F and e["d_pe_per_atom_setup"] < TOL_PE
                    and e["d_pe_per_atom_after_steps"] < TOL_PE and e["max_abs_dx_after_steps"] < TOL_X)
            e["ok"] = bool(good)
            ok = ok and good
            line[name] = e
        line["ok"] = bool(ok)
        s = json.dumps(line)
        print(s, flush=True)
        if args.out:
            with open(args.out, "w") as fo:
                fo.write(s + "\n")
    flag = torch.tensor([1 if ok else 0])
    dist.broadcast(flag, 0)
    dist.barrier()
    dist.destroy_process_group()
    if int(flag.item()) != 1:
        raise SystemExit("rehearsal: multi-rank results differ from the single rank beyond the tolerances (see the JSON line)")


if __name__ == "__main__":
    main()
