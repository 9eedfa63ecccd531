#!/usr/bin/env python3
"""bench.py -- atom-steps/s of the MI355X-native `pair_style allegro` hot path.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

--config {2,3,4,5} selects the BASELINE.json configuration (default 4 = the one the metric is quoted on);
`python bench.py --gpus N` without a launcher starts its own N rank processes.

Workload (BASELINE.json metric / configs[3]): 1 000 000-atom bulk Si (50^3 diamond cells, a = 5.431 A,
Gaussian jitter 0.05 A seed 0), model S (l_max = 1, 32 tensor features, 64 scalars, 2 layers,
seeded random weights), r_max 5 A, skin 1 A, NVE, dt 1 fs, velocities 300 K.  STRONG scaling: the
same 1 M atoms are brick-decomposed over N GPUs (one process per GPU; the per-step ghost exchange is the
library's own -- HIP pack / unpack kernels + RCCL send / receive groups over xGMI, csrc/comm.hip -- and
torch.distributed (backend nccl = RCCL) carries the setup and the re-neighboring steps).  A "step" is one full MD step: integrate, ghost forward comm
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
# harness-side HIP runtime default (pageable copies issued by torch take the runtime's staged path; the library stages its own): capi.harness_pinned_copy_default
os.environ.setdefault("GPU_PINNED_MIN_XFER_SIZE", "4095")

SI_MASS = 28.0855

# Algorithmic work per edge of model S (DESIGN.md "Roofline accounting"): MACs of every dense
# contraction, forward; the backward pass needs input gradients only, i.e. the same MAC count again.
def model_macs_per_edge(cfg, two_body_tabulated=False, readout_folded=False):
    """Dense multiply-accumulates per edge of one forward pass.  With the fused kernel's tabulated two-body embedding
    (default) the two-body MLP is not executed per edge and is left out of the count; readout_folded: k_fused
    multiplies the last layer's third latent linear into the read-out's first one (W -> S and S -> R become W -> R and S -> R)."""
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
    if readout_folded:
        fwd += (W * R + S * R) - (W * S + S * R)
    return fwd


def model_algorithmic_bytes_per_edge(cfg, edges_per_centre):
    """ALGORITHMIC HBM bytes per edge of one force evaluation, layer-at-a-time formulation, float32 state -- SURVEY.md 8(d), "ALGORITHMIC bytes" row,
    summed per kernel exactly as that row lists them (this is the figure `roofline.achieved` is priced on, whatever a fused kernel really moves):
      gather / embed   read j 4 + x_j 24 + type_j 4; write r_e 12 + Bessel x cutoff 4 B + Y 4 D                 (92 B for l_max 1, 112 B for l_max 2)
      tensor product   per layer, forward: read V_e 4 U D + environment weights 4 U (l_max + 1) + the centre's env 4 U D (once per ATOM, i.e. / edges per
                       centre); write V'_e 4 U D   (1.3 KB for model S, 5.5 KB for model L);   backward = 2 x forward
      latent           per layer: x_e read, x_e' written, its gradient read once: 3 x 4 S   (the rest of SURVEY's "4.6 KB / edge / layer, fwd + bwd" for model S)
      force scatter    read g_e 12, one 24-byte atomic per edge, 24 bytes written per atom
    Model S on Si (28 edges per centre, 2 layers): 9.45 KB per edge = 264.7 KB per atom-step (SURVEY 8d rounds the same sum to "about 260 KB / atom-step")."""
    S = cfg["num_scalar_features"]; U = cfg["num_tensor_features"]; L = cfg["l_max"]; NL = cfg["num_layers"]; B = cfg["num_bessels"]
    D = (L + 1) ** 2
    gather = 4 + 24 + 4 + 12 + 4 * B + 4 * D
    tp_fwd = 4 * U * D + 4 * U * (L + 1) + 4 * U * D / max(edges_per_centre, 1.0) + 4 * U * D
    layer = 3.0 * tp_fwd + 3 * 4 * S
    scatter = 12 + 24 + 24.0 / max(edges_per_centre, 1.0)
    return gather + NL * layer + scatter


# ---- BASELINE.json configs.  configs[0] (64-atom Si on CPU libtorch) is the reference's own plumbing case
# and appears only in the parity tests; --config 4 (1 M-atom Si, model S) is the configuration the metric is quoted on.
def workload(config: int, ncell: int = 0):
    """-> dict(name, cell, pos, mtype (model type per atom), cfg, masses (per model type), lammps_names)."""
    from pair_allegro_amd import lmp_like, model_file
    if config in (2, 4):
        n = ncell or (11 if config == 2 else 50)
        cell, pos, types = lmp_like.diamond_si(n)
        cfg = model_file.model_S()
        return dict(name=f"{len(pos)}-atom bulk Si ({n}^3 diamond cells), model S (l_max=1, U=32, S=64, 2 layers)",
                    cell=cell, pos=pos, mtype=np.zeros(len(pos), np.int32), cfg=cfg, masses=[SI_MASS], lammps_names=["Si"],
                    lammps_types=types)
    if config == 3:
        reps = (10, 16, 20) if not ncell else (ncell, ncell, ncell)
        cell, pos, types = lmp_like.li3po4(reps)
        cfg = model_file.model_S(type_names=["Li", "P", "O"], avg_num_neighbors=48.6)
        names = lmp_like.LI3PO4_LAMMPS_NAMES                      # pair_coeff * * f Li P O O
        mapper = np.array([cfg["type_names"].index(s) for s in names], dtype=np.int32)
        return dict(name=f"{len(pos)}-atom Li3PO4 ({reps[0]}x{reps[1]}x{reps[2]} Pnma cells, 4 LAMMPS types Li P O1 O2 -> model types "
                         f"Li P O), model S (l_max=1, U=32, S=64, 2 layers)",
                    cell=cell, pos=pos, mtype=mapper[types - 1], cfg=cfg,
                    masses=[lmp_like.LI3PO4_MASSES[s] for s in cfg["type_names"]], lammps_names=names, lammps_types=types)
    if config == 5:
        m = ncell or 55
        cell, pos, types = lmp_like.water(m)
        cfg = model_file.model_L(avg_num_neighbors=53.6)
        return dict(name=f"{len(pos)}-atom water ({m}^3 molecules, O/H), model L (l_max=2, U=64, S=64, 3 layers)",
                    cell=cell, pos=pos, mtype=(types - 1).astype(np.int32), cfg=cfg,
                    masses=[lmp_like.WATER_MASSES[s] for s in cfg["type_names"]], lammps_names=["O", "H"], lammps_types=types)
    if config == 6:
        # not a BASELINE config: config 5's water box with the reference test YAML's own model shape (l_max = 2, 32 tensor features, 3 layers,
        # /root/reference/tests/test_data/test_repro_allegro.yaml:89-99) -- the shape k_fused_lx serves
        m = ncell or 55
        cell, pos, types = lmp_like.water(m)
        cfg = model_file.model_L(num_tensor_features=32, avg_num_neighbors=53.6)
        return dict(name=f"{len(pos)}-atom water ({m}^3 molecules, O/H), reference-YAML model shape (l_max=2, U=32, S=64, 3 layers)",
                    cell=cell, pos=pos, mtype=(types - 1).astype(np.int32), cfg=cfg,
                    masses=[lmp_like.WATER_MASSES[s] for s in cfg["type_names"]], lammps_names=["O", "H"], lammps_types=types)
    raise SystemExit(f"bench.py: unknown --config {config} (2, 3, 4, 5 or 6)")


def kernel_source_hash() -> str:
    """sha256[:16] over the kernel sources: stored with every profiles/traffic.json entry, compared here."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "pair_allegro_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "pair_allegro_amd", "csrc", "*.h"))):
        h.update(open(f, "rb").read())
    # the Makefile carries per-kernel compiler options: its rules count, its comments do not
    for line in open(os.path.join(ROOT, "pair_allegro_amd", "csrc", "Makefile"), "rb").read().splitlines():
        if not line.lstrip().startswith(b"#"):
            h.update(line + b"\n")
    return h.hexdigest()[:16]


def spawn_ranks(n: int, budget_s: float = None) -> int:
    """`python bench.py --gpus N` typed as is (no launcher): start N fresh rank processes BEFORE anything touches the GPU
    (never re-exec a process that has initialised HIP) and relay their exit status; rank 0 prints the JSON line.
    Fail fast: all children are polled; the first one that exits non-zero -- or a wall-clock budget
    (AHIP_BENCH_BUDGET_S, default 900 s) running out -- ends the others (a rank dying inside ncclCommInitRank would otherwise leave its
    peers in a collective for the launcher's whole timeout) and the parent exits non-zero.  What mpirun does for the reference
    (/root/reference/README.md:37-40)."""
    import signal
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    if budget_s is None:
        budget_s = float(os.environ.get("AHIP_BENCH_BUDGET_S", "900"))
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    t0 = time.monotonic()
    rc, why = 0, None
    live = set(range(n))
    while live and why is None:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0:
                rc, why = abs(code) or 1, f"rank {r} exited with status {code}"
                break
        if why is None and live and time.monotonic() - t0 > budget_s:
            rc, why = 124, f"wall-clock budget of {budget_s:.0f} s spent"
        if why is None and live:
            time.sleep(0.1)
    if why is not None:
        print(f"bench.py: {why}; stopping the other ranks", file=sys.stderr, flush=True)
        for r in live:                       # exactly the processes started above
            procs[r].send_signal(signal.SIGTERM)
        t1 = time.monotonic()
        for r in live:
            try:
                procs[r].wait(timeout=max(0.1, 10.0 - (time.monotonic() - t1)))
            except subprocess.TimeoutExpired:
                procs[r].kill()
                procs[r].wait()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=4, help="BASELINE.json config: 2 (10k Si), 3 (100k Li3PO4), 4 (1M Si, the metric), 5 (500k water, model L); 6 = config 5's box with the reference YAML's model shape (U=32)")
    ap.add_argument("--ncell", type=int, default=0, help="override the replication of the chosen config (smaller boxes for quick runs)")
    ap.add_argument("--path", default="auto", choices=["auto", "fused", "generic"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-ncell", type=int, default=11)
    ap.add_argument("--no-overlap", action="store_true", help="serial ghost exchange (A/B against the overlapped schedule)")
    ap.add_argument("--mlp-depth", type=int, default=0, help="configs 2 / 3 / 4 only: hidden layers of the two-body and latent MLPs (default: the model's 2; 1 and 3 select the depth instances of k_fused)")
    ap.add_argument("--l-max", type=int, default=-1, help="override the model's l_max (e.g. 3 on config 2: a shape outside every fused kernel, i.e. a line of the layer-at-a-time path on an l_max = 3 model)")
    ap.add_argument("--eval-only", action="store_true", help="ablation runs only (timing builds that compute wrong forces): time force evaluations at fixed positions instead of NVE steps; the line is NOT a benchmark result")
    ap.add_argument("--force-overlap", action="store_true", help="overlapped three-range schedule even on one rank (A/B: what the schedule itself costs)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))

    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        # A rank that hangs (a collective whose peer died, an exchange kernel that never gets a CU) must not sit there until the launcher's own
        # timeout: after the wall-clock budget every rank prints where its threads are and exits non-zero.
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ.get("AHIP_BENCH_BUDGET_S", "900")), exit=True)

    import torch
    from pair_allegro_amd import capi, md, model_file

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    dist = None
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: allegro-hip has no CPU fallback")
    # AHIP_BENCH_ONE_DEVICE=1 (debugging / tests only, tests/test_gpu_md.py): every rank runs on cuda:0 and the messages are staged
    # through gloo -- RCCL does not form a communicator between processes that share a GPU.  It exercises this file's multi-rank
    # logic (decomposition, max-over-ranks timing, the JSON line) on a 1-GPU box; its numbers are not benchmark numbers.
    one_device = world > 1 and os.environ.get("AHIP_BENCH_ONE_DEVICE") == "1"
    dev_index = 0 if one_device else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    dist_raw = None
    if world > 1:
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist_raw = dist_mod
        if one_device:
            dist_mod.init_process_group(backend="gloo")
            dist = md.HostStagedDist(dist_mod)
        else:
            dist_mod.init_process_group(backend="nccl", device_id=device)
            dist = dist_mod

    wl = workload(args.config, args.ncell)
    cfg = wl["cfg"]
    if args.mlp_depth:
        cfg["mlp_depth"] = args.mlp_depth
        wl["name"] += f", MLP depth {args.mlp_depth}"
    if args.l_max >= 0:
        cfg["l_max"] = args.l_max
        wl["name"] += f", l_max overridden to {args.l_max}"
    weights = model_file.init_weights(cfg)
    tmpdir = tempfile.mkdtemp(prefix="ahip_bench_")
    model_path = os.path.join(tmpdir, f"model_{rank}.ahip")
    model_file.save_ahip(model_path, cfg, weights)

    lib = capi.Library()
    model = capi.Model(model_path, dev_index, lib)
    if one_device:
        model.set_option("edge_schedule", "dynamic")      # several processes share the GPU: no resident-grid assumption
    model.set_option("path", args.path)
    model.set_option("timing", "1")

    pos = wl["pos"]
    natoms = len(pos)
    box = np.diag(wl["cell"])
    mass_by_mtype = np.asarray(wl["masses"], dtype=np.float64)
    vel = md.maxwell_boltzmann(natoms, mass_by_mtype[wl["mtype"]], 300.0, 12345)
    grid = md.choose_grid(world)
    backend = md.HipBackend(model, wl["masses"])
    sim = md.Simulation(backend, box, cfg["r_max"], 1.0, pos, wl["mtype"], vel, device,
                        grid=grid, rank=rank, dist=dist, dt=0.001, overlap=False if args.no_overlap else (True if args.force_overlap else None))
    sim.setup()
    ci = getattr(sim, "comm_info", None) or {"transport": "none", "rccl_version": 0, "init_ms": 0.0}
    if world > 1:          # one line per rank on stderr: what a first multi-GPU contact needs to see
        print(f"[bench] rank {rank}/{world} device {dev_index}: ghost exchange transport={ci['transport']} rccl_version={ci['rccl_version']} "
              f"comm_init_ms={ci['init_ms']} nlocal={sim.nlocal} nghost={sim.nall - sim.nlocal}", file=sys.stderr, flush=True)
    if os.environ.get("AHIP_BENCH_TEST_KILL_RANK", "") == str(rank):        # tests only (tests/test_gpu_md.py): a rank that dies after start-up
        os._exit(17)
    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # Exchange schedule with more than one rank: overlapped (three centre ranges per step, the exchange beside the interior ones) or serial
    # (exchange, one call, exchange).  Which one wins depends on what the exchange costs on the machine at hand against the fixed cost of two more
    # range calls (0.2 ms at 125 k atoms per rank, 1.4 ms at 1 M: profiles/r04_f_final.md), so unless a flag says otherwise both are timed for
    # a few steps (minimum step time of each, maximum over ranks) before the warm-up and the faster one runs the benchmark.
    autotune = None
    if world > 1 and not args.no_overlap and not args.force_overlap and sim.overlap:
        def min_step_ms(n):
            best = float("inf")
            for _ in range(n):
                barrier()
                t = time.perf_counter()
                sim.step()
                torch.cuda.synchronize()
                best = min(best, 1e3 * (time.perf_counter() - t))
            tt = torch.tensor([best], dtype=torch.float64, device=device)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            return float(tt.item())
        sim.step()                                   # first-launch costs out of the way
        t_ov = min_step_ms(3)
        sim.set_overlap(False)
        sim.step()
        t_se = min_step_ms(3)
        use_overlap = t_ov <= t_se                   # identical on every rank (all-reduced times)
        sim.set_overlap(use_overlap)
        autotune = {"overlapped_ms": round(t_ov, 3), "serial_ms": round(t_se, 3), "chosen": "overlapped" if use_overlap else "serial"}
    step = sim.compute_forces if args.eval_only else sim.step
    for _ in range(args.warmup):
        step()

    nrebuild0 = sim.nrebuild
    # Stage timings: the library records HIP events on its launch stream around every stage of every call and nobody waits for them
    # inside the timed region; they are read once behind the closing barrier (sums and launch counts over exactly the K timed steps).
    backend.stats = {}
    backend.defer_stats = True
    model.timings()                        # drop what warm-up and setup recorded
    sim.time_comm = True
    sim.comm_ms()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    stage_sum, stage_cnt = model.timings_and_counts()
    stage_sum["comm"] = sim.comm_ms()                 # device time of the library's forward + reverse exchanges (events on their stream)
    stage_cnt["comm"] = 2 * args.steps
    sim.time_comm = False
    backend.stats = None
    STAGES = ["edge_build", "tile_pack", "model_fused", "model_generic", "comm"]
    if dist is not None:
        # max over ranks of the wall time and of every stage's summed device time (the slowest rank sets the step)
        tmax = torch.tensor([dt] + [stage_sum.get(k, 0.0) for k in STAGES], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax[0].item())
        stage_max = {k: float(tmax[1 + i].item()) for i, k in enumerate(STAGES) if float(tmax[1 + i].item()) > 0.0}
    else:
        stage_max = dict(stage_sum)
    # edges of this rank's centres: one un-timed evaluation of all of them in a single call (the timed calls do not read their counts back)
    scratch_f = torch.zeros_like(sim.f)
    scratch_ev = torch.zeros(7, dtype=torch.float64, device=device)
    backend.compute(sim.x, sim.mtype, scratch_f, sim.nlocal, scratch_ev)
    edges_rank0 = model.nedges()
    slots_used, slots_total = model.tile_occupancy()
    model.timings()
    del scratch_f
    th = sim.thermo(wl["masses"])
    rebuilds_timed = sim.nrebuild - nrebuild0
    # cost of one re-neighboring (migration, borders, cell list + neighbor table), measured outside the timed region: the metric counts
    # amortised rebuilds; a short timed window may contain none
    barrier()
    tr0 = time.perf_counter()
    sim.rebuild()
    barrier()
    rebuild_ms = 1e3 * (time.perf_counter() - tr0)
    # ... and how often the run re-neighbors: when the timed window held no re-neighboring, the cadence is MEASURED on an untimed
    # continuation of the same trajectory (up to 96 steps, until two re-neighborings have happened) and `value_with_amortised_rebuilds` prices it in
    cadence_steps, cadence_rebuilds = 0, 0
    if rebuilds_timed == 0 and not args.eval_only:
        nr0 = sim.nrebuild
        while cadence_steps < 96 and sim.nrebuild - nr0 < 2:
            sim.step()
            cadence_steps += 1
        cadence_rebuilds = sim.nrebuild - nr0
        barrier()

    if rank == 0:
        ms_per_step = 1e3 * dt / args.steps
        value = natoms * args.steps / dt
        used_path = model.last_path
        # ---- roofline of the dominant kernel -------------------------------------------------
        tb_tab = used_path.startswith("fused_") and os.environ.get("AHIP_FUSED_TB", "table") != "mlp"
        # arithmetic of the fused kernels' dense contractions, by kernel path: every one but tf32eq is float32 or float32-equivalent
        ARITH = {"fused_f32": "f32-input MFMA (v_mfma_f32_16x16x4_f32): exact float32 fmaf chains",
                 "fused_f16x2": "f16x2: every float32 operand as two float16 terms (round-to-nearest, remainder scaled by 2^11: 22 bits + sign), three "
                                "v_mfma_f32_16x16x32_f16 products per float32 product, float32 accumulate; backward pass scaled by a power of two; "
                                "float32-equivalent: max|dF| vs the float64 oracle as the f32 form (parity_vs_oracle; tests/test_arith_emulation.py)",
                 "fused_bf16x3": "bf16x3: exact three-way bf16 split of both operands, six bf16-MFMA products, float32 accumulate; float32-equivalent",
                 "fused_tf32eq": "two-term bf16 split, three products: TF32-class, only for model files with allow_tf32 = 1"}
        # the arithmetic type the path computes in (not a precision claim: parity_vs_oracle carries the numbers)
        DTYPE = {"fused_f16x2": "f32-equivalent (f16x2 split on the f16 matrix cores, f32 accumulate)", "fused_bf16x3": "f32-equivalent (bf16x3 split, f32 accumulate)",
                 "fused_tf32eq": "tf32-class (two-term bf16 split; model file sets allow_tf32)", "generic_f64": "f64"}
        # ALGORITHMIC flops = the model's dense contractions (SURVEY 8d / DESIGN 4.2), whatever the kernel does with them;
        # the fused kernels' tabulated two-body embedding executes fewer: frac_executed is priced on those.
        flops_per_edge = 2.0 * model_macs_per_edge(cfg) * 2.0       # 2 flop per MAC x (forward + input-gradient backward)
        executed_flops_per_edge = 2.0 * model_macs_per_edge(cfg, two_body_tabulated=tb_tab, readout_folded=tb_tab and cfg["l_max"] == 1) * 2.0
        stage_avg = {k: float(v) / args.steps for k, v in stage_sum.items()}          # ms per step, summed over the calls of a step
        launches_per_step = {k: stage_cnt[k] / args.steps for k in stage_cnt}
        dom = max((k for k in stage_avg if k.startswith("model")), key=lambda k: stage_avg[k], default=None)
        roof = None
        traffic, traffic_src, traffic_hash = None, None, None
        tj = os.path.join(ROOT, "profiles", "traffic.json")          # measured in separate --pmc passes (cannot be collected live)
        if os.path.exists(tj) and world == 1:
            ent = json.load(open(tj)).get(f"config{args.config}:{used_path}:{natoms}")
            if ent:
                traffic, traffic_src = ent["traffic_bytes_per_launch"], ent["source"]
                traffic_hash = ent.get("kernel_hash")
        if dom is not None:
            t_dom = stage_avg[dom] * 1e-3                       # seconds of the dominant kernel per step (all its launches: one evaluation of edges_rank0 edges)
            ach = flops_per_edge * edges_rank0 / t_dom / 1e12
            # Headline roof (north-star wording): the HBM roofline on SURVEY 8d's ALGORITHMIC bytes -- bytes function above x the edges one
            # evaluation processes / the kernel's HIP-event time of this run / 8 TB/s.  A fused kernel moves fewer bytes than the layer-at-a-time
            # formulation the figure describes (`traffic`, from PMC passes, says how many), so this fraction may exceed 1: SURVEY 8d's own caveat.
            bytes_per_edge = model_algorithmic_bytes_per_edge(cfg, edges_rank0 / max(sim.nlocal, 1))
            alg_bytes = bytes_per_edge * edges_rank0
            gbs_alg = alg_bytes / t_dom / 1e9
            # the matrix pipe the kernel's contractions really run on, and how many of its products one float32 product costs
            PIPE = {"fused_f32": ("f32-input MFMA", 157.3, 1.0), "fused_f16x2": ("f16 MFMA (dense)", 2500.0, 3.0),
                    "fused_bf16x3": ("bf16 MFMA (dense)", 2500.0, 6.0), "fused_tf32eq": ("bf16 MFMA (dense)", 2500.0, 3.0)}
            pipe = PIPE.get(used_path, ("f32-input MFMA (GEMM stages only)", 157.3, 1.0))
            ex_tf = ach * executed_flops_per_edge / flops_per_edge * pipe[2]
            roof = {"bound": "hbm", "achieved": round(gbs_alg, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(gbs_alg / 8000.0, 4),
                    "traffic": traffic, "traffic_source": traffic_src, "kernel": dom,
                    "basis": "SURVEY 8d algorithmic bytes (bench.py: model_algorithmic_bytes_per_edge) x edges_per_launch / avg_ms; may exceed 1 for a fused kernel (SURVEY 8d caveat): see hbm_measured for the bytes really moved",
                    "algorithmic_bytes_per_edge": round(bytes_per_edge, 1), "algorithmic_bytes_per_launch": alg_bytes,
                    "avg_ms": round(stage_avg[dom], 3), "edges_per_launch": edges_rank0, "launches_per_step": launches_per_step[dom],
                    "flops_per_edge": flops_per_edge, "executed_flops_per_edge": executed_flops_per_edge,
                    # the model's float32 contractions per second; NOT a fraction of anything for the split arithmetics (they issue no f32 MFMA)
                    "f32_equivalent_tflops": round(ach, 3),
                    "matrix_pipe": {"unit": pipe[0], "products_per_f32_product": pipe[2], "achieved": round(ex_tf, 1), "peak": pipe[1], "unit_of_measure": "TFLOP/s executed",
                                    "frac": round(ex_tf / pipe[1], 4)},
                    "two_body": "table" if tb_tab else "mlp",
                    "arithmetic": ARITH.get(used_path, "float32 (layer-at-a-time kernels)"),
                    # padding tax of the tile packing: edge slots that held an edge / slots of all tiles (one full evaluation)
                    "slots_used": slots_used, "slots_total": slots_total, "slot_occupancy": (round(slots_used / slots_total, 4) if slots_total else None)}
            if traffic:
                # the byte counts come from committed --pmc passes: flag them when the kernel sources changed since
                roof["traffic_kernel_hash"] = traffic_hash
                roof["traffic_stale"] = traffic_hash != kernel_source_hash()
                # north-star wording: ">= 40 % of the HBM roofline on the neighbor-gather + tensor-product kernels": the model kernel's
                # measured HBM bytes (committed --pmc passes) over its HIP-event time of this run, next to the MFMA figure above
                hb = traffic / (stage_avg[dom] * 1e-3) / 1e9
                roof["hbm_measured"] = {"achieved": round(hb, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(hb / 8000.0, 4)}
            vj = os.path.join(ROOT, "profiles", "vector_memory_path.json")
            if os.path.exists(vj):
                # What actually bounds the f16x2 kernels is neither figure above: the CU's vector-memory data-return unit (TD, 64 B/clk) -- weight fragments,
                # saved rows and gathers all return through it.  Its busy fraction comes from committed --pmc passes on a reduced box of the same workload.
                vent = json.load(open(vj)).get(f"config{args.config}:{used_path}")
                if vent:
                    roof["vector_memory_path"] = {"td_busy": vent["td_busy"], "ta_busy": vent["ta_busy"], "byte_utilisation": vent.get("return_path_byte_utilisation"), "natoms_measured": vent["natoms_measured"],
                                                  "source": vent["source"], "stale": vent["kernel_hash"] != kernel_source_hash()}
            if "edge_build" in stage_avg:
                # the neighbor gather (HBM-bound): algorithmic bytes per list entry 4 (j) + 24 (x_j) + 4 (type_j), per edge 20
                # (e_ii, e_j, rvec) + 1 (packed types) -- DESIGN.md 4.1
                nb = model.nneigh() * 32.0 + edges_rank0 * 21.0
                gbs = nb / (stage_avg["edge_build"] * 1e-3) / 1e9
                roof["neighbor_gather"] = {"bound": "hbm", "achieved": round(gbs, 1), "peak": 8000.0, "unit": "GB/s",
                                           "frac": round(gbs / 8000.0, 4), "avg_ms": round(stage_avg["edge_build"], 4),
                                           "algorithmic_bytes": nb}
                if os.path.exists(tj) and world == 1:         # HBM-side bytes of k_build_edges from the committed --pmc passes
                    ent = json.load(open(tj)).get(f"config{args.config}:k_build_edges:{natoms}")
                    if ent:
                        roof["neighbor_gather"]["traffic"] = ent["traffic_bytes_per_launch"]
                        roof["neighbor_gather"]["traffic_source"] = ent["source"]
        # ---- CPU baseline + max|dF| on a bounded sample --------------------------------------
        cpu = None
        max_df = None
        parity = None
        if not args.no_cpu_baseline and world == 1:       # reported baseline: rank 0 at N = 1 only
            cpu, parity = cpu_baseline_and_parity(lib, args.config, dev_index, args.cpu_sample_ncell, args.path)
            max_df = parity["max_abs_dF"]
        # the metric counts amortised re-neighborings: a timed window that held some has paid for them; one that held none is corrected with
        # the measured cost of one re-neighboring and the measured cadence of this trajectory
        value_amortised = value
        if rebuilds_timed == 0 and cadence_rebuilds > 0:
            value_amortised = natoms / ((ms_per_step + rebuild_ms * cadence_rebuilds / cadence_steps) * 1e-3)
        out = {
            "metric": "atom_steps_per_sec" if not args.eval_only else "atom_evaluations_per_sec (ablation run, not the benchmark)", "value": round(value, 1), "unit": "atom-steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": DTYPE.get(used_path, "f32"), "data": "synthetic",
            "config": {"workload": f"{('BASELINE config ' + str(args.config)) if args.config <= 5 else 'extra config 6 (not in BASELINE.json)'}: {wl['name']}, r_max {cfg['r_max']} A + skin 1.0 A, NVE dt=1 fs",
                       "grid": "x".join(map(str, grid)), "kernel_path": used_path, "arith_note": model.arith_note, "rebuilds": sim.nrebuild,
                       "rebuilds_in_timed_steps": rebuilds_timed, "rebuild_ms": round(rebuild_ms, 3),
                       "steps_per_rebuild": (round(args.steps / rebuilds_timed, 1) if rebuilds_timed else None),
                       "rebuild_cadence_measured": ({"steps": cadence_steps, "rebuilds": cadence_rebuilds} if cadence_steps else None),
                       "comm": "overlapped" if sim.overlap else "serial", "comm_autotune": autotune,
                       "comm_transport": (("library/" + sim.comm.transport + ("/single-rank gather-scatter" if world == 1 else "")) if getattr(sim, "comm", None) is not None else "torch.distributed"),
                       "stage_ms_rank0": {k: round(v, 3) for k, v in stage_avg.items()},
                       "stage_ms": {k: round(v / args.steps, 3) for k, v in stage_max.items()},          # max over ranks, per step
                       "comm_ms": round(stage_max.get("comm", 0.0) / args.steps, 4),
                       "rccl_version": ci["rccl_version"], "comm_init_ms": ci["init_ms"],
                       "pe_per_atom": th["pe"] / natoms},
            "value_with_amortised_rebuilds": round(value_amortised, 1),
            "max_abs_dF_vs_oracle": max_df,
            "parity_vs_oracle": parity,
            "roofline": roof,
            "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    model.close()
    if dist is not None:
        dist.barrier()
        dist_raw.destroy_process_group()
    if world > 1:
        import faulthandler
        faulthandler.cancel_dump_traceback_later()


def cpu_sample(config: int, ncell: int):
    """Bounded sample of the same workload for the CPU leg (about 10-30 s of host work)."""
    if config in (2, 4):
        return workload(2, ncell or 11)                       # 10 648-atom Si = BASELINE configs[1] (SURVEY 8d sizes)
    if config == 3:
        wl = workload(3, 0)
        from pair_allegro_amd import lmp_like
        cell, pos, types = lmp_like.li3po4((3, 5, 6))         # 2 880 atoms, same cell
        mapper = np.array([wl["cfg"]["type_names"].index(s) for s in wl["lammps_names"]], dtype=np.int32)
        return dict(wl, name="2880-atom Li3PO4 (3x5x6 cells)", cell=cell, pos=pos, lammps_types=types, mtype=mapper[types - 1])
    return workload(config if config == 6 else 5, 6)          # 648-atom water, model L / Y (a CPU evaluation of model L costs ~20 ms per atom)


def cpu_baseline_and_parity(lib, config, device_index, ncell, path):
    """Times the oracle (torch CPU, all host cores; a port of the reference's libtorch path) on a bounded sample of the
    benchmarked workload and measures max|dF| of the HIP path against it on the same configuration."""
    import torch
    from oracle import allegro_torch, glue
    from pair_allegro_amd import capi, lmp_like, model_file
    wl = cpu_sample(config, ncell)
    cfg = wl["cfg"]
    weights = model_file.init_weights(cfg)
    tmpdir = tempfile.mkdtemp(prefix="ahip_bench_cpu_")
    model_path = os.path.join(tmpdir, "model.ahip")
    model_file.save_ahip(model_path, cfg, weights)
    names = wl["lammps_names"]
    rs = lmp_like.build_rank_system(wl["cell"], wl["pos"], wl["lammps_types"], cfg["r_max"] + 1.0)
    mapper = np.array([cfg["type_names"].index(s) for s in names], dtype=np.int32)
    cm = np.full((len(names), len(names)), cfg["r_max"])
    oracle = torch.jit.script(allegro_torch.build(cfg, weights).eval())
    f_ref = np.zeros_like(rs.x)
    e_ref = np.zeros(len(rs.x))
    pe_ref, vir_ref, _ = glue.compute(oracle, rs.x, rs.type, rs.nlocal, rs.ilist, rs.numneigh, rs.firstneigh, mapper, cm, f_ref, e_ref)
    harness = os.path.join(ROOT, "oracle", "_build", "cpu_baseline")
    cpu = None
    if os.path.exists(harness):
        # SURVEY 8d protocol: the contract-conforming TorchScript file executed by libtorch (C++) with the reference's own call
        # sequence (oracle/cpu_baseline.cpp: jit::load + freeze + preprocess -> forward(Dict) -> scatter), 3 warm-up + >= 10 timed
        pth = os.path.join(tmpdir, "model.nequip.pth")
        allegro_torch.export_nequip_pth(pth, cfg, weights)
        import subprocess

        def write_sys(r):
            sysf = os.path.join(tmpdir, f"sys_{r.nlocal}.bin")
            with open(sysf, "wb") as fh:
                np.array([r.nlocal, r.nghost, len(names), int(r.offsets[-1])], dtype=np.int32).tofile(fh)
                r.x.astype(np.float64).tofile(fh); r.type.astype(np.int32).tofile(fh); r.tag.astype(np.int32).tofile(fh)
                r.numneigh.astype(np.int32).tofile(fh); r.flat.astype(np.int32).tofile(fh)
            return sysf

        def run(sysf, threads, warm, reps, budget, bind=None):
            env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads))
            if bind:
                env["OMP_PROC_BIND"] = bind
            pr = subprocess.run([harness, sysf, pth, "--warmup", str(warm), "--reps", str(reps), "--budget", str(budget), "--threads", str(threads)]
                                + list(names), stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
            if pr.returncode != 0:
                return None
            d = json.loads(pr.stdout.decode().strip().splitlines()[-1])
            d["bind"] = bind or "unset"
            return d

        # Thread count: all host cores oversubscribe this model (r02: 128 threads = 500 ms for 64 atoms, 8 threads = 30 ms), so the count is
        # swept {8, 16, 32, 64, 128} (capped at the host's cores) on a small sample of the same workload, then OMP_PROC_BIND close / spread
        # at the best count; the benchmarked sample is then timed with the winner: 3 warm-up + 10 timed evaluations when one takes
        # under 2 s, else 1 + 3 inside a ~25 s budget.
        ncores = os.cpu_count() or 8
        counts = [t for t in (8, 16, 32, 64, 128) if t <= ncores] or [ncores]
        if config in (2, 4):
            c_s, p_s, t_s = lmp_like.diamond_si(5)                       # 1 000-atom Si for the sweep
            sweep_rs, sweep_label = lmp_like.build_rank_system(c_s, p_s, t_s, cfg["r_max"] + 1.0), "1000-atom bulk Si"
        else:
            sweep_rs, sweep_label = rs, wl["name"]
        sweep_sys = write_sys(sweep_rs)
        sweep = []
        heavy_sweep = cfg["l_max"] >= 2
        for t in counts:
            d = run(sweep_sys, t, 1, 1 if heavy_sweep else 3, 8)
            if d:
                sweep.append({"threads": t, "bind": "unset", "ms_model": d["ms_model"]})
        best_t, best_bind = (min(sweep, key=lambda r: r["ms_model"])["threads"], None) if sweep else (min(8, ncores), None)
        if sweep and not heavy_sweep:
            for bind in ("close", "spread"):
                d = run(sweep_sys, best_t, 1, 3, 8, bind)
                if d:
                    sweep.append({"threads": best_t, "bind": bind, "ms_model": d["ms_model"]})
            w = min(sweep, key=lambda r: r["ms_model"])
            best_t, best_bind = w["threads"], (None if w["bind"] == "unset" else w["bind"])
        runs = []
        samples = [(wl["name"], rs)]
        if config in (2, 4):                                          # SURVEY 8d sizes: 64 and 10 648 atoms; 1 728 atoms: the largest box that affords the full 3 + 10 protocol
            c64, p64, t64 = lmp_like.diamond_si(2)
            c2k, p2k, t2k = lmp_like.diamond_si(6)
            samples.insert(0, ("1728-atom bulk Si", lmp_like.build_rank_system(c2k, p2k, t2k, cfg["r_max"] + 1.0)))
            samples.insert(0, ("64-atom bulk Si (config 1)", lmp_like.build_rank_system(c64, p64, t64, cfg["r_max"] + 1.0)))
        for label, r in samples:
            sysf = write_sys(r)
            probe = run(sysf, best_t, 1, 1, 1, best_bind)             # one evaluation to choose the protocol
            fast = probe is not None and probe["ms_model"] < 2000.0
            d = run(sysf, best_t, 3 if fast else 1, 10 if fast else 3, 25, best_bind)
            if d:
                d["sample"] = label
                runs.append(d)
        if runs:
            main_run = runs[-1]
            cpu = {"value": round(main_run["nlocal"] / (main_run["ms_model"] * 1e-3), 1), "unit": "atom-steps/s", "cores": main_run["threads"],
                   "kind": "port", "host_cores": ncores, "omp_proc_bind": main_run["bind"],
                   "sample": f"{main_run['sample']}: libtorch C++ harness (oracle/cpu_baseline.cpp, the reference's load / freeze / "
                             f"preprocess / forward(Dict) / scatter sequence) on the oracle's TorchScript export, float32 model, "
                             f"{main_run['threads']} intra-op threads (best of the sweep), "
                             f"{main_run['warmup']} warm-up + {main_run['reps']} timed evaluations, {main_run['ms_model']:.0f} ms each (model only), "
                             f"{main_run['ms_total']:.0f} ms with preprocess + scatter; "
                             f"{1e3 * main_run['ms_model'] / max(main_run['nedges'], 1):.1f} us per edge",
                   # SURVEY 8d asks for 3 warm-up + >= 10 timed evaluations; the bounded-sample rule (10-30 s of CPU work) allows that only
                   # while one evaluation takes under 2 s -- stated per run so nobody has to infer it from the text above
                   "protocol": (f"{main_run['warmup']} warm-up + {main_run['reps']} timed evaluations; rule: 3 + 10 when one evaluation takes < 2 s, "
                                f"else 1 + 3 inside a 25 s budget (this sample: {main_run['ms_model'] / 1e3:.2f} s per evaluation)"),
                   "value_glue_inclusive": round(main_run["nlocal"] / (main_run["ms_total"] * 1e-3), 1),
                   # ... and the full SURVEY 8d protocol (3 warm-up + 10 timed) on the sample(s) small enough to allow it inside the budget, next to the line above
                   "full_protocol_runs": [{"sample": r["sample"], "value": round(r["nlocal"] / (r["ms_model"] * 1e-3), 1), "unit": "atom-steps/s", "warmup": r["warmup"],
                                           "reps": r["reps"], "ms_model": r["ms_model"]} for r in runs if r["warmup"] >= 3 and r["reps"] >= 10],
                   "thread_sweep": {"sample": sweep_label, "runs": sweep},
                   "runs": [{k: r[k] for k in ("sample", "nlocal", "nedges", "threads", "bind", "warmup", "reps", "ms_model", "ms_total")} for r in runs]}
    if cpu is None:
        inp = glue.preprocess(rs.x, rs.type, rs.nlocal, rs.ilist, rs.numneigh, rs.firstneigh, mapper, cm)
        tin = {k: torch.from_numpy(v) for k, v in inp.items()}
        for _ in range(3):
            oracle(tin)
        reps, t0 = 0, time.perf_counter()
        while reps < 3 or (time.perf_counter() - t0 < 20.0 and reps < 10):
            oracle(tin)
            reps += 1
        t_eval = (time.perf_counter() - t0) / reps
        cpu = {"value": round(rs.nlocal / t_eval, 1), "unit": "atom-steps/s", "cores": torch.get_num_threads(), "kind": "port",
               "sample": f"{wl['name']}: 3 warm-up + {reps} timed force evaluations of the TorchScript oracle through torch (Python; the C++ "
                         f"harness oracle/_build/cpu_baseline was not built), {t_eval*1e3:.0f} ms each (model only)"}
    def hip_eval(arith):
        m = capi.Model(model_path, device_index, lib)
        m.set_option("path", path)
        if arith:
            m.set_option("fused_arith", arith)
        m.neigh_update_csr(rs.nall, rs.ilist, rs.offsets, rs.flat)
        f = np.zeros_like(rs.x)
        e = np.zeros(len(rs.x))
        pe, vir = m.compute(rs.nlocal, rs.nghost, rs.x, rs.type, mapper, cm, f, e)
        used = m.last_path
        m.close()
        return f, e, pe, vir, used
    f, e, pe, vir, used = hip_eval(None)
    max_df = float(np.abs(f - f_ref).max())
    # the same sample on the float32 instance of the same kernel (f32-input MFMA = exact fmaf chains): what a split arithmetic is held against is the
    # float32 kernel's own distance from the float64 oracle on THIS sample, not an absolute number
    max_df_f32, used_f32 = None, None
    if used.startswith("fused_") and used != "fused_f32":
        try:
            f32f, _, _, _, used_f32 = hip_eval("f32")
            max_df_f32 = float(np.abs(f32f - f_ref).max())
        except Exception as ex:            # a model shape without a float32 instance (MLP depth 1 / 3): the layer-at-a-time float32 kernels answer
            used_f32 = f"unavailable: {ex}"
    # the other observables of the reference's own comparison: per-atom energy, PE per atom, virial per atom
    parity = {"max_abs_dF": max_df, "max_abs_dEatom": float(np.abs(e[: rs.nlocal] - e_ref[: rs.nlocal]).max()),
              "abs_dPE_per_atom": float(abs(pe - pe_ref) / rs.nlocal),
              "max_abs_dvirial_per_atom": float(np.abs(vir - vir_ref).max() / rs.nlocal), "atoms": int(rs.nlocal),
              "kernel_path": used, "sample": wl["name"],
              "max_abs_dF_f32_instance": max_df_f32, "f32_instance_path": used_f32,
              "split_over_f32_instance": (round(max_df / max_df_f32, 3) if max_df_f32 else None),
              "max_abs_F": float(np.abs(f_ref).max())}
    return cpu, parity


if __name__ == "__main__":
    main()
