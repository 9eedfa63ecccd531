"""Stand-alone MD driver around the C-ABI: what LAMMPS does around `pair->compute` in the reference's
test deck (`units metal`, `newton on`, `neighbor 1.0 bin`, `fix nve`, `run N`;
/root/reference/tests/test_python_repro_allegro.py:84-120), reduced to the pieces the hot path needs:

* brick domain decomposition of a periodic orthogonal box over a px*py*pz process grid
  (one process per GPU, /root/reference/pair_nequip_allegro.cpp:92-120),
* `borders`: ghost atoms = periodic images / neighbour-rank atoms within r_max+skin of a face,
  built dimension by dimension (x, then y, then z) so edges and corners propagate,
* `forward_comm` (ghost positions) and `reverse_comm` (ghost forces summed into owners, required by
  `newton on`, pair_nequip_allegro.cpp:149,366-368) every step -- torch.distributed P2P
  (backend nccl == RCCL over xGMI on the GPU box, gloo in the CPU tests); a periodic dimension with
  one rank is a local copy,
* neighbor rebuild when any atom moved more than skin/2, atom migration between bricks,
* velocity-Verlet NVE.

All per-atom state lives in torch tensors on the compute device (plumbing); neighbor build, force
evaluation and integration go through a backend.  The product backend is :class:`HipBackend`
(liballegro_hip.so, device pointers, no copies).  Tests inject a CPU backend to exercise the
decomposition / exchange logic with gloo.
"""
from __future__ import annotations

import math
import os
import time
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

FTM2V = 1.0 / 1.0364269e-4          # metal units: (eV/A)/(g/mol) -> A/ps^2
MVV2E = 1.0364269e-4
KB = 8.617343e-5


def choose_grid(nranks: int) -> Tuple[int, int, int]:
    """1,2,4,8 -> 1x1x1, 2x1x1, 2x2x1, 2x2x2; general: most cubic factorisation."""
    best = (nranks, 1, 1)
    for px in range(1, nranks + 1):
        if nranks % px:
            continue
        for py in range(1, nranks // px + 1):
            if (nranks // px) % py:
                continue
            pz = nranks // px // py
            cand = tuple(sorted((px, py, pz), reverse=True))
            if max(cand) - min(cand) < max(best) - min(best):
                best = cand
    return best


class HipBackend:
    """Device-resident calls into liballegro_hip.so (no host copies in the step loop)."""

    def __init__(self, model, mass_by_mtype: Sequence[float]):
        self.model = model
        self.mass = np.asarray(mass_by_mtype, dtype=np.float64)
        self.stats = None          # bench.py sets this to a dict: per-stage ms and edge counts are summed over the calls of a step

    def _collect(self) -> None:
        # stats = None: nothing is read back.  stats = {} with defer_stats: the library keeps recording its stage events and the caller
        # reads the sums once (model.timings_and_counts()) -- per-call collection waits for the call's kernels and for its edge count,
        # i.e. it synchronises host and device at every force evaluation (fine for tests, wrong inside a timed loop).
        if self.stats is None or getattr(self, "defer_stats", False):
            return
        for k, v in self.model.timings().items():
            self.stats[k] = self.stats.get(k, 0.0) + v
        self.stats["edges"] = self.stats.get("edges", 0) + self.model.nedges()
        self.stats["calls"] = self.stats.get("calls", 0) + 1

    def build_neighbors(self, x: torch.Tensor, nlocal: int, lo, hi, rc_list: float) -> None:
        self.model.build_neighbors_dev(nlocal, x.shape[0], x.data_ptr(), lo, hi, rc_list)

    def compute(self, x, mtype, f, nlocal: int, engvir: torch.Tensor, eatom: Optional[torch.Tensor] = None) -> None:
        nall = x.shape[0]
        self.model.compute_dev(nlocal, nall - nlocal, x.data_ptr(), mtype.data_ptr(), f.data_ptr(),
                               eatom.data_ptr() if eatom is not None else 0, engvir.data_ptr())
        self._collect()

    def compute_range(self, c0: int, c1: int, x, mtype, f, nlocal: int, engvir: torch.Tensor) -> None:
        """Centres ilist[c0:c1) only (ahip_compute_dev_range): lets the ghost exchange run beside the interior centres."""
        nall = x.shape[0]
        self.model.compute_dev_range(c0, c1, nlocal, nall - nlocal, x.data_ptr(), mtype.data_ptr(), f.data_ptr(), 0, engvir.data_ptr())
        self._collect()

    def nve(self, mode: int, n: int, x, v, f, mtype, dt: float) -> None:
        self.model.nve_dev(mode, n, x.data_ptr(), v.data_ptr(), f.data_ptr(), mtype.data_ptr(), self.mass, dt, FTM2V)

    def nve_first(self, n: int, x, v, f, mtype, dt: float) -> None:
        """first half step of the owned atoms + zero-fill of ALL rows of f, in one launch (ahip_nve_first_dev)"""
        self.model.nve_first_dev(n, f.shape[0], x.data_ptr(), v.data_ptr(), f.data_ptr(), mtype.data_ptr(), self.mass, dt, FTM2V)

    def fill_zero(self, t: torch.Tensor, stream: int = 0) -> None:
        self.model.L.check(self.model.L.lib.ahip_fill_zero_dev(t.data_ptr(), t.numel() * t.element_size(), stream or None))

    def borders_local(self, nlocal: int, x, mtype, lo, hi, box, rc: float, capacity: int, xg, mtg, src, shift) -> int:
        """ghosts of a single rank = periodic images of its own atoms inside the halo, built by the library (ahip_borders_local_dev)"""
        return self.model.borders_local_dev(nlocal, x.data_ptr(), mtype.data_ptr(), lo, hi, box, rc, capacity, xg.data_ptr(), mtg.data_ptr(),
                                            src.data_ptr(), shift.data_ptr())

    def reneighbor_flag(self, n: int, x, x_hold, v, dt: float, half_skin: float, flag) -> None:
        """flag[0] = max displacement since the last build + 2 dt max|v| > half_skin (ahip_reneighbor_flag_dev)."""
        self.model.reneighbor_flag_dev(n, x.data_ptr(), x_hold.data_ptr() if n else 0, v.data_ptr(), dt, half_skin, flag.data_ptr())


@dataclass
class _Swap:
    """One directed ghost exchange of one dimension (LAMMPS `swap`)."""
    dim: int
    sendrank: int            # destination of my slab
    recvrank: int            # source of the ghosts I receive
    shift: float             # added to coordinate `dim` of what I SEND (periodic wrap), 0 otherwise
    send_idx: Optional[torch.Tensor] = None
    nsend: int = 0
    nrecv: int = 0
    first_recv: int = 0


class Simulation:
    def __init__(self, backend, box: Sequence[float], r_max: float, skin: float, x_global: np.ndarray,
                 mtype_global: np.ndarray, v_global: Optional[np.ndarray], device: torch.device,
                 grid: Tuple[int, int, int] = (1, 1, 1), rank: int = 0, dist=None, dt: float = 0.001, overlap: Optional[bool] = None,
                 neigh_every: int = 1, neigh_delay: int = 0, neigh_check: bool = True):
        self.backend = backend
        # Overlapped schedule: local atoms are ordered interior-first at every re-neighboring; the first half of
        # the interior centres is evaluated while ghost positions travel (forward comm), then the boundary centres, and the
        # second half of the interior while the ghost forces travel back (reverse comm).  On a GPU the exchange runs on a
        # second (non-blocking) stream; on CPU tensors (gloo tests) the same schedule runs in program order.
        # Default: overlapped when there is an exchange between ranks to hide; one rank evaluates all centres in one call (three
        # launches instead of one cost 2.5 % at 1 M atoms on one GPU: profiles/r02_a_overlap_1gpu.md).
        nranks = int(grid[0]) * int(grid[1]) * int(grid[2])
        self.overlap = (nranks > 1 if overlap is None else bool(overlap)) and hasattr(backend, "compute_range")
        self.comm_stream = torch.cuda.Stream(device) if (self.overlap and device.type == "cuda") else None
        if self.comm_stream is not None and nranks > 1 and hasattr(backend, "model"):
            # the fused kernels are persistent and fill every CU: leave a few workgroup slots free, otherwise the pack / unpack
            # and RCCL kernels of the exchange stream could only start when the force kernel ends
            backend.model.set_option("reserve_wgs", 8)
        # LAMMPS `neigh_modify every N delay M check yes|no` (the reference decks use the defaults `every 1 delay 0 check yes`
        # via `neighbor 1.0 bin`, tests/test_python_repro_allegro.py:84-120): a rebuild is considered only on steps that are a
        # multiple of `every` and at least `delay` steps after the last build; with `check` it happens only if some atom moved
        # more than skin/2 since then, without it unconditionally.
        if neigh_every < 1 or neigh_delay < 0:
            raise ValueError("neigh_modify: every must be >= 1 and delay >= 0")
        self.neigh_every, self.neigh_delay, self.neigh_check = int(neigh_every), int(neigh_delay), bool(neigh_check)
        self.ago = 0                 # steps since the last build (LAMMPS neighbor->ago)
        self.n_int = 0
        self.n_half = 0
        self._flag_host = None
        self._flag_event = None
        self.box = np.asarray(box, dtype=np.float64)
        self.rc = float(r_max) + float(skin)
        self.skin = float(skin)
        self.dev = device
        self.grid = tuple(int(g) for g in grid)
        self.rank = rank
        self.dist = dist
        self.dt = dt
        self.nranks = self.grid[0] * self.grid[1] * self.grid[2]
        self.coord = (rank // (self.grid[1] * self.grid[2]), (rank // self.grid[2]) % self.grid[1], rank % self.grid[2])
        self.lo = np.array([self.box[d] * self.coord[d] / self.grid[d] for d in range(3)])
        self.hi = np.array([self.box[d] * (self.coord[d] + 1) / self.grid[d] for d in range(3)])
        for d in range(3):
            if self.hi[d] - self.lo[d] < self.rc:
                raise ValueError("sub-domain thinner than r_max+skin: use fewer ranks along that dimension")
        xg = np.asarray(x_global, dtype=np.float64)
        own = np.all((xg >= self.lo) & (xg < self.hi), axis=1)
        idx = np.flatnonzero(own)
        self.natoms_global = len(xg)
        self.nlocal = len(idx)
        self.x = torch.tensor(xg[idx], dtype=torch.float64, device=device)
        self.tag = torch.tensor(idx, dtype=torch.int64, device=device)
        self.mtype = torch.tensor(np.asarray(mtype_global)[idx], dtype=torch.int32, device=device)
        v0 = np.zeros((self.nlocal, 3)) if v_global is None else np.asarray(v_global, dtype=np.float64)[idx]
        self.v = torch.tensor(v0, dtype=torch.float64, device=device)
        self.f = torch.zeros((self.nlocal, 3), dtype=torch.float64, device=device)
        self.engvir = torch.zeros(7, dtype=torch.float64, device=device)
        self.engvir3 = torch.zeros((3, 7), dtype=torch.float64, device=device)
        self.swaps: List[_Swap] = []
        self.nrebuild = 0
        self.x_hold = None
        self._ghost_src = self._ghost_shift = None
        self._flag_posted = False
        self.comm = self._make_comm()
        self.rebuild()

    # ---- the library's ghost exchange (csrc/comm.hip) -----------------------------------------------
    def _comm_timed(self, fn, ptr: int) -> None:
        """One library exchange on the current stream; with `time_comm` set (bench.py) bracketed by events on that stream -- read by
        comm_ms() after the timed region, never inside it."""
        if not getattr(self, "time_comm", False) or self.dev.type != "cuda":
            fn(ptr, self._stream())
            return
        st = torch.cuda.current_stream(self.dev)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st)
        fn(ptr, self._stream())
        b.record(st)
        self._comm_events = getattr(self, "_comm_events", [])
        self._comm_events.append((a, b))

    def comm_ms(self) -> float:
        """device time (ms) of the exchanges recorded since the last call (stream time of pack / transport / unpack, both directions)."""
        ev, self._comm_events = getattr(self, "_comm_events", []), []
        tot = 0.0
        for a, b in ev:
            b.synchronize()
            tot += a.elapsed_time(b)
        return tot

    def _stream(self) -> int:
        return torch.cuda.current_stream(self.dev).cuda_stream if self.dev.type == "cuda" else 0

    def _tensor_at(self, ptr: int, nbytes: int, dtype=torch.uint8) -> torch.Tensor:
        """A tensor over raw memory the library handed to the hosted transfer callback (host memory in the CPU emulation, device
        memory on the GPU)."""
        esz = torch.empty((), dtype=dtype).element_size()
        if self.dev.type == "cuda":
            class _Mem:
                pass
            m = _Mem()
            m.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (int(ptr), False), "version": 2}
            t = torch.as_tensor(m, device=self.dev)
        else:
            import ctypes
            t = torch.frombuffer((ctypes.c_char * nbytes).from_address(int(ptr)), dtype=torch.uint8)
        return t.view(dtype) if esz != 1 else t

    def _hosted_xfer(self, ops) -> None:
        """Moves one group of packed buffers with the process group handed to the constructor (gloo on CPU tensors; HostStagedDist
        for ranks that share one GPU)."""
        d = self.dist
        p2p = []
        for kind, peer, ptr, nbytes in ops:
            if kind == 0:
                p2p.append(d.P2POp(d.isend, self._tensor_at(ptr, nbytes), peer))
            elif kind == 1:
                p2p.append(d.P2POp(d.irecv, self._tensor_at(ptr, nbytes), peer))
            elif kind == 2:
                d.all_reduce(self._tensor_at(ptr, nbytes, torch.float64))
            else:
                d.all_reduce(self._tensor_at(ptr, nbytes, torch.int32), op=d.ReduceOp.MAX)
        if p2p:
            for w in d.batch_isend_irecv(p2p):
                w.wait()

    def _make_comm(self):
        """The exchange runs inside the library when the backend is the library (HipBackend): RCCL when the process group is RCCL
        ("nccl"), the hosted transport otherwise.  Other backends (none today) keep the torch implementation below."""
        model = getattr(self.backend, "model", None)
        if model is None or not hasattr(model.L.lib, "ahip_comm_forward"):
            return None
        from . import capi
        if self.nranks == 1:
            return capi.Comm(model.L, 0, 1)
        is_rccl = self.dev.type == "cuda" and hasattr(self.dist, "get_backend") and self.dist.get_backend() == "nccl"
        if os.environ.get("AHIP_COMM", "") == "hosted":        # A/B and first-contact debugging: force the host-staged transport over the process group
            is_rccl = False
        self.comm_info = {"transport": "hosted", "rccl_version": 0, "init_ms": 0.0}
        if is_rccl:
            t_init = time.perf_counter()
            # the library's own RCCL communicator: rank 0's ncclUniqueId travels through the process group.  Every rank reports whether
            # its communicator came up; if any did not (librccl.so missing, init failure) ALL ranks fall back to the hosted transport over the
            # process group, so a first-contact problem costs speed, not the run.
            idt = torch.zeros(129, dtype=torch.uint8, device=self.dev)
            if self.rank == 0:
                try:
                    idt[:128].copy_(torch.frombuffer(bytearray(capi.Comm.unique_id(model.L)), dtype=torch.uint8))
                    idt[128] = 1
                except capi.AhipError as e:
                    print(f"[md] ahip_comm_unique_id failed ({e}); ghost exchange through torch.distributed", flush=True)
            self.dist.broadcast(idt, src=0)
            comm, ok = None, torch.zeros(1, dtype=torch.int32, device=self.dev)
            if int(idt[128].item()) == 1:
                try:
                    comm = capi.Comm(model.L, self.rank, self.nranks, rccl_id=bytes(idt[:128].cpu().numpy().tobytes()), device=self.dev.index or 0)
                    ok[0] = 1
                except capi.AhipError as e:
                    print(f"[md] rank {self.rank}: ahip_comm_create_rccl failed ({e})", flush=True)
            self.dist.all_reduce(ok, op=self.dist.ReduceOp.MIN)
            if int(ok.item()) == 1:
                self.comm_info = {"transport": "rccl", "rccl_version": int(model.L.lib.ahip_comm_rccl_version()),
                                  "init_ms": round(1e3 * (time.perf_counter() - t_init), 1)}
                return comm
            if comm is not None:
                comm.close()
        return capi.Comm(model.L, self.rank, self.nranks, xfer=self._hosted_xfer)

    def _set_comm_plan(self) -> None:
        if self.comm is None or getattr(self, "_lib_plan", False):      # (ahip_comm_borders has installed its own plan)
            return
        # the single-rank plan (image chains resolved locally) only on ONE rank: with several ranks a brick that received no ghosts may still
        # have slabs to send (a cluster or slab next to an empty neighbour brick) and its peer posts the matching receive
        if self.nranks == 1 and (self._ghost_src is not None or self.nall == self.nlocal):
            ng = self.nall - self.nlocal
            self.comm.set_plan_local(self.nlocal, ng, self._ghost_src.data_ptr() if ng else 0, self._ghost_shift.data_ptr() if ng else 0)
            return
        sw = self.swaps
        self.comm.set_plan([s.dim for s in sw], [s.sendrank for s in sw], [s.recvrank for s in sw], [s.shift for s in sw],
                           [s.nsend for s in sw], [s.nrecv for s in sw], [s.first_recv for s in sw],
                           [s.send_idx.data_ptr() if s.nsend else 0 for s in sw])

    # ---- rank helpers ---------------------------------------------------------------------------
    def _rank_of(self, c) -> int:
        return (c[0] * self.grid[1] + c[1]) * self.grid[2] + c[2]

    def _neighbor(self, dim: int, step: int) -> Tuple[int, float]:
        """rank of my neighbour in direction `step` along `dim`, and the shift applied to atoms sent there."""
        c = list(self.coord)
        c[dim] += step
        shift = 0.0
        if c[dim] < 0:
            c[dim] += self.grid[dim]; shift = +self.box[dim]
        elif c[dim] >= self.grid[dim]:
            c[dim] -= self.grid[dim]; shift = -self.box[dim]
        return self._rank_of(c), shift

    def _sendrecv(self, send: torch.Tensor, sendrank: int, recvrank: int, nrecv: int) -> torch.Tensor:
        """Exchange with neighbours; self-exchange is a local copy."""
        if sendrank == self.rank and recvrank == self.rank:
            return send
        recv = torch.empty((nrecv,) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
        ops = []
        if send.shape[0] > 0:
            ops.append(self.dist.P2POp(self.dist.isend, send.contiguous(), sendrank))
        if nrecv > 0:
            ops.append(self.dist.P2POp(self.dist.irecv, recv, recvrank))
        if ops:
            for w in self.dist.batch_isend_irecv(ops):
                w.wait()
        return recv

    def _sendrecv_pair(self, sends, swaps, reverse: bool = False):
        """Both directed exchanges of one dimension in ONE batch_isend_irecv (one RCCL group call instead of two).
        sends[k] goes with swaps[k]; forward: to sendrank, from recvrank (nrecv rows); reverse: the other way (nsend rows)."""
        out, ops = [None, None], []
        for k, (buf, sw) in enumerate(zip(sends, swaps)):
            to, frm, n = (sw.recvrank, sw.sendrank, sw.nsend) if reverse else (sw.sendrank, sw.recvrank, sw.nrecv)
            if to == self.rank and frm == self.rank:
                out[k] = buf
                continue
            out[k] = torch.empty((n,) + tuple(buf.shape[1:]), dtype=buf.dtype, device=buf.device)
            if buf.shape[0] > 0:
                ops.append(self.dist.P2POp(self.dist.isend, buf.contiguous(), to, tag=k))
            if n > 0:
                ops.append(self.dist.P2POp(self.dist.irecv, out[k], frm, tag=k))
        if ops:
            for w in self.dist.batch_isend_irecv(ops):
                w.wait()
        return out

    def _exchange_counts(self, n: int, sendrank: int, recvrank: int) -> int:
        if sendrank == self.rank and recvrank == self.rank:
            return n
        s = torch.tensor([n], dtype=torch.int64, device=self.dev)
        r = torch.zeros(1, dtype=torch.int64, device=self.dev)
        for w in self.dist.batch_isend_irecv([self.dist.P2POp(self.dist.isend, s, sendrank),
                                              self.dist.P2POp(self.dist.irecv, r, recvrank)]):
            w.wait()
        return int(r.item())

    # ---- re-neighboring: wrap, migrate, borders, neighbor build -----------------------------------
    def _lib_reneighbor(self) -> bool:
        """Several ranks with the library's communicator: Comm::exchange / Comm::borders run inside the library (csrc/comm.hip, round 6) instead of the torch swap
        chain below.  AHIP_LIB_BORDERS=0 keeps the torch path (A/B, and the reference the tests compare the library path with)."""
        return (self.nranks > 1 and self.comm is not None and hasattr(self.comm, "borders") and hasattr(self.comm.L.lib, "ahip_comm_borders")
                and os.environ.get("AHIP_LIB_BORDERS", "1") != "0")

    def _migrate_lib(self) -> None:
        n = self.nlocal
        cap = n + max(4096, n // 4)
        x = torch.empty((cap, 3), dtype=torch.float64, device=self.dev); x[:n] = self.x[:n]
        v = torch.empty((cap, 3), dtype=torch.float64, device=self.dev); v[:n] = self.v[:n]
        tag = torch.empty(cap, dtype=torch.int64, device=self.dev); tag[:n] = self.tag[:n]
        mt = torch.empty(cap, dtype=torch.int32, device=self.dev); mt[:n] = self.mtype[:n]
        nn = self.comm.migrate(n, x.data_ptr(), v.data_ptr(), tag.data_ptr(), mt.data_ptr(), cap, self.box, self.grid, self.coord, self._stream())
        if nn < 0:
            raise RuntimeError(f"migration: a brick would hold {-nn} atoms, more than the {cap} rows provided (rank {self.rank})")
        self.nlocal = nn
        self.x, self.v, self.tag, self.mtype = x[:nn], v[:nn], tag[:nn], mt[:nn]

    def _borders_lib(self) -> None:
        nl = self.nlocal
        brick = np.maximum(np.asarray(self.hi) - np.asarray(self.lo), 1e-9)
        est = int(nl * (np.prod(1.0 + 2.0 * self.rc / brick) - 1.0) * 1.3) + 2048
        cap = nl + max(est, int(1.25 * getattr(self, "_nghost_last", 0)))
        if os.environ.get("AHIP_TEST_SMALL_BORDERS_CAP") == "1":      # tests only: start too small, so that the collective overflow answer and the retry are exercised
            cap = nl + 8
        while True:
            xa = torch.empty((cap, 3), dtype=torch.float64, device=self.dev); xa[:nl] = self.x[:nl]
            mta = torch.empty(cap, dtype=torch.int32, device=self.dev); mta[:nl] = self.mtype[:nl]
            nall = self.comm.borders(nl, xa.data_ptr(), mta.data_ptr(), cap, self.lo, self.hi, self.box, self.rc, self.grid, self.coord, self._stream())
            if nall <= cap:
                break
            cap = int(1.1 * nall) + 1024
        self._nghost_last = nall - nl
        self.swaps = []
        self._lib_plan = True
        self.x, self.mtype = xa[:nall], mta[:nall]
        self.nall = nall
        self._ghost_src = self._ghost_shift = None

    def _migrate(self) -> None:
        if self._lib_reneighbor():
            return self._migrate_lib()
        n = self.nlocal
        x, v, tag, mt = self.x[:n], self.v[:n], self.tag[:n], self.mtype[:n]
        if getattr(self, "_box_t", None) is None:
            self._box_t = torch.tensor(self.box, dtype=torch.float64, device=self.dev)       # once: a host -> device copy per re-neighboring otherwise
        box = self._box_t
        x = x - torch.floor(x / box) * box                      # periodic wrap into the global box
        for d in range(3):
            if self.grid[d] == 1:
                continue
            g = self.grid[d]
            cell = torch.clamp(torch.floor(x[:, d] / (self.box[d] / g)), 0, g - 1).to(torch.int64)
            delta = (cell - self.coord[d]) % g                  # atoms move at most one brick between rebuilds
            below = delta == (g - 1)
            above = (delta == 1) & ~below                       # g == 2: both directions reach the same rank
            keep = ~(below | above)
            pieces = []
            for mask, step in ((below, -1), (above, +1)):
                dest, _ = self._neighbor(d, step)
                src, _ = self._neighbor(d, -step)
                pack = torch.cat([x[mask], v[mask], tag[mask].to(torch.float64).unsqueeze(1),
                                  mt[mask].to(torch.float64).unsqueeze(1)], dim=1)
                nrecv = self._exchange_counts(pack.shape[0], dest, src)
                pieces.append(self._sendrecv(pack, dest, src, nrecv))
            got = torch.cat(pieces, dim=0)
            x = torch.cat([x[keep], got[:, 0:3]]); v = torch.cat([v[keep], got[:, 3:6]])
            tag = torch.cat([tag[keep], got[:, 6].to(torch.int64)]); mt = torch.cat([mt[keep], got[:, 7].to(torch.int32)])
        self.nlocal = x.shape[0]
        self.x, self.v, self.tag, self.mtype = x.contiguous(), v.contiguous(), tag.contiguous(), mt.contiguous()

    def _borders_local(self) -> bool:
        """One rank, library backend: the ghosts (periodic images of the rank's own atoms) come from ONE library call -- count, scan, fill -- instead of six
        rounds of torch masking / indexing (0.8 of the 1.9 ms a re-neighboring of 10 648 atoms cost, every 16 steps at 300 K).  The per-step exchange of a
        single rank only needs the image -> source map and the shifts (ahip_comm_set_plan_local), which the call returns; no swap list is built."""
        if self.nranks != 1 or not hasattr(self.backend, "borders_local") or not hasattr(self.backend.model.L.lib, "ahip_borders_local_dev"):
            return False
        if min(self.box) < self.rc:
            return False
        nl = self.nlocal
        cap = max(1024, int(1.25 * getattr(self, "_nghost_last", 0)) or int(nl * (np.prod(1.0 + 2.0 * self.rc / np.asarray(self.box)) - 1.0) * 1.25) + 1024)
        while True:                                            # locals and images in one allocation: the call writes the images behind the locals
            xa = torch.empty((nl + cap, 3), dtype=torch.float64, device=self.dev)
            mta = torch.empty(nl + cap, dtype=torch.int32, device=self.dev)
            src = torch.empty(cap, dtype=torch.long, device=self.dev)
            shv = torch.empty((cap, 3), dtype=torch.float64, device=self.dev)
            ng = self.backend.borders_local(nl, self.x, self.mtype, self.lo, self.hi, np.asarray(self.box, dtype=np.float64), self.rc, cap,
                                            xa[nl:], mta[nl:], src, shv)
            if ng <= cap:
                break
            cap = int(1.1 * ng) + 64
        self._nghost_last = ng
        self.swaps = []
        xa[:nl] = self.x[:nl]
        mta[:nl] = self.mtype[:nl]
        self.x, self.mtype = xa[: nl + ng], mta[: nl + ng]
        self.nall = nl + ng
        self._ghost_src, self._ghost_shift = (src[:ng], shv[:ng]) if ng else (None, None)
        return True

    def _borders(self) -> None:
        self._lib_plan = False
        if self._borders_local():
            return
        if self._lib_reneighbor():
            return self._borders_lib()
        self.swaps = []
        x, mt = self.x, self.mtype
        for d in range(3):
            nprev = x.shape[0]
            new_x, new_mt = [], []
            for step in (-1, +1):
                sendrank, shift = self._neighbor(d, step)
                recvrank, _ = self._neighbor(d, -step)
                sw = _Swap(dim=d, sendrank=sendrank, recvrank=recvrank, shift=shift)
                xs = x[:nprev, d]                                  # everything known before this dimension
                mask = (xs < self.lo[d] + self.rc) if step < 0 else (xs >= self.hi[d] - self.rc)
                sw.send_idx = torch.nonzero(mask, as_tuple=False).squeeze(1)
                sw.nsend = int(sw.send_idx.shape[0])
                if sendrank == self.rank and recvrank == self.rank:      # one rank along this dimension: the ghosts are my own images
                    sw.nrecv = sw.nsend
                    rx = x[sw.send_idx]                                  # (advanced indexing: a copy)
                    if shift != 0.0:
                        rx[:, d] += shift
                    rmt = mt[sw.send_idx]
                else:
                    sw.nrecv = self._exchange_counts(sw.nsend, sendrank, recvrank)
                    buf = x[sw.send_idx].clone()
                    buf[:, d] += shift
                    rx = self._sendrecv(buf, sendrank, recvrank, sw.nrecv)
                    rmt = self._sendrecv(mt[sw.send_idx].to(torch.float64).unsqueeze(1), sendrank, recvrank, sw.nrecv).squeeze(1).to(torch.int32)
                sw.first_recv = nprev + sum(t.shape[0] for t in new_x)
                new_x.append(rx); new_mt.append(rmt)
                self.swaps.append(sw)
            x = torch.cat([x] + new_x); mt = torch.cat([mt] + new_mt)
        self.x, self.mtype = x.contiguous(), mt.contiguous()
        self.nall = self.x.shape[0]
        # One rank: every ghost is a periodic image of a LOCAL atom.  Resolving the per-dimension chains (a ghost of a ghost ...) once
        # per re-neighboring turns the six swaps of the per-step exchanges into one gather (positions) and one index_add (forces).
        self._ghost_src = self._ghost_shift = None
        if self.nranks == 1 and self.nall > self.nlocal:
            nl = self.nlocal
            src = torch.empty(self.nall - nl, dtype=torch.long, device=self.dev)
            shv = torch.zeros((self.nall - nl, 3), dtype=torch.float64, device=self.dev)
            for sw in self.swaps:                                  # in creation order: sources are locals or earlier ghosts
                idx = sw.send_idx
                isg = idx >= nl
                g = (idx - nl).clamp(min=0)
                s_fin = torch.where(isg, src[g], idx)
                sh = torch.where(isg.unsqueeze(1), shv[g], torch.zeros((), dtype=torch.float64, device=self.dev))
                sh = sh.clone(); sh[:, sw.dim] += sw.shift
                a = sw.first_recv - nl
                src[a: a + sw.nrecv] = s_fin
                shv[a: a + sw.nrecv] = sh
            self._ghost_src, self._ghost_shift = src, shv

    def _order_interior_first(self) -> None:
        """Local atoms farther than r_max+skin from every brick face come first: every list neighbour of such an atom is a
        local atom, so its edges need no ghost position and produce no ghost force."""
        n = self.nlocal
        lo = torch.tensor(self.lo + self.rc, dtype=torch.float64, device=self.dev)
        hi = torch.tensor(self.hi - self.rc, dtype=torch.float64, device=self.dev)
        x = self.x[:n]
        interior = ((x >= lo) & (x < hi)).all(dim=1)
        order = torch.argsort((~interior).to(torch.int8), stable=True)
        self.x, self.v = self.x[order].contiguous(), self.v[order].contiguous()
        self.tag, self.mtype = self.tag[order].contiguous(), self.mtype[order].contiguous()
        self.n_int = int(interior.sum().item())
        self.n_half = self.n_int // 2

    def set_overlap(self, flag: bool) -> None:
        """Switch between the overlapped three-range schedule and the serial one (exchange, all centres in one call, exchange) at run time.
        The interior-first atom order of the last re-neighboring serves both; switching overlap ON after a re-neighboring done without it
        re-neighbors once.  The persistent kernels keep workgroup slots free for the exchange kernels only while there is something to overlap."""
        flag = bool(flag) and hasattr(self.backend, "compute_range")
        if flag == self.overlap:
            return
        self.overlap = flag
        if flag and self.comm_stream is None and self.dev.type == "cuda":
            self.comm_stream = torch.cuda.Stream(self.dev)
        if self.dev.type == "cuda" and self.nranks > 1 and hasattr(self.backend, "model"):
            self.backend.model.set_option("reserve_wgs", 8 if flag else 0)
        if flag and not getattr(self, "_ordered", False):
            self.rebuild()                               # between two steps: the new force array must hold the forces the next half kick uses
            self._flag_posted = False
            self.compute_forces(comm_first=False)

    def rebuild(self) -> None:
        self._migrate()
        self._ordered = bool(self.overlap)
        if self.overlap:
            self._order_interior_first()
        self._borders()
        self._set_comm_plan()
        self.f = torch.zeros((self.nall, 3), dtype=torch.float64, device=self.dev)
        self._f_zeroed = True
        lo = self.lo - self.rc - 1e-6
        hi = self.hi + self.rc + 1e-6
        self.backend.build_neighbors(self.x, self.nlocal, lo, hi, self.rc)
        self.x_hold = self.x[: self.nlocal].clone()
        self.nrebuild += 1
        self.ago = 0

    # ---- per-step communication -------------------------------------------------------------------
    def forward_comm(self) -> None:
        if self.comm is not None:                                  # HIP pack kernels + RCCL groups inside the library
            self._comm_timed(self.comm.forward, self.x.data_ptr())
            return
        if self._ghost_src is not None:                            # one rank: all ghosts are images of local atoms
            self.x[self.nlocal:] = self.x[self._ghost_src] + self._ghost_shift
            return
        for k in range(0, len(self.swaps), 2):
            pair = self.swaps[k: k + 2]
            bufs = []
            for sw in pair:
                buf = self.x[sw.send_idx]
                if sw.shift != 0.0:
                    buf[:, sw.dim] += sw.shift                    # advanced indexing made a copy
                bufs.append(buf)
            for sw, rx in zip(pair, self._sendrecv_pair(bufs, pair)):
                self.x[sw.first_recv: sw.first_recv + sw.nrecv] = rx

    def reverse_comm(self) -> None:
        if self.comm is not None:
            self._comm_timed(self.comm.reverse, self.f.data_ptr())
            return
        if self._ghost_src is not None:
            self.f[: self.nlocal].index_add_(0, self._ghost_src, self.f[self.nlocal:])      # sources are local rows: disjoint from the ghost rows read
            return
        for k in range(len(self.swaps) - 2, -1, -2):
            pair = self.swaps[k: k + 2]
            # both directions of a dimension read ghost rows received in that dimension and add into rows known before it
            bufs = [self.f[sw.first_recv: sw.first_recv + sw.nrecv] for sw in pair]
            got = self._sendrecv_pair(bufs, pair, reverse=True)
            for sw, rx in zip(pair, got):
                if sw.recvrank == self.rank and sw.sendrank == self.rank:
                    rx = rx.clone()                              # self-exchange returns a view of f
                self.f.index_add_(0, sw.send_idx, rx)

    def _post_rebuild_flag(self) -> None:
        """max displacement since the last build > skin/2 (any rank)?  Evaluated on the device, all-reduced there, copied to
        pinned host memory asynchronously and READ ONE STEP LATER, so no step waits on a device->host round trip; the
        criterion therefore includes the motion of the step in between (2 dt max|v|)."""
        if hasattr(self.backend, "reneighbor_flag"):
            # one library call (two launches) instead of a dozen elementwise / reduction kernels per step
            flag = self._flag_dev if getattr(self, "_flag_dev", None) is not None else torch.zeros(1, dtype=torch.int32, device=self.dev)
            self._flag_dev = flag
            self.backend.reneighbor_flag(self.nlocal, self.x, self.x_hold, self.v, self.dt, 0.5 * self.skin, flag)
        else:
            if self.nlocal:
                d = self.x[: self.nlocal] - self.x_hold
                v = self.v[: self.nlocal]
                # displacement now + twice the fastest atom's next step (the flag is acted on one step late)
                reach = (d * d).sum(dim=1).max().sqrt() + 2.0 * self.dt * (v * v).sum(dim=1).max().sqrt()
            else:
                reach = torch.zeros((), dtype=torch.float64, device=self.dev)
            flag = (reach > 0.5 * self.skin).to(torch.int32).reshape(1)
        if self.nranks > 1:
            if self.comm is not None:
                self.comm.allreduce(flag.data_ptr(), 1, 1, self._stream())
            else:
                self.dist.all_reduce(flag, op=self.dist.ReduceOp.MAX)
        if self.dev.type == "cuda":
            if self._flag_host is None:
                self._flag_host = torch.zeros(1, dtype=torch.int32).pin_memory()
                self._flag_event = torch.cuda.Event()
            self._flag_host.copy_(flag, non_blocking=True)
            self._flag_event.record()
        else:
            self._flag_host = flag.clone()

    def needs_rebuild(self) -> bool:
        """`neigh_modify every / delay / check`, with the displacement flag posted during the previous step."""
        self.ago += 1
        if self.ago < self.neigh_delay or self.ago % self.neigh_every:
            return False
        if not self.neigh_check:
            return True
        if self._flag_host is None or not self._flag_posted:
            return False
        if self._flag_event is not None:
            self._flag_event.synchronize()           # recorded a whole force evaluation ago: already complete
        return bool(int(self._flag_host[0]))

    # ---- force evaluation and time step -----------------------------------------------------------
    def compute_forces(self, comm_first: bool = True) -> None:
        """forward comm -> forces of all centres -> reverse comm.  Overlapped schedule: see __init__."""
        if getattr(self, "_f_zeroed", False):           # the integrator's first half step, or the re-neighboring's fresh array, left it zero
            self._f_zeroed = False
        elif hasattr(self.backend, "fill_zero"):
            self.backend.fill_zero(self.f, self._stream())
        else:
            self.f.zero_()
        if not self.overlap:
            if comm_first:
                self.forward_comm()
            self.backend.compute(self.x, self.mtype, self.f, self.nlocal, self.engvir)
            self.reverse_comm()
            return
        cs = self.comm_stream
        n0, n1, nl = self.n_half, self.n_int, self.nlocal
        ev = self.engvir3
        ev.zero_()
        if cs is not None:
            cur = torch.cuda.current_stream(self.dev)
            e0 = torch.cuda.Event(); e0.record(cur)
            with torch.cuda.stream(cs):
                cs.wait_event(e0)                    # positions integrated, forces zeroed
                if comm_first:
                    self.forward_comm()
                e1 = torch.cuda.Event(); e1.record(cs)
            self.backend.compute_range(0, n0, self.x, self.mtype, self.f, nl, ev[0])          # interior, first half
            cur.wait_event(e1)
            self.backend.compute_range(n1, nl, self.x, self.mtype, self.f, nl, ev[1])         # boundary: needs ghost positions
            e2 = torch.cuda.Event(); e2.record(cur)
            with torch.cuda.stream(cs):
                cs.wait_event(e2)
                self.reverse_comm()                  # ghost forces exist once the boundary centres are done
                e3 = torch.cuda.Event(); e3.record(cs)
            self.backend.compute_range(n0, n1, self.x, self.mtype, self.f, nl, ev[2])         # interior, second half
            cur.wait_event(e3)
        else:
            if comm_first:
                self.forward_comm()
            self.backend.compute_range(0, n0, self.x, self.mtype, self.f, nl, ev[0])
            self.backend.compute_range(n1, nl, self.x, self.mtype, self.f, nl, ev[1])
            self.reverse_comm()
            self.backend.compute_range(n0, n1, self.x, self.mtype, self.f, nl, ev[2])
        torch.sum(ev, dim=0, out=self.engvir)

    def setup(self) -> None:
        self.compute_forces()
        self._flag_posted = False

    def step(self) -> None:
        n = self.nlocal
        if hasattr(self.backend, "nve_first") and hasattr(self.backend.model.L.lib, "ahip_nve_first_dev"):
            self.backend.nve_first(n, self.x, self.v, self.f, self.mtype, self.dt)      # v += dt/2 f/m ; x += dt v ; f = 0 for the evaluation below
            self._f_zeroed = True
        else:
            self.backend.nve(0, n, self.x, self.v, self.f, self.mtype, self.dt)         # v += dt/2 f/m ; x += dt v
        if self.needs_rebuild():
            self.rebuild()
            self._flag_posted = False
            self.compute_forces(comm_first=False)    # borders() just placed fresh ghost positions
        else:
            self._post_rebuild_flag()
            self._flag_posted = True
            self.compute_forces()
        self.backend.nve(1, self.nlocal, self.x, self.v, self.f, self.mtype, self.dt)   # v += dt/2 f/m

    # ---- observables (reduced over ranks) ----------------------------------------------------------
    def thermo(self, mass_by_mtype: Sequence[float]) -> dict:
        mass = torch.tensor(np.asarray(mass_by_mtype), dtype=torch.float64, device=self.dev)[self.mtype[: self.nlocal].long()]
        ke = 0.5 * MVV2E * (mass.unsqueeze(1) * self.v[: self.nlocal] ** 2).sum()
        t = torch.cat([self.engvir.clone(), ke.reshape(1)])
        if self.nranks > 1:
            self.dist.all_reduce(t)
        return dict(pe=float(t[0]), virial=t[1:7].tolist(), ke=float(t[7]))

    def gather_forces(self) -> np.ndarray:
        """forces by global atom id (tests)."""
        out = torch.zeros((self.natoms_global, 3), dtype=torch.float64, device=self.dev)
        out[self.tag[: self.nlocal]] = self.f[: self.nlocal]
        if self.nranks > 1:
            self.dist.all_reduce(out)
        return out.cpu().numpy()


class HostStagedDist:
    """Debug transport: the subset of `torch.distributed` that :class:`Simulation` uses, for CUDA tensors over a process group
    that only moves CPU tensors (gloo).  Every message is staged through host memory.  It exists so that the multi-rank GPU code
    path (interior / boundary split on HIP streams, ghost exchange between the range evaluations, migration) can be exercised
    with several processes sharing ONE GPU, where RCCL refuses to form a communicator; production runs pass `torch.distributed`
    itself (backend nccl = RCCL over xGMI)."""

    class _Done:
        def __init__(self, works, backs):
            self.works, self.backs = works, backs

        def wait(self):
            for w in self.works:
                w.wait()
            for t, c in self.backs:
                t.copy_(c)
            self.works, self.backs = [], []

    def __init__(self, dist):
        self.d = dist
        self.ReduceOp = dist.ReduceOp
        self.isend, self.irecv = dist.isend, dist.irecv

    def P2POp(self, op, tensor, peer, tag=0):
        return (op, tensor, peer, tag)

    def batch_isend_irecv(self, ops):
        real, backs = [], []
        for op, t, peer, tag in ops:
            if op is self.d.isend:
                real.append(self.d.P2POp(self.d.isend, t.detach().cpu().contiguous(), peer, tag=tag))
            else:
                c = torch.empty(tuple(t.shape), dtype=t.dtype)
                backs.append((t, c))
                real.append(self.d.P2POp(self.d.irecv, c, peer, tag=tag))
        return [HostStagedDist._Done(self.d.batch_isend_irecv(real), backs)]

    def all_reduce(self, t, op=None):
        c = t.detach().cpu()
        self.d.all_reduce(c, op=op if op is not None else self.d.ReduceOp.SUM)
        t.copy_(c)

    def barrier(self):
        self.d.barrier()


def maxwell_boltzmann(n: int, mass: np.ndarray, temperature: float, seed: int) -> np.ndarray:
    rng = np.random.RandomState(seed)
    sigma = np.sqrt(KB * temperature / (mass * MVV2E))
    v = rng.normal(size=(n, 3)) * sigma[:, None]
    v -= (v * mass[:, None]).sum(0) / mass.sum()
    return v
