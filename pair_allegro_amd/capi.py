"""ctypes binding of liballegro_hip.so (include/allegro_hip.h).

The library is the product; this file is the thinnest possible Python view of its C-ABI, used by
the Python host mirror (pair.py), the mini-MD driver (md.py), bench.py and the tests.  It fails
loudly when the shared library is missing -- there is no Python/CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, "liballegro_hip.so")


def harness_pinned_copy_default() -> None:
    """For Python HARNESSES (tests/conftest.py, bench.py, the md workers), not for the library: torch moving large numpy arrays to the device goes
    through the HIP runtime's pin-on-the-fly path for pageable copies above 1 MiB, whose address-keyed cache went stale in long-lived processes
    (round 4: 4 of 10 whole-suite runs died in it).  Raising the runtime's threshold (MiB) sends those copies down its staged path.  The
    library does not depend on this -- every pageable copy of its own is staged through page-locked memory (csrc/engine.h: copy_h2d / copy_d2h) --
    and importing the binding no longer changes process-wide runtime behaviour.  Only effective before the first HIP call."""
    os.environ.setdefault("GPU_PINNED_MIN_XFER_SIZE", "4095")       # the largest value that also survives a 32-bit MiB -> bytes conversion


AHIP_OK, AHIP_ERR_ARG, AHIP_ERR_FILE, AHIP_ERR_DEVICE, AHIP_ERR_STATE, AHIP_ERR_UNSUPPORTED = range(6)

# every symbol declared in include/allegro_hip.h (tests check the .so exports all of them)
SYMBOLS = [
    "ahip_last_error", "ahip_device_count", "ahip_model_load", "ahip_model_free", "ahip_model_meta",
    "ahip_set_option", "ahip_get_timing_counts", "ahip_last_tile_occupancy", "ahip_neigh_update", "ahip_neigh_update_csr", "ahip_neigh_update_dev",
    "ahip_compute", "ahip_compute_dev", "ahip_output_register", "ahip_output_get", "ahip_get_edges", "ahip_debug_dump_edges", "ahip_get_timings",
    "ahip_compute_dev_range", "ahip_last_list_size", "ahip_neigh_update_dev_table", "ahip_map_types_dev", "ahip_reneighbor_flag_dev",
    "ahip_last_path", "ahip_last_max_degree", "ahip_debug_fused_linear", "ahip_debug_fused_edges", "ahip_build_neighbors_dev", "ahip_nve_dev", "ahip_nve_first_dev",
    "ahip_model_allow_tf32", "ahip_comm_unique_id", "ahip_comm_create_rccl", "ahip_comm_create_hosted", "ahip_comm_rccl_version", "ahip_comm_free", "ahip_comm_set_plan", "ahip_comm_set_plan_local",
    "ahip_comm_forward", "ahip_comm_reverse", "ahip_comm_allreduce", "ahip_comm_selftest", "ahip_fill_zero_dev", "ahip_borders_local_dev", "ahip_arith_note", "ahip_comm_borders", "ahip_comm_migrate",
]


class XferOp(C.Structure):
    _fields_ = [("kind", C.c_int), ("peer", C.c_int), ("ptr", C.c_void_p), ("bytes", C.c_longlong)]


XFER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(XferOp))


class AhipError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"[ahip {code}] {msg}")
        self.code = code
        self.msg = msg


def _p(a, ctype):
    return a.ctypes.data_as(C.POINTER(ctype)) if a is not None else None


class Library:
    def __init__(self, path: Optional[str] = None):
        path = path or os.environ.get("ALLEGRO_HIP_LIB") or DEFAULT_LIB
        if not os.path.exists(path):
            raise FileNotFoundError(
                f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  allegro-hip has no CPU fallback.")
        self.path = path
        self.lib = C.CDLL(path)
        L = self.lib
        L.ahip_last_error.restype = C.c_char_p
        L.ahip_last_path.restype = C.c_char_p
        L.ahip_last_path.argtypes = [C.c_void_p]
        L.ahip_arith_note.restype = C.c_char_p
        L.ahip_arith_note.argtypes = [C.c_void_p]
        L.ahip_last_max_degree.argtypes = [C.c_void_p]
        L.ahip_output_register.argtypes = [C.c_void_p, C.c_char_p]
        L.ahip_output_get.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_double), C.c_longlong, C.POINTER(C.c_longlong)]
        L.ahip_device_count.argtypes = [C.POINTER(C.c_int)]
        L.ahip_model_load.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]
        L.ahip_model_free.argtypes = [C.c_void_p]
        L.ahip_model_free.restype = None
        L.ahip_model_meta.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_char_p),
                                      C.POINTER(C.POINTER(C.c_double)), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                      C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_char_p)]
        L.ahip_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
        L.ahip_model_allow_tf32.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        L.ahip_neigh_update.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                        C.POINTER(C.POINTER(C.c_int)), C.c_int]
        L.ahip_neigh_update_csr.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_longlong),
                                            C.POINTER(C.c_int), C.c_int]
        L.ahip_neigh_update_dev.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong]
        L.ahip_compute.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_int,
                                   C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                   C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.ahip_compute_dev.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_double),
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ahip_compute_dev_range.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_double),
                                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ahip_last_list_size.argtypes = [C.c_void_p]
        L.ahip_neigh_update_dev_table.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_longlong,
                                                  C.c_int, C.c_void_p]
        L.ahip_reneighbor_flag_dev.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_void_p, C.c_void_p]
        L.ahip_map_types_dev.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_void_p, C.c_void_p]
        L.ahip_last_list_size.restype = C.c_longlong
        L.ahip_get_edges.argtypes = [C.c_void_p, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.POINTER(C.c_double)]
        L.ahip_debug_dump_edges.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        L.ahip_get_timings.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.POINTER(C.c_double)), C.POINTER(C.c_int)]
        L.ahip_get_timing_counts.argtypes = [C.c_void_p, C.POINTER(C.POINTER(C.c_double)), C.POINTER(C.c_int)]
        L.ahip_build_neighbors_dev.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_double),
                                               C.POINTER(C.c_double), C.c_double, C.c_void_p]
        L.ahip_nve_dev.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.POINTER(C.c_double), C.c_double, C.c_double, C.c_void_p]

        L.ahip_debug_fused_linear.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.ahip_comm_unique_id.argtypes = [C.c_char_p]
        L.ahip_comm_create_rccl.argtypes = [C.c_int, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]
        L.ahip_comm_create_hosted.argtypes = [C.c_int, C.c_int, XFER_FN, C.c_void_p, C.POINTER(C.c_void_p)]
        L.ahip_comm_free.argtypes = [C.c_void_p]
        L.ahip_comm_free.restype = None
        L.ahip_comm_set_plan.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double),
                                         C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_void_p)]
        L.ahip_comm_set_plan_local.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.ahip_comm_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.ahip_comm_reverse.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.ahip_comm_allreduce.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.ahip_comm_migrate.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int),
                                        C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p]
        L.ahip_comm_borders.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                        C.c_double, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p]
        L.ahip_comm_selftest.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.ahip_fill_zero_dev.argtypes = [C.c_void_p, C.c_longlong, C.c_void_p]

    def debug_fused_linear(self, W: np.ndarray, x: np.ndarray) -> np.ndarray:
        W = np.ascontiguousarray(W, dtype=np.float64)
        x = np.ascontiguousarray(x, dtype=np.float32)
        K, N = W.shape
        assert x.shape == (32, K)
        out = np.zeros((32, N), dtype=np.float32)
        self.check(self.lib.ahip_debug_fused_linear(K, N, _p(W, C.c_double), _p(x, C.c_float), _p(out, C.c_float)))
        return out

    def check(self, rc: int) -> None:
        if rc != 0:
            raise AhipError(rc, self.lib.ahip_last_error().decode(errors="replace"))

    def device_count(self) -> int:
        n = C.c_int(0)
        self.check(self.lib.ahip_device_count(C.byref(n)))
        return n.value


_default: Optional[Library] = None


def default_library() -> Library:
    global _default
    if _default is None:
        _default = Library()
    return _default


class Model:
    """Owning handle of an ahip_model."""

    def __init__(self, path: str, device: int = 0, lib: Optional[Library] = None):
        self.L = lib or default_library()
        h = C.c_void_p()
        self.L.check(self.L.lib.ahip_model_load(os.fsencode(path), device, C.byref(h)))
        self.h = h
        self._keep = []
        r = C.c_double(); nt = C.c_int(); tn = C.c_char_p(); pc = C.POINTER(C.c_double)()
        lm = C.c_int(); U = C.c_int(); S = C.c_int(); nl = C.c_int(); dt = C.c_char_p()
        self.L.check(self.L.lib.ahip_model_meta(h, C.byref(r), C.byref(nt), C.byref(tn), C.byref(pc), C.byref(lm),
                                                 C.byref(U), C.byref(S), C.byref(nl), C.byref(dt)))
        self.r_max = r.value
        self.num_types = nt.value
        self.type_names = tn.value.decode().split()
        self.per_edge_type_cutoff = (np.ctypeslib.as_array(pc, shape=(nt.value, nt.value)).copy() if pc else None)
        self.l_max, self.num_tensor_features, self.num_scalar_features = lm.value, U.value, S.value
        self.num_layers, self.model_dtype = nl.value, dt.value.decode()
        tf = C.c_int(0)
        self.L.check(self.L.lib.ahip_model_allow_tf32(h, C.byref(tf)))
        self.allow_tf32 = bool(tf.value)

    def close(self):
        if getattr(self, "h", None):
            self.L.lib.ahip_model_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, key: str, value) -> None:
        self.L.check(self.L.lib.ahip_set_option(self.h, key.encode(), str(value).encode()))

    # ---- neighbor lists ----------------------------------------------------------------------
    def neigh_update_csr(self, nall: int, ilist: np.ndarray, offsets: np.ndarray, neigh: np.ndarray,
                         neighmask: int = 0x1FFFFFFF) -> None:
        ilist = np.ascontiguousarray(ilist, dtype=np.int32)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        neigh = np.ascontiguousarray(neigh, dtype=np.int32)
        self.L.check(self.L.lib.ahip_neigh_update_csr(self.h, len(ilist), nall, _p(ilist, C.c_int),
                                                       _p(offsets, C.c_longlong), _p(neigh, C.c_int), neighmask))

    def neigh_update_paged(self, nall: int, ilist: np.ndarray, numneigh: np.ndarray, rows: Sequence[np.ndarray],
                           neighmask: int = 0x1FFFFFFF) -> None:
        """LAMMPS-shaped call: numneigh[i] and firstneigh[i] indexed by atom index."""
        ilist = np.ascontiguousarray(ilist, dtype=np.int32)
        numneigh = np.ascontiguousarray(numneigh, dtype=np.int32)
        rows = [np.ascontiguousarray(r, dtype=np.int32) for r in rows]
        first = (C.POINTER(C.c_int) * len(rows))(*[_p(r, C.c_int) for r in rows])
        self.L.check(self.L.lib.ahip_neigh_update(self.h, len(ilist), nall, _p(ilist, C.c_int), _p(numneigh, C.c_int),
                                                   first, neighmask))

    def neigh_update_dev(self, inum: int, nall: int, ilist_ptr: int, off_ptr: int, neigh_ptr: int, total: int) -> None:
        self.L.check(self.L.lib.ahip_neigh_update_dev(self.h, inum, nall, ilist_ptr, off_ptr, neigh_ptr, total))

    def neigh_update_dev_table(self, inum: int, nall: int, ilist_ptr: int, numneigh_ptr: int, table_ptr: int, stride_atom: int,
                               stride_slot: int, neighmask: int = 0x1FFFFFFF, stream: int = 0) -> None:
        """Device-resident KOKKOS-layout list (padded 2-D table) -> the library's CSR rows."""
        self.L.check(self.L.lib.ahip_neigh_update_dev_table(self.h, inum, nall, ilist_ptr, numneigh_ptr, table_ptr, stride_atom, stride_slot,
                                                             neighmask, stream))

    def map_types_dev(self, n: int, type_ptr: int, type_mapper: np.ndarray, mtype_ptr: int, stream: int = 0) -> None:
        tm = np.ascontiguousarray(type_mapper, dtype=np.int32)
        self.L.check(self.L.lib.ahip_map_types_dev(self.h, n, type_ptr, len(tm), _p(tm, C.c_int), mtype_ptr, stream))

    def reneighbor_flag_dev(self, n: int, x_ptr: int, xhold_ptr: int, v_ptr: int, dt: float, half_skin: float, flag_ptr: int, stream: int = 0) -> None:
        self.L.check(self.L.lib.ahip_reneighbor_flag_dev(self.h, n, x_ptr, xhold_ptr, v_ptr, dt, half_skin, flag_ptr, stream or None))

    def build_neighbors_dev(self, nlocal: int, nall: int, x_ptr: int, lo, hi, rc_list: float, stream: int = 0) -> None:
        lo = np.ascontiguousarray(lo, dtype=np.float64)
        hi = np.ascontiguousarray(hi, dtype=np.float64)
        self.L.check(self.L.lib.ahip_build_neighbors_dev(self.h, nlocal, nall, x_ptr, _p(lo, C.c_double),
                                                          _p(hi, C.c_double), rc_list, stream))

    # ---- compute ------------------------------------------------------------------------------
    def compute(self, nlocal: int, nghost: int, x: np.ndarray, type_: np.ndarray, type_mapper: np.ndarray,
                cutoff_matrix: np.ndarray, f: np.ndarray, eatom: Optional[np.ndarray] = None, want_virial: bool = True):
        """f is accumulated in place; returns (eng, virial[6] or None)."""
        assert x.dtype == np.float64 and x.flags.c_contiguous and f.dtype == np.float64 and f.flags.c_contiguous
        type_ = np.ascontiguousarray(type_, dtype=np.int32)
        type_mapper = np.ascontiguousarray(type_mapper, dtype=np.int32)
        cutoff_matrix = np.ascontiguousarray(cutoff_matrix, dtype=np.float64)
        ntypes = len(type_mapper)
        eng = C.c_double(0)
        vir = np.zeros(6) if want_virial else None
        self.L.check(self.L.lib.ahip_compute(self.h, nlocal, nghost, _p(x, C.c_double), _p(type_, C.c_int), ntypes,
                                              _p(type_mapper, C.c_int), _p(cutoff_matrix, C.c_double),
                                              _p(f, C.c_double), _p(eatom, C.c_double), C.byref(eng),
                                              _p(vir, C.c_double)))
        return eng.value, vir

    def compute_dev(self, nlocal: int, nghost: int, x_ptr: int, mtype_ptr: int, f_ptr: int, eatom_ptr: int,
                    engvir_ptr: int, cutoff_matrix_model: Optional[np.ndarray] = None, stream: int = 0) -> None:
        cm = None
        if cutoff_matrix_model is not None:
            cm = np.ascontiguousarray(cutoff_matrix_model, dtype=np.float64)
        self.L.check(self.L.lib.ahip_compute_dev(self.h, nlocal, nghost, x_ptr, mtype_ptr, _p(cm, C.c_double), f_ptr,
                                                  eatom_ptr or None, engvir_ptr, stream or None))

    def compute_dev_range(self, c0: int, c1: int, nlocal: int, nghost: int, x_ptr: int, mtype_ptr: int, f_ptr: int, eatom_ptr: int,
                          engvir_ptr: int, cutoff_matrix_model: Optional[np.ndarray] = None, stream: int = 0) -> None:
        cm = None
        if cutoff_matrix_model is not None:
            cm = np.ascontiguousarray(cutoff_matrix_model, dtype=np.float64)
        self.L.check(self.L.lib.ahip_compute_dev_range(self.h, c0, c1, nlocal, nghost, x_ptr, mtype_ptr, _p(cm, C.c_double), f_ptr,
                                                        eatom_ptr or None, engvir_ptr, stream or None))

    def borders_local_dev(self, nlocal: int, x_ptr: int, mtype_ptr: int, lo, hi, box, rc: float, capacity: int, xg_ptr: int, mtg_ptr: int,
                          src_ptr: int, shift_ptr: int, stream: int = 0) -> int:
        """periodic images of a single rank's own atoms inside the halo (ahip_borders_local_dev); returns their number (> capacity: retry with larger arrays)"""
        a = [np.ascontiguousarray(v, dtype=np.float64) for v in (lo, hi, box)]
        ng = C.c_int(0)
        self.L.lib.ahip_borders_local_dev.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                                      C.c_double, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.c_void_p]
        self.L.check(self.L.lib.ahip_borders_local_dev(self.h, nlocal, x_ptr or None, mtype_ptr or None, _p(a[0], C.c_double), _p(a[1], C.c_double), _p(a[2], C.c_double),
                                                        float(rc), capacity, xg_ptr or None, mtg_ptr or None, src_ptr or None, shift_ptr or None, C.byref(ng), stream or None))
        return int(ng.value)

    def nve_dev(self, mode: int, n: int, x_ptr: int, v_ptr: int, f_ptr: int, mtype_ptr: int, mass_by_mtype,
                dt: float, ftm2v: float, stream: int = 0) -> None:
        mass = np.ascontiguousarray(mass_by_mtype, dtype=np.float64)
        self.L.check(self.L.lib.ahip_nve_dev(self.h, mode, n, x_ptr, v_ptr, f_ptr, mtype_ptr, _p(mass, C.c_double),
                                              dt, ftm2v, stream or None))

    def nve_first_dev(self, nlocal: int, nall: int, x_ptr: int, v_ptr: int, f_ptr: int, mtype_ptr: int, mass_by_mtype,
                      dt: float, ftm2v: float, stream: int = 0) -> None:
        mass = np.ascontiguousarray(mass_by_mtype, dtype=np.float64)
        self.L.lib.ahip_nve_first_dev.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                  C.POINTER(C.c_double), C.c_double, C.c_double, C.c_void_p]
        self.L.check(self.L.lib.ahip_nve_first_dev(self.h, nlocal, nall, x_ptr, v_ptr, f_ptr, mtype_ptr, _p(mass, C.c_double),
                                                    dt, ftm2v, stream or None))

    # ---- introspection -----------------------------------------------------------------------
    def nedges(self) -> int:
        n = C.c_longlong(0)
        self.L.check(self.L.lib.ahip_get_edges(self.h, C.byref(n), None, None))
        return int(n.value)

    def nneigh(self) -> int:
        return int(self.L.lib.ahip_last_list_size(self.h))

    def get_edges(self):
        n = C.c_longlong(0)
        self.L.check(self.L.lib.ahip_get_edges(self.h, C.byref(n), None, None))
        E = n.value
        ei = np.zeros((2, E), dtype=np.int64)
        r = np.zeros(E)
        if E:
            self.L.check(self.L.lib.ahip_get_edges(self.h, C.byref(n), _p(ei, C.c_longlong), _p(r, C.c_double)))
        return ei, r

    def debug_dump_edges(self, tag: Optional[np.ndarray] = None) -> None:
        t = np.ascontiguousarray(tag, dtype=np.int32) if tag is not None else None
        self.L.check(self.L.lib.ahip_debug_dump_edges(self.h, _p(t, C.c_int)))

    def timings(self) -> dict:
        names = C.c_char_p(); ms = C.POINTER(C.c_double)(); n = C.c_int(0)
        self.L.check(self.L.lib.ahip_get_timings(self.h, C.byref(names), C.byref(ms), C.byref(n)))
        if n.value == 0:
            return {}
        return dict(zip(names.value.decode().split(";"), [ms[k] for k in range(n.value)]))

    def tile_occupancy(self):
        """(slots holding an edge, slots of all tiles) of the last fused evaluation."""
        a = C.c_longlong(0); b = C.c_longlong(0)
        self.L.lib.ahip_last_tile_occupancy.argtypes = [C.c_void_p, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]
        self.L.check(self.L.lib.ahip_last_tile_occupancy(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def timings_and_counts(self):
        """({stage: summed ms}, {stage: launches}) since the previous timings() / timings_and_counts() call."""
        names = C.c_char_p(); ms = C.POINTER(C.c_double)(); n = C.c_int(0)
        self.L.check(self.L.lib.ahip_get_timings(self.h, C.byref(names), C.byref(ms), C.byref(n)))
        if n.value == 0:
            return {}, {}
        cnt = C.POINTER(C.c_double)(); n2 = C.c_int(0)
        self.L.check(self.L.lib.ahip_get_timing_counts(self.h, C.byref(cnt), C.byref(n2)))
        keys = names.value.decode().split(";")
        return dict(zip(keys, [ms[k] for k in range(n.value)])), dict(zip(keys, [int(cnt[k]) for k in range(n2.value)]))

    @property
    def last_path(self) -> str:
        return self.L.lib.ahip_last_path(self.h).decode()

    @property
    def arith_note(self) -> str:
        """What fused_arith=auto decided for this model and why (empty before the first evaluation)."""
        return self.L.lib.ahip_arith_note(self.h).decode()

    @property
    def last_max_degree(self) -> int:
        return int(self.L.lib.ahip_last_max_degree(self.h))

    # ---- `compute allegro` outputs (pair_nequip_allegro.cpp:403-406,681-684) ---------------------
    def output_register(self, name: str) -> None:
        self.L.check(self.L.lib.ahip_output_register(self.h, name.encode()))

    def output_get(self, name: str) -> np.ndarray:
        n = C.c_longlong(0)
        self.L.check(self.L.lib.ahip_output_get(self.h, name.encode(), None, 0, C.byref(n)))
        out = np.zeros(n.value, dtype=np.float64)
        self.L.check(self.L.lib.ahip_output_get(self.h, name.encode(), out.ctypes.data_as(C.POINTER(C.c_double)), n.value, C.byref(n)))
        return out


class Comm:
    """Owning handle of an ahip_comm: the ghost exchange of a decomposed system (csrc/comm.hip).  Transports: "rccl" (unique id from
    rank 0, handed over by the caller) or "hosted" (a Python callable moves the packed buffers: CPU tests, ranks sharing one GPU)."""

    def __init__(self, lib: Library, rank: int, nranks: int, rccl_id: Optional[bytes] = None, device: int = 0, xfer=None):
        self.L = lib
        self.rank, self.nranks = rank, nranks
        self.h = C.c_void_p()
        self._keep = []
        self._cb = None
        if rccl_id is not None:
            assert len(rccl_id) == 128
            self.L.check(self.L.lib.ahip_comm_create_rccl(rank, nranks, rccl_id, device, C.byref(self.h)))
            self.transport = "rccl"
        else:
            if xfer is not None:
                def _cb(user, nops, ops):
                    try:
                        xfer([(ops[k].kind, ops[k].peer, ops[k].ptr, ops[k].bytes) for k in range(nops)])
                        return 0
                    except Exception as e:          # never let an exception cross the C boundary
                        import traceback
                        traceback.print_exc()
                        return 1
                self._cb = XFER_FN(_cb)
            else:
                self._cb = C.cast(None, XFER_FN)
            self.L.check(self.L.lib.ahip_comm_create_hosted(rank, nranks, self._cb, None, C.byref(self.h)))
            self.transport = "hosted"

    @staticmethod
    def unique_id(lib: Library) -> bytes:
        buf = C.create_string_buffer(128)
        lib.check(lib.lib.ahip_comm_unique_id(buf))
        return buf.raw

    def close(self):
        if getattr(self, "h", None):
            self.L.lib.ahip_comm_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_plan(self, dims, sendranks, recvranks, shifts, nsend, nrecv, first_recv, idx_ptrs) -> None:
        n = len(dims)
        ai = lambda v: np.ascontiguousarray(v, dtype=np.int32)
        a = [ai(dims), ai(sendranks), ai(recvranks), np.ascontiguousarray(shifts, dtype=np.float64), ai(nsend), ai(nrecv), ai(first_recv)]
        ptrs = (C.c_void_p * max(n, 1))(*[int(p) for p in idx_ptrs])
        self._keep = a + [ptrs]
        self.L.check(self.L.lib.ahip_comm_set_plan(self.h, n, _p(a[0], C.c_int), _p(a[1], C.c_int), _p(a[2], C.c_int), _p(a[3], C.c_double),
                                                   _p(a[4], C.c_int), _p(a[5], C.c_int), _p(a[6], C.c_int), ptrs))

    def set_plan_local(self, nlocal: int, nghost: int, src_ptr: int, shift_ptr: int) -> None:
        self.L.check(self.L.lib.ahip_comm_set_plan_local(self.h, nlocal, nghost, src_ptr or None, shift_ptr or None))

    def forward(self, x_ptr: int, stream: int = 0) -> None:
        self.L.check(self.L.lib.ahip_comm_forward(self.h, x_ptr, stream or None))

    def reverse(self, f_ptr: int, stream: int = 0) -> None:
        self.L.check(self.L.lib.ahip_comm_reverse(self.h, f_ptr, stream or None))

    def migrate(self, nlocal: int, x_ptr: int, v_ptr: int, tag_ptr: int, mtype_ptr: int, capacity: int, box, grid, coord, stream: int = 0) -> int:
        """Comm::exchange inside the library (csrc/comm.hip: comm_migrate); returns the new owned count, negative = rows some brick would need."""
        b = np.ascontiguousarray(box, dtype=np.float64); g = np.ascontiguousarray(grid, dtype=np.int32); c = np.ascontiguousarray(coord, dtype=np.int32)
        out = C.c_int(0)
        self.L.check(self.L.lib.ahip_comm_migrate(self.h, nlocal, C.c_void_p(x_ptr), C.c_void_p(v_ptr), C.c_void_p(tag_ptr), C.c_void_p(mtype_ptr), capacity,
                                                  _p(b, C.c_double), _p(g, C.c_int), _p(c, C.c_int), C.byref(out), stream or None))
        return out.value

    def borders(self, nlocal: int, x_ptr: int, mtype_ptr: int, capacity: int, lo, hi, box, rc: float, grid, coord, stream: int = 0) -> int:
        """Comm::borders inside the library (comm_borders): ghosts behind the owned rows, the exchange plan installed; returns owned + ghost rows (> capacity: retry)."""
        a = [np.ascontiguousarray(v, dtype=np.float64) for v in (lo, hi, box)]
        g = np.ascontiguousarray(grid, dtype=np.int32); c = np.ascontiguousarray(coord, dtype=np.int32)
        out = C.c_int(0)
        self.L.check(self.L.lib.ahip_comm_borders(self.h, nlocal, C.c_void_p(x_ptr), C.c_void_p(mtype_ptr), capacity, _p(a[0], C.c_double), _p(a[1], C.c_double),
                                                  _p(a[2], C.c_double), float(rc), _p(g, C.c_int), _p(c, C.c_int), C.byref(out), stream or None))
        return out.value

    def allreduce(self, ptr: int, count: int, kind: int, stream: int = 0) -> None:
        self.L.check(self.L.lib.ahip_comm_allreduce(self.h, ptr, count, kind, stream or None))

    def selftest(self, n: int = 1024, stream: int = 0) -> None:
        self.L.check(self.L.lib.ahip_comm_selftest(self.h, n, stream or None))
