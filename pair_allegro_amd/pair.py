"""Host-side mirror of the reference pair style's operator interface, above the C-ABI.

``PairAllegro`` has the methods LAMMPS calls on ``PairNequIPAllegro<false>``
(/root/reference/pair_nequip_allegro.h:43-50): ``settings``, ``coeff``, ``init_style``,
``init_one``, ``compute`` -- same argument meaning, same checks, same error texts -- so the tests
read like the reference's own (deck of tests/test_python_repro_allegro.py:84-120).  The C++
counterpart that actually plugs into LAMMPS is pair_allegro_amd/lammps/pair_allegro_hip.{h,cpp};
both are thin marshalling layers over liballegro_hip.so.
"""
from __future__ import annotations

import os
import sys
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np

from . import capi
from .lmp_like import NEIGHMASK, RankSystem


class LammpsError(RuntimeError):
    """error->all(FLERR, ...) of the reference."""


@dataclass
class Atom:
    """The slice of LAMMPS ``Atom`` the pair style touches."""
    ntypes: int
    nlocal: int
    nghost: int
    x: np.ndarray
    f: np.ndarray
    type: np.ndarray
    tag: np.ndarray
    tag_enable: int = 1


@dataclass
class NeighList:
    inum: int
    gnum: int
    ilist: np.ndarray
    numneigh: np.ndarray
    firstneigh: Sequence[np.ndarray]


class PairAllegro:
    def __init__(self, me: int = 0, nprocs: int = 1, shm_rank: Optional[int] = None,
                 lib: Optional[capi.Library] = None, quiet: bool = False):
        # pair_nequip_allegro.cpp:66-125
        self.restartinfo = 0
        self.manybody_flag = 1
        self.me, self.nprocs = me, nprocs
        self.quiet = quiet
        self.lib = lib or capi.default_library()
        self.debug_mode = 1 if os.environ.get("_NEQUIP_LOG_LEVEL") == "DEBUG" else 0      # :78-83
        if self.debug_mode and not quiet:
            print("Debug mode enabled, since _NEQUIP_LOG_LEVEL is set to DEBUG")
        # device selection by node-local rank (:92-120)
        deviceidx = 0
        devicecount = self.lib.device_count()
        if devicecount <= 0:
            raise LammpsError("pair_allegro (HIP): no GPU visible; this pair style has no CPU path")
        if nprocs > 1:
            deviceidx = shm_rank if shm_rank is not None else me
            if deviceidx >= devicecount:
                if self.debug_mode:                                                        # :104-110
                    print(f"WARNING (Allegro): my rank ({deviceidx}) is bigger than the number of visible devices "
                          f"({devicecount}), wrapping around to use device {deviceidx % devicecount} again!!!",
                          file=sys.stderr)
                    deviceidx %= devicecount
                else:                                                                      # :112-117
                    raise LammpsError("pair_allegro: mismatch between number of ranks and number of available GPUs")
        self.device = deviceidx
        self.allocated = 0
        self.model: Optional[capi.Model] = None
        self.cutoff = 0.0
        self.type_mapper: List[int] = []
        self.cutoff_matrix: Optional[np.ndarray] = None
        self.setflag: Optional[np.ndarray] = None
        self.eng_vdwl = 0.0
        self.virial = np.zeros(6)
        self.eatom: Optional[np.ndarray] = None
        self._list_id = None
        self.custom_output_names: List[str] = []

    # ---- Pair::settings (:168-172) ------------------------------------------------------------
    def settings(self, args: Sequence[str]) -> None:
        if len(args) > 0:
            raise LammpsError("Illegal pair_style command, too many arguments")

    # ---- Pair::coeff (:174-330) ---------------------------------------------------------------
    def coeff(self, args: Sequence[str], ntypes: int) -> None:
        self.ntypes = ntypes
        self.setflag = np.zeros((ntypes + 1, ntypes + 1), dtype=np.int32)
        self.allocated = 1
        if len(args) != 3 + ntypes:                                                        # :185-188
            raise LammpsError("Incorrect args for pair coefficients, should be * * <model>.nequip.pth/pt2 "
                              "<type1> <type2> ... <typen>")
        if args[0] != "*" or args[1] != "*":                                               # :191-192
            raise LammpsError("Incorrect args for pair coefficients")
        self.model_path = args[2]
        if self.me == 0 and not self.quiet:
            print(f"NequIP/Allegro: Loading model from {self.model_path}")
        try:
            self.model = capi.Model(self.model_path, self.device, self.lib)                # :214-232
            for name in self.custom_output_names:
                self.model.output_register(name)
        except capi.AhipError as e:
            if e.code == capi.AHIP_ERR_FILE:
                raise RuntimeError(e.msg) from None      # reference throws std::runtime_error (:205)
            raise LammpsError(e.msg) from None
        # :267-270 -- the reference hands the key to at::globalContext().setAllowTF32*; here it selects the reduced-split matrix arithmetic
        # of the fused kernel (include/allegro_hip.h: ahip_model_allow_tf32)
        self.allow_tf32 = self.model.allow_tf32
        if self.me == 0 and not self.quiet and (self.allow_tf32 or self.debug_mode):
            print(f"NequIP/Allegro: model metadata allow_tf32 = {int(self.allow_tf32)}")
        self.cutoff = self.model.r_max                                                     # :272
        names = self.model.type_names
        self.type_mapper = [-1] * ntypes                                                   # :274
        if self.me == 0 and not self.quiet:
            print("Type mapping:")
            print("NequIP/Allegro type | NequIP/Allegro name | LAMMPS type | LAMMPS name")
        for i, ele in enumerate(names):                                                    # :284-294
            for itype in range(1, ntypes + 1):
                if ele == args[itype + 3 - 1]:
                    self.type_mapper[itype - 1] = i
                    if self.me == 0 and not self.quiet:
                        print(f"{i} | {ele} | {itype} | {args[itype + 3 - 1]}")
        for i in range(1, ntypes + 1):                                                     # :297-301
            for j in range(i, ntypes + 1):
                if self.type_mapper[i - 1] >= 0 and self.type_mapper[j - 1] >= 0:
                    self.setflag[i][j] = 1
        cm = np.full((ntypes, ntypes), self.cutoff)                                        # :325-327
        pc = self.model.per_edge_type_cutoff
        if pc is not None:                                                                 # :303-323
            # every LAMMPS type that maps to a model type gets that model type's cutoffs (the
            # reference's reverse_type_mapper keeps only the last LAMMPS type per model type, App. D)
            for a in range(ntypes):
                for b in range(ntypes):
                    if self.type_mapper[a] >= 0 and self.type_mapper[b] >= 0:
                        cm[a, b] = pc[self.type_mapper[a], self.type_mapper[b]]
        self.cutoff_matrix = cm

    # ---- Pair::init_style (:137-151) ----------------------------------------------------------
    def init_style(self, tag_enable: int = 1, newton_pair: int = 1) -> dict:
        if tag_enable == 0:
            raise LammpsError("Pair style Allegro requires atom IDs")
        if newton_pair == 0:
            raise LammpsError("Pair style allegro requires newton pair on")
        return {"full": True, "ghost": True}            # REQ_FULL | REQ_GHOST

    # ---- Pair::init_one (:153-156) ------------------------------------------------------------
    def init_one(self, i: int, j: int) -> float:
        return self.cutoff

    # ---- Pair::compute (:333-407) -------------------------------------------------------------
    def compute(self, atom: Atom, lst: NeighList, eflag_atom: bool = True, vflag: bool = True,
                vflag_atom: bool = False, list_changed: bool = True) -> None:
        if self.model is None:
            raise LammpsError("All pair coeffs are not set")
        if vflag_atom:                                                                     # :394
            raise LammpsError("Pair styles nequip and allegro do not support per-atom virial")
        self.eng_vdwl = 0.0
        self.virial[:] = 0.0
        if lst.inum == 0:                                                                  # :340-341
            return
        nall = atom.nlocal + atom.nghost
        if list_changed or self._list_id != id(lst):
            self.model.neigh_update_paged(nall, lst.ilist, lst.numneigh, lst.firstneigh, NEIGHMASK)
            self._list_id = id(lst)
        if eflag_atom:
            self.eatom = np.zeros(nall)
        try:
            eng, vir = self.model.compute(atom.nlocal, atom.nghost, atom.x, atom.type,
                                          np.asarray(self.type_mapper, dtype=np.int32), self.cutoff_matrix,
                                          atom.f, self.eatom if eflag_atom else None, want_virial=vflag)
        except capi.AhipError as e:
            raise LammpsError(e.msg) from None
        self.eng_vdwl = eng
        if vflag:
            self.virial[:] = vir
        if self.debug_mode:                                                                # :562-565,620-633
            self.model.debug_dump_edges(atom.tag)

    def add_custom_output(self, name: str) -> None:                                        # :681-684
        self.custom_output_names.append(name)
        if self.model is not None:
            self.model.output_register(name)

    def custom_output(self, name: str) -> np.ndarray:
        """`custom_output.at(name).cpu().ravel()` of the reference (pair_nequip_allegro.h:80-82): the entry of the
        model's output dict kept from the last compute()."""
        try:
            return self.model.output_get(name)
        except capi.AhipError as e:
            raise LammpsError(e.msg) from None


def atom_from_rank_system(rs: RankSystem, ntypes: int) -> Atom:
    return Atom(ntypes=ntypes, nlocal=rs.nlocal, nghost=rs.nghost, x=rs.x, f=np.zeros_like(rs.x), type=rs.type,
                tag=rs.tag)


def list_from_rank_system(rs: RankSystem) -> NeighList:
    return NeighList(inum=rs.nlocal, gnum=rs.nghost, ilist=rs.ilist, numneigh=rs.numneigh, firstneigh=rs.firstneigh)
