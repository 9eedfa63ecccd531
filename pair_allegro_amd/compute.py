"""Python mirror of the reference's `compute allegro` / `compute allegro/atom`
(/root/reference/compute/compute_allegro.cpp:39-189, compute/README.md): extracts a named entry of the model's
output dict from the pair style after a force evaluation.

  compute ID all allegro      <quantity> <length>             -> global vector, summed over ranks (assumed extensive)
  compute ID all allegro/atom <quantity> <length> <newton>    -> per-atom array; newton=1 reverse-communicates ghost rows

Same argument checks and error texts as the reference; the MPI all-reduce / reverse communication are the caller's
(this mirror returns the rank's contribution and exposes pack/unpack like the LAMMPS class)."""
from typing import List, Optional

import numpy as np

from .pair import LammpsError, PairAllegro


class ComputeAllegro:
    def __init__(self, args: List[str], pair: Optional[PairAllegro], me: int = 0, quiet: bool = True):
        # args as LAMMPS hands them over: [ID, group, style, quantity, length(, newton)]
        style = args[2] if len(args) > 2 else ""
        self.peratom = style == "allegro/atom"
        if not self.peratom:
            if len(args) != 5:                                                     # :44-45
                raise LammpsError("Incorrect args for compute allegro")
        else:
            if len(args) != 6:                                                     # :47-48
                raise LammpsError("Incorrect args for compute allegro/atom")
        if args[1] != "all":                                                       # :51-52
            raise LammpsError("compute allegro can only operate on group 'all'")
        self.quantity = args[3]
        self.me = me
        if self.peratom:                                                           # :55-64
            self.nperatom = int(args[4])
            self.newton = int(args[5])
            self.comm_reverse = self.nperatom if self.newton else 0
            self.size_peratom_cols = 0 if self.nperatom == 1 else self.nperatom
            self.array_atom: Optional[np.ndarray] = None
            if me == 0 and not quiet:
                print(f"compute allegro/atom will evaluate the quantity {self.quantity} of length "
                      f"{self.size_peratom_cols} with newton {self.newton}")
        else:                                                                      # :65-75
            self.size_vector = int(args[4])
            if self.size_vector <= 0:
                raise LammpsError("Incorrect vector length!")
            self.vector = np.zeros(self.size_vector)
            self.extvector = 1
            if me == 0 and not quiet:
                print(f"compute allegro will evaluate the quantity {self.quantity} of length {self.size_vector}")
        if pair is None:                                                           # :77-79
            raise LammpsError("no pair style; compute allegro must be defined after pair style")
        self.pair = pair
        pair.add_custom_output(self.quantity)                                     # :81
        self._quantity_flat: Optional[np.ndarray] = None

    # ---- compute allegro (:104-128) ---------------------------------------------------------------
    def compute_vector(self, nlocal: int) -> np.ndarray:
        """This rank's contribution; the caller sums over ranks (MPI_Allreduce, :127)."""
        assert not self.peratom
        if nlocal == 0:                                                            # empty domain (:108-112)
            self.vector[:] = 0.0
            return self.vector
        q = self.pair.custom_output(self.quantity)
        if q.size != self.size_vector:                                             # :118-121
            raise LammpsError(f"size {q.size} of quantity tensor {self.quantity} does not match expected "
                              f"{self.size_vector} on rank {self.me}")
        self.vector[:] = q
        return self.vector

    # ---- compute allegro/atom (:130-160) ----------------------------------------------------------
    def compute_peratom(self, nlocal: int, nmax: int) -> np.ndarray:
        """array_atom [nmax][nperatom] with the local rows filled; with newton=1 the caller then runs the reverse
        communication through pack_reverse_comm / unpack_reverse_comm (:159)."""
        assert self.peratom
        if self.array_atom is None or self.array_atom.shape[0] < nmax:
            self.array_atom = np.zeros((nmax, self.nperatom))
        if nlocal > 0:                                                             # :143-156
            q = self.pair.custom_output(self.quantity).reshape(-1, self.nperatom)
            self._quantity_flat = q.ravel()
            self.array_atom[:nlocal] = q[:nlocal]
        return self.array_atom

    def pack_reverse_comm(self, n: int, first: int) -> np.ndarray:                 # :163-176: rows of the MODEL output
        q = self._quantity_flat.reshape(-1, self.nperatom)
        return q[first:first + n].ravel().copy()

    def unpack_reverse_comm(self, owners: np.ndarray, buf: np.ndarray) -> None:    # :178-189
        np.add.at(self.array_atom, np.asarray(owners), buf.reshape(-1, self.nperatom))
