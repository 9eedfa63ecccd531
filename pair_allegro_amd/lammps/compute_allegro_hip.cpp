// `compute allegro` / `compute allegro/atom` on top of PairAllegroHIP.  Behaviour follows the reference compute
// (/root/reference/compute/compute_allegro.cpp; cited by line); only the source of the numbers differs.
#include "compute_allegro_hip.h"

#include "atom.h"
#include "comm.h"
#include "error.h"
#include "force.h"
#include "memory.h"
#include "pair_allegro_hip.h"
#include "update.h"

#include <cstdlib>
#include <cstring>

using namespace LAMMPS_NS;

template <int peratom> ComputeAllegroHIP<peratom>::ComputeAllegroHIP(LAMMPS *lmp, int narg, char **arg) : Compute(lmp, narg, arg)
{
  if (!peratom) {
    if (narg != 5) error->all(FLERR, "Incorrect args for compute allegro");         // :44-45
  } else {
    if (narg != 6) error->all(FLERR, "Incorrect args for compute allegro/atom");    // :47-48
  }
  if (strcmp(arg[1], "all") != 0) error->all(FLERR, "compute allegro can only operate on group 'all'");    // :51-52

  quantity = arg[3];
  newton = 0;
  nperatom = 0;
  nmax = 0;
  if (peratom) {    // :55-64
    peratom_flag = 1;
    nperatom = std::atoi(arg[4]);
    newton = std::atoi(arg[5]);
    if (newton) comm_reverse = nperatom;
    size_peratom_cols = nperatom == 1 ? 0 : nperatom;
  } else {    // :65-75; global quantities are assumed extensive and summed over ranks (compute/README.md)
    vector_flag = 1;
    extvector = 1;
    size_vector = std::atoi(arg[4]);
    if (size_vector <= 0) error->all(FLERR, "Incorrect vector length!");
    memory->create(vector, size_vector, "ComputeAllegro:vector");
  }
  if (force->pair == nullptr) error->all(FLERR, "no pair style; compute allegro must be defined after pair style");    // :77-79
  ((PairAllegroHIP *) force->pair)->add_custom_output(quantity);                                                        // :81
}

template <int peratom> void ComputeAllegroHIP<peratom>::init() {}

template <int peratom> ComputeAllegroHIP<peratom>::~ComputeAllegroHIP()
{
  if (copymode) return;
  if (peratom) memory->destroy(array_atom);
  else memory->destroy(vector);
}

template <int peratom> void ComputeAllegroHIP<peratom>::compute_vector()    // :104-128
{
  invoked_vector = update->ntimestep;
  if (atom->nlocal == 0) {    // empty domain: the pair style stored nothing
    for (int i = 0; i < size_vector; i++) vector[i] = 0.0;
  } else {
    quantity_values = ((PairAllegroHIP *) force->pair)->custom_output(quantity);
    if ((int) quantity_values.size() != size_vector)
      error->one(FLERR, "size of quantity tensor {} does not match the expected length", quantity.c_str());
    for (int i = 0; i < size_vector; i++) vector[i] = quantity_values[i];
  }
  MPI_Allreduce(MPI_IN_PLACE, vector, size_vector, MPI_DOUBLE, MPI_SUM, world);    // even if empty domain
}

template <int peratom> void ComputeAllegroHIP<peratom>::compute_peratom()    // :130-160
{
  invoked_peratom = update->ntimestep;
  if (atom->nmax > nmax) {
    nmax = atom->nmax;
    memory->destroy(array_atom);
    memory->create(array_atom, nmax, nperatom, "allegro/atom:array");
    if (nperatom == 1) vector_atom = &array_atom[0][0];
  }
  if (atom->nlocal > 0) {    // guard against empty domain
    quantity_values = ((PairAllegroHIP *) force->pair)->custom_output(quantity);
    const int nlocal = atom->nlocal;
    if (quantity_values.size() < (size_t) nlocal * nperatom)
      error->one(FLERR, "size of quantity tensor {} does not match the expected per-atom length", quantity.c_str());
    for (int i = 0; i < nlocal; i++)
      for (int j = 0; j < nperatom; j++) array_atom[i][j] = quantity_values[(size_t) i * nperatom + j];
  }
  if (newton) comm->reverse_comm(this);    // even if empty domain
}

template <int peratom> int ComputeAllegroHIP<peratom>::pack_reverse_comm(int n, int first, double *buf)    // :163-176
{
  int m = 0;
  const int last = first + n;
  for (int i = first; i < last; i++)
    for (int j = 0; j < nperatom; j++) buf[m++] = quantity_values[(size_t) i * nperatom + j];
  return m;
}

template <int peratom> void ComputeAllegroHIP<peratom>::unpack_reverse_comm(int n, int *list, double *buf)    // :178-189
{
  int m = 0;
  for (int i = 0; i < n; i++) {
    const int j = list[i];
    for (int k = 0; k < nperatom; k++) array_atom[j][k] += buf[m++];
  }
}

namespace LAMMPS_NS {
template class ComputeAllegroHIP<0>;
template class ComputeAllegroHIP<1>;
}    // namespace LAMMPS_NS
