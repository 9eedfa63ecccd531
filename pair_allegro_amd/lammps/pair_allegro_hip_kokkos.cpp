/* ----------------------------------------------------------------------
   pair_style allegro/kk over liballegro_hip: the KOKKOS (HIP) coupling.  Line references are to the reference
   implementation this file replaces, mir-group/pair_allegro pair_nequip_allegro_kokkos.cpp.

   Per step the reference (a) filters the whole padded neighbor table into a second padded table, (b) scans, (c) writes int64
   edges plus padding edges, (d) copies positions and mapped types, (e) calls libtorch, (f) adds forces in a reduction kernel
   (:142-319).  Here: on list-rebuild steps the table is compacted once on the device (ahip_neigh_update_dev_table) and the
   types are mapped (ahip_map_types_dev); every step is ONE library call that reads x and accumulates into f in place.
------------------------------------------------------------------------- */

#include "pair_allegro_hip_kokkos.h"

#include <type_traits>

#include "allegro_hip.h"

#include "atom_kokkos.h"
#include "atom_masks.h"
#include "comm.h"
#include "error.h"
#include "force.h"
#include "kokkos.h"
#include "memory_kokkos.h"
#include "neigh_list_kokkos.h"
#include "neigh_request.h"
#include "neighbor.h"

using namespace LAMMPS_NS;

PairAllegroHIPKokkos::PairAllegroHIPKokkos(LAMMPS *lmp) : PairAllegroHIP(lmp)
{
  respa_enable = 0;    // :60
  kokkosable = 1;
  atomKK = (AtomKokkos *) atom;
  execution_space = ExecutionSpaceFromDevice<DeviceType>::space;
  datamask_read = X_MASK | F_MASK | TAG_MASK | TYPE_MASK | ENERGY_MASK | VIRIAL_MASK;    // :65
  datamask_modify = F_MASK | ENERGY_MASK | VIRIAL_MASK;                                 // :66
  d_engvir = Kokkos::View<double *, DeviceType>("Allegro: engvir", 8);
  h_engvir = Kokkos::create_mirror_view(d_engvir);
}

PairAllegroHIPKokkos::~PairAllegroHIPKokkos()
{
  if (!copymode) {    // :76-81
    memoryKK->destroy_kokkos(k_eatom, eatom);
    eatom = nullptr;
  }
}

void PairAllegroHIPKokkos::coeff(int narg, char **arg)
{
  PairAllegroHIP::coeff(narg, arg);    // model load, type mapping, setflag, LAMMPS-index cutoff matrix (:366)
  // The device path filters in model-type index with the model's own per-edge-type matrix (the reference uploads the
  // LAMMPS-index matrix and looks it up by LAMMPS type, :377-386 -- the same numbers, since the one is built from the other).
  ahip_model_meta(model, nullptr, nullptr, nullptr, &cutoff_model, nullptr, nullptr, nullptr, nullptr, nullptr);
  // the KOKKOS path of the reference keeps an edge iff rsq < cut^2 (:174), the host path iff rsq <= cut^2
  if (ahip_set_option(model, "cutoff_compare", "lt") != AHIP_OK) error->all(FLERR, "pair_allegro/kk: {}", ahip_last_error());
}

void PairAllegroHIPKokkos::init_style()
{
  PairAllegroHIP::init_style();    // atom IDs, full + ghost request, newton on (:391)
  auto request = neighbor->find_request(this);
  request->set_kokkos_host(std::is_same<DeviceType, LMPHostType>::value && !std::is_same<DeviceType, LMPDeviceType>::value);    // :394-396
  request->set_kokkos_device(std::is_same<DeviceType, LMPDeviceType>::value);
  neighflag = lmp->kokkos->neighflag;
  if (neighflag == FULL)    // same requirement, same message as the reference (:399-401)
    error->all(FLERR, "pair style allegro/kk requires the 'neigh half' flag due to 'newton on'");
}

void PairAllegroHIPKokkos::compute(int eflag_in, int vflag_in)
{
  ev_init(eflag_in, vflag_in, 0);
  if (vflag_atom) error->all(FLERR, "Pair style Allegro does not support per-atom virial");    // :336-338

  if (eflag_atom) {    // :97-101
    memoryKK->destroy_kokkos(k_eatom, eatom);
    memoryKK->create_kokkos(k_eatom, eatom, maxeatom, "pair:eatom");
    d_eatom = k_eatom.view<DeviceType>();
  }

  atomKK->sync(execution_space, datamask_read);    // :108-110
  if (eflag_in || vflag_in) atomKK->modified(execution_space, datamask_modify);
  else atomKK->modified(execution_space, F_MASK);

  x = atomKK->k_x.view<DeviceType>();
  f = atomKK->k_f.view<DeviceType>();
  type = atomKK->k_type.view<DeviceType>();
  const int nlocal = atom->nlocal, nghost = atom->nghost, nall = nlocal + nghost;
  const int inum = list->inum;
  if (inum == 0) return;    // empty domain (:128)

  void *stream = (void *) DeviceType().hip_stream();    // the execution space instance the views' last writers ran on

  if (neighbor->ago == 0 || last_list_build < 0) {    // list rebuilt this step, or a new run: see PairAllegroHIP::compute
    NeighListKokkos<DeviceType> *k_list = static_cast<NeighListKokkos<DeviceType> *>(list);    // :121-124
    d_ilist = k_list->d_ilist;
    d_numneigh = k_list->d_numneigh;
    d_neighbors = k_list->d_neighbors;
    if (ahip_neigh_update_dev_table(model, inum, nall, d_ilist.data(), d_numneigh.data(), d_neighbors.data(),
                                    (long long) d_neighbors.stride(0), (long long) d_neighbors.stride(1), NEIGHMASK, stream) != AHIP_OK)
      error->one(FLERR, "pair_allegro/kk: {}", ahip_last_error());
    if ((int) d_mtype.extent(0) < nall) d_mtype = Kokkos::View<int *, DeviceType>("Allegro: model types", (size_t)(1.05 * nall) + 2);    // growth like :209-220
    if (ahip_map_types_dev(model, nall, type.data(), atom->ntypes, type_mapper.data(), d_mtype.data(), stream) != AHIP_OK)
      error->one(FLERR, "pair_allegro/kk: {}", ahip_last_error());
    last_list_build = 1;
  }

  // the C-ABI takes float64 positions and forces (the reference's inputtype / outputtype, pair_nequip_allegro.h:73-78): a KOKKOS build
  // with single or mixed precision views must not be reinterpreted silently
  static_assert(sizeof(typename std::remove_reference<decltype(x(0, 0))>::type) == sizeof(double) &&
                sizeof(typename std::remove_reference<decltype(f(0, 0))>::type) == sizeof(double),
                "pair_style allegro/kk needs KOKKOS built with double-precision X_FLOAT and F_FLOAT");
  // forces are accumulated in place into the KOKKOS force view, E_i written for the inum centre atoms (:305-319)
  if (ahip_compute_dev(model, nlocal, nghost, x.data(), d_mtype.data(), cutoff_model, f.data(), eflag_atom ? d_eatom.data() : nullptr,
                       d_engvir.data(), stream) != AHIP_OK)
    error->one(FLERR, "pair_allegro/kk: {}", ahip_last_error());
  Kokkos::deep_copy(DeviceType(), h_engvir, d_engvir);
  DeviceType().fence();

  eng_vdwl = h_engvir(0);    // sum over the local atoms only (:313-316)
  if (vflag_in) for (int k = 0; k < 6; k++) virial[k] = h_engvir(1 + k);    // xx,yy,zz,xy,xz,yz, no sign change (:327-334)

  if (eflag_atom) {    // :321-326
    k_eatom.modify<DeviceType>();
    k_eatom.sync<LMPHostType>();
  }
}
