/* -*- c++ -*- ----------------------------------------------------------
   pair_style allegro/kk -- device-resident coupling of the MI355X-native Allegro pair style to the LAMMPS KOKKOS package
   (HIP backend).  Positions, forces, types and the neighbor list never leave the GPU: the class hands the KOKKOS views'
   device pointers to liballegro_hip's `_dev` entry points (include/allegro_hip.h) and reads back seven doubles per step.

   Reference interface replaced: /root/reference/pair_nequip_allegro_kokkos.h:16,31-110 (PairAllegroKokkos<false>),
   compute/coeff/init_style of /root/reference/pair_nequip_allegro_kokkos.cpp:86-353,364-406.
   The class uses no Kokkos parallel dispatch of its own -- every device loop lives in the library -- so it needs from Kokkos
   only views, dual views and the execution space's stream.
------------------------------------------------------------------------- */

#ifdef PAIR_CLASS
// clang-format off
PairStyle(allegro/kk,PairAllegroHIPKokkos)
PairStyle(allegro/kk/device,PairAllegroHIPKokkos)
// clang-format on
#else

#ifndef LMP_PAIR_ALLEGRO_HIP_KOKKOS_H
#define LMP_PAIR_ALLEGRO_HIP_KOKKOS_H

#include "pair_allegro_hip.h"
#include "pair_kokkos.h"

namespace LAMMPS_NS {

class PairAllegroHIPKokkos : public PairAllegroHIP {
 public:
  using DeviceType = LMPDeviceType;
  typedef LMPDeviceType device_type;
  typedef ArrayTypes<DeviceType> AT;

  PairAllegroHIPKokkos(class LAMMPS *);
  ~PairAllegroHIPKokkos() override;
  void compute(int, int) override;
  void coeff(int, char **) override;
  void init_style() override;

  typename AT::t_efloat_1d d_eatom;

 protected:
  typename AT::t_x_array_randomread x;
  typename AT::t_f_array f;
  typename AT::t_int_1d_randomread type;
  DAT::tdual_efloat_1d k_eatom;

  typename AT::t_neighbors_2d d_neighbors;
  typename AT::t_int_1d_randomread d_ilist;
  typename AT::t_int_1d_randomread d_numneigh;

  Kokkos::View<int *, DeviceType> d_mtype;          // model type of every local + ghost atom, refreshed on list rebuilds
  Kokkos::View<double *, DeviceType> d_engvir;      // {E, xx, yy, zz, xy, xz, yz} of the last evaluation
  Kokkos::View<double *, DeviceType>::HostMirror h_engvir;
  const double *cutoff_model = nullptr;             // [num_types^2] in model-type index (owned by the library), NULL = r_max
  int neighflag = 0;
};

}    // namespace LAMMPS_NS

#endif
#endif
