/* -*- c++ -*- ----------------------------------------------------------
   `compute allegro` / `compute allegro/atom` for pair_style allegro (HIP): extracts a named entry of the model's
   output dict from PairAllegroHIP after a force evaluation.  Same deck syntax, argument checks and messages as the
   reference compute (/root/reference/compute/compute_allegro.{h,cpp}); the tensors come through the C-ABI
   (ahip_output_register / ahip_output_get, include/allegro_hip.h) instead of torch.

     compute ID all allegro      <quantity> <length>
     compute ID all allegro/atom <quantity> <length> <newton 1/0>
------------------------------------------------------------------------- */

#ifdef COMPUTE_CLASS
// clang-format off
ComputeStyle(allegro,ComputeAllegroHIP<0>)
ComputeStyle(allegro/atom,ComputeAllegroHIP<1>)
// clang-format on
#else

#ifndef LMP_COMPUTE_ALLEGRO_HIP_H
#define LMP_COMPUTE_ALLEGRO_HIP_H

#include "compute.h"

#include <string>
#include <vector>

namespace LAMMPS_NS {

template <int peratom> class ComputeAllegroHIP : public Compute {
 public:
  ComputeAllegroHIP(class LAMMPS *, int, char **);
  ~ComputeAllegroHIP() override;
  void compute_vector() override;
  void compute_peratom() override;
  void init() override;

  int pack_reverse_comm(int, int, double *) override;
  void unpack_reverse_comm(int, int *, double *) override;

 protected:
  std::string quantity;
  std::vector<double> quantity_values;    // the model output of the last evaluation, rows for locals and ghosts
  int newton;
  int nperatom;
  int nmax;
};

}    // namespace LAMMPS_NS

#endif
#endif
