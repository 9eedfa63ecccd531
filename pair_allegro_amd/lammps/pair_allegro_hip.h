/* -*- c++ -*- ----------------------------------------------------------
   pair_style allegro -- MI355X-native implementation over liballegro_hip (C-ABI, include/allegro_hip.h).

   Drop-in for `pair_style allegro` of mir-group/pair_allegro: same command surface
   (`pair_style allegro`, `pair_coeff * * <model>.nequip.pth <type names...>`), same neighbor request
   (full + ghost), same newton/atom-ID requirements, same outputs.  The class only marshals LAMMPS
   pointers into the library; there is no libtorch and no Kokkos dependency.
   Reference interface replaced: /root/reference/pair_nequip_allegro.h:43-50,80-82.
------------------------------------------------------------------------- */

#ifdef PAIR_CLASS
// clang-format off
PairStyle(allegro,PairAllegroHIP)
// clang-format on
#else

#ifndef LMP_PAIR_ALLEGRO_HIP_H
#define LMP_PAIR_ALLEGRO_HIP_H

#include "pair.h"

#include <map>
#include <string>
#include <vector>

struct ahip_model;

namespace LAMMPS_NS {

class PairAllegroHIP : public Pair {
 public:
  PairAllegroHIP(class LAMMPS *);
  ~PairAllegroHIP() override;
  void compute(int, int) override;
  void settings(int, char **) override;
  void coeff(int, char **) override;
  double init_one(int, int) override;
  void init_style() override;
  void allocate();

  double cutoff;
  int device = 0;
  std::vector<int> type_mapper;    // LAMMPS type-1 -> model type, -1 = unmapped
  std::string model_path;

  // `compute allegro` hooks of the reference (pair_nequip_allegro.h:80-82): names registered here are kept from the
  // model's output dict at every compute(); custom_output(name) is the reference's custom_output.at(name).cpu().ravel().
  std::vector<std::string> custom_output_names;
  void add_custom_output(std::string);
  std::vector<double> custom_output(const std::string &name);

 protected:
  int debug_mode = 0;
  double **cutoff_matrix = nullptr;    // [ntypes][ntypes], LAMMPS type index
  ahip_model *model = nullptr;
  bigint last_list_build = -1;         // < 0: the library holds no valid copy of the list (set by init_style at every run init)
  bool arith_note_printed = false;     // ahip_arith_note is reported once, after the first evaluation
};

}    // namespace LAMMPS_NS

#endif
#endif
