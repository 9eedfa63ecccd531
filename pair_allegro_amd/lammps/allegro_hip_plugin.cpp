// LAMMPS plugin packaging of the HIP `pair_style allegro` and `compute allegro[/atom]` (SURVEY 8b "alternative
// packaging"): build this file together with pair_allegro_hip.cpp and compute_allegro_hip.cpp into a shared object linked
// against liballegro_hip.so and load it from a deck with
//     plugin load allegro_hip_plugin.so
// -- no LAMMPS rebuild.  Interface: LAMMPS' src/PLUGIN/lammpsplugin.h (lammpsplugin_t, lammpsplugin_init).
#include "lammpsplugin.h"

#include "compute_allegro_hip.h"
#include "lammps.h"
#include "pair_allegro_hip.h"
#include "version.h"

using namespace LAMMPS_NS;

static Pair *allegro_pair_creator(LAMMPS *lmp) { return new PairAllegroHIP(lmp); }
// `pair_style nequip` (the reference builds it from the same template, pair_nequip_allegro.cpp:86-89,150,535-556): not part of
// this build -- NequIP message passing needs ghost-to-ghost edges and a different model graph.  The style name is registered so
// that a deck asking for it stops with a clear message instead of "Unrecognized pair style".
static Pair *nequip_pair_creator(LAMMPS *lmp)
{
  lmp->error->all(FLERR, "pair_style nequip is not provided by the MI355X-native allegro-hip build: only pair_style allegro "
                         "(strictly local Allegro models) is implemented; run NequIP models with the reference pair style");
  return nullptr;
}
static Compute *allegro_compute_creator(LAMMPS *lmp, int argc, char **argv) { return new ComputeAllegroHIP<0>(lmp, argc, argv); }
static Compute *allegro_atom_compute_creator(LAMMPS *lmp, int argc, char **argv) { return new ComputeAllegroHIP<1>(lmp, argc, argv); }

extern "C" void lammpsplugin_init(void *lmp, void *handle, void *regfunc)
{
  lammpsplugin_t plugin;
  lammpsplugin_regfunc register_plugin = (lammpsplugin_regfunc) regfunc;

  plugin.version = LAMMPS_VERSION;
  plugin.author = "allegro-hip";
  plugin.handle = handle;

  plugin.style = "pair";
  plugin.name = "allegro";
  plugin.info = "Allegro pair style on AMD MI355X (HIP kernels, liballegro_hip.so)";
  plugin.creator.v1 = (lammpsplugin_factory1 *) &allegro_pair_creator;
  (*register_plugin)(&plugin, lmp);

  plugin.name = "nequip";
  plugin.info = "placeholder: stops with a clear message (only pair_style allegro is implemented on the HIP path)";
  plugin.creator.v1 = (lammpsplugin_factory1 *) &nequip_pair_creator;
  (*register_plugin)(&plugin, lmp);

  plugin.style = "compute";
  plugin.name = "allegro";
  plugin.info = "global quantity from the Allegro model's output (HIP pair style)";
  plugin.creator.v2 = (lammpsplugin_factory2 *) &allegro_compute_creator;
  (*register_plugin)(&plugin, lmp);

  plugin.name = "allegro/atom";
  plugin.info = "per-atom quantity from the Allegro model's output (HIP pair style)";
  plugin.creator.v2 = (lammpsplugin_factory2 *) &allegro_atom_compute_creator;
  (*register_plugin)(&plugin, lmp);
}
