/* ----------------------------------------------------------------------
   pair_style allegro over liballegro_hip.  Host code only: every per-step quantity crosses the
   C-ABI of include/allegro_hip.h as plain pointers.

   Line references are to the reference implementation this file replaces,
   mir-group/pair_allegro pair_nequip_allegro.cpp.
------------------------------------------------------------------------- */

#include "pair_allegro_hip.h"

#include "allegro_hip.h"

#include "atom.h"
#include "comm.h"
#include "error.h"
#include "force.h"
#include "memory.h"
#include "neigh_list.h"
#include "neighbor.h"
#include "update.h"

#include <cstdlib>
#include <cstring>
#include <iostream>
#include <sstream>
#include <stdexcept>

#include <mpi.h>

using namespace LAMMPS_NS;

PairAllegroHIP::PairAllegroHIP(LAMMPS *lmp) : Pair(lmp)
{
  restartinfo = 0;      // :68
  manybody_flag = 1;    // :69
  // the model owns the virial (do not let fdotr overwrite it)
  no_virial_fdotr_compute = 1;

  if (comm->me == 0)
    std::cout << "Allegro (HIP) is using input precision d and output precision d" << std::endl;

  if (const char *env_p = std::getenv("_NEQUIP_LOG_LEVEL")) {    // :78-83
    if (std::string(env_p) == "DEBUG") {
      std::cout << "Debug mode enabled, since _NEQUIP_LOG_LEVEL is set to DEBUG\n";
      debug_mode = 1;
    }
  }

  // === device = node-local rank (:92-120) ===
  int devicecount = 0;
  if (ahip_device_count(&devicecount) != AHIP_OK || devicecount <= 0)
    error->all(FLERR, "pair_allegro (HIP): no GPU visible; this pair style has no CPU path");
  int deviceidx = 0;
  if (comm->nprocs > 1) {
    MPI_Comm shmcomm;
    MPI_Comm_split_type(world, MPI_COMM_TYPE_SHARED, 0, MPI_INFO_NULL, &shmcomm);
    int shmrank;
    MPI_Comm_rank(shmcomm, &shmrank);
    MPI_Comm_free(&shmcomm);
    deviceidx = shmrank;
    if (deviceidx >= devicecount) {
      if (debug_mode) {    // :104-110
        std::cerr << "WARNING (Allegro): my rank (" << deviceidx << ") is bigger than the number of visible devices ("
                  << devicecount << "), wrapping around to use device " << deviceidx % devicecount << " again!!!";
        deviceidx = deviceidx % devicecount;
      } else {             // :112-117
        std::cerr << "ERROR (Allegro): my rank (" << deviceidx << ") is bigger than the number of visible devices ("
                  << devicecount << ")!!!";
        error->all(FLERR, "pair_allegro: mismatch between number of ranks and number of available GPUs");
      }
    }
  }
  device = deviceidx;
  if (debug_mode) std::cout << "Allegro (HIP) is using device " << device << "\n";
}

PairAllegroHIP::~PairAllegroHIP()
{
  if (copymode) return;
  if (model) ahip_model_free(model);
  if (allocated) {
    memory->destroy(setflag);
    memory->destroy(cutsq);
    memory->destroy(cutoff_matrix);
  }
}

void PairAllegroHIP::init_style()
{
  if (atom->tag_enable == 0) error->all(FLERR, "Pair style Allegro requires atom IDs");    // :139
  neighbor->add_request(this, NeighConst::REQ_FULL | NeighConst::REQ_GHOST);               // :146
  if (force->newton_pair == 0) error->all(FLERR, "Pair style allegro requires newton pair on");    // :149
  last_list_build = -1;    // a new run: whatever list the library holds is stale (atoms may have been displaced between runs)
}

double PairAllegroHIP::init_one(int /*i*/, int /*j*/) { return cutoff; }    // :153-156

void PairAllegroHIP::allocate()
{
  allocated = 1;
  int n = atom->ntypes;
  memory->create(setflag, n + 1, n + 1, "pair:setflag");
  memory->create(cutsq, n + 1, n + 1, "pair:cutsq");
  memory->create(cutoff_matrix, n, n, "pair:cutoff_matrix");
}

void PairAllegroHIP::settings(int narg, char ** /*arg*/)
{
  if (narg > 0) error->all(FLERR, "Illegal pair_style command, too many arguments");    // :171
}

void PairAllegroHIP::coeff(int narg, char **arg)
{
  if (!allocated) allocate();
  int ntypes = atom->ntypes;
  for (int i = 1; i <= ntypes; i++)
    for (int j = i; j <= ntypes; j++) setflag[i][j] = 0;

  if (narg != (3 + ntypes))    // :185-188
    error->all(FLERR,
               "Incorrect args for pair coefficients, should be * * <model>.nequip.pth/pt2 <type1> <type2> ... <typen>");
  if (strcmp(arg[0], "*") != 0 || strcmp(arg[1], "*") != 0)    // :191-192
    error->all(FLERR, "Incorrect args for pair coefficients");

  model_path = std::string(arg[2]);
  if (comm->me == 0) std::cout << "NequIP/Allegro: Loading model from " << model_path << "\n";
  if (model) { ahip_model_free(model); model = nullptr; }
  int rc = ahip_model_load(model_path.c_str(), device, &model);
  if (rc == AHIP_ERR_FILE) throw std::runtime_error(ahip_last_error());    // the reference throws here too (:205)
  if (rc != AHIP_OK) error->all(FLERR, "pair_allegro: {}", ahip_last_error());
  for (const std::string &name : custom_output_names)       // computes defined before pair_coeff
    if (ahip_output_register(model, name.c_str()) != AHIP_OK) error->all(FLERR, "pair_allegro: {}", ahip_last_error());

  double r_max;
  int num_model_types;
  const char *type_names;
  const double *pc;
  ahip_model_meta(model, &r_max, &num_model_types, &type_names, &pc, nullptr, nullptr, nullptr, nullptr, nullptr);
  {
    // :267-270: the reference hands the key to at::globalContext().setAllowTF32CuBLAS / CuDNN; the library selects its reduced-split
    // matrix arithmetic from it (ahip_model_allow_tf32)
    int allow_tf32 = 0;
    ahip_model_allow_tf32(model, &allow_tf32);
    if (comm->me == 0 && (allow_tf32 || debug_mode)) std::cout << "NequIP/Allegro: model metadata allow_tf32 = " << allow_tf32 << "\n";
  }
  cutoff = r_max;    // :272

  type_mapper.assign(ntypes, -1);    // :274
  std::stringstream ss;
  ss << type_names;
  if (comm->me == 0) std::cout << "Type mapping:\nNequIP/Allegro type | NequIP/Allegro name | LAMMPS type | LAMMPS name\n";
  for (int i = 0; i < num_model_types; i++) {    // :284-294
    std::string ele;
    ss >> ele;
    for (int itype = 1; itype <= ntypes; itype++) {
      if (ele.compare(arg[itype + 3 - 1]) == 0) {
        type_mapper[itype - 1] = i;
        if (comm->me == 0) std::cout << i << " | " << ele << " | " << itype << " | " << arg[itype + 3 - 1] << "\n";
      }
    }
  }
  for (int i = 1; i <= ntypes; i++)    // :297-301
    for (int j = i; j <= ntypes; j++)
      if ((type_mapper[i - 1] >= 0) && (type_mapper[j - 1] >= 0)) setflag[i][j] = 1;

  // per-edge-type cutoffs in LAMMPS type index (:303-328).  Every LAMMPS type that maps to a model
  // type inherits that model type's row (the reference's reverse map keeps only the last one, App. D).
  for (int i = 0; i < ntypes; i++)
    for (int j = 0; j < ntypes; j++) {
      double c = cutoff;
      if (pc && type_mapper[i] >= 0 && type_mapper[j] >= 0) c = pc[type_mapper[i] * num_model_types + type_mapper[j]];
      cutoff_matrix[i][j] = c;
    }
  last_list_build = -1;
}

void PairAllegroHIP::compute(int eflag, int vflag)
{
  ev_init(eflag, vflag);
  if (vflag_atom) error->all(FLERR, "Pair styles nequip and allegro do not support per-atom virial");    // :394

  int inum = list->inum;
  if (inum == 0) return;    // empty sub-domain (:340-341)
  int nlocal = atom->nlocal;
  int nghost = atom->nghost;
  int ntotal = nlocal + nghost;

  // Hand the list over only on steps where LAMMPS (re)built it (replaces the per-step list walk of :488-512,:566-629).
  // neighbor->ago == 0 is LAMMPS' own "the lists were built this step" (Neighbor::build resets it; every run setup
  // rebuilds the list WITHOUT advancing the timestep, so the timestep of the last build cannot be the key);
  // init_style() runs at every run init and drops the installed copy.
  if (neighbor->ago == 0 || last_list_build < 0) {
    if (ahip_neigh_update(model, inum, ntotal, list->ilist, list->numneigh, list->firstneigh, NEIGHMASK) != AHIP_OK)
      error->one(FLERR, "pair_allegro: {}", ahip_last_error());
    last_list_build = 1;
  }

  double eng = 0.0;
  double vir[6];
  int rc = ahip_compute(model, nlocal, nghost, &atom->x[0][0], atom->type, atom->ntypes, type_mapper.data(),
                        &cutoff_matrix[0][0], &atom->f[0][0], eflag_atom ? eatom : nullptr, &eng, vflag ? vir : nullptr);
  if (rc != AHIP_OK) error->one(FLERR, "pair_allegro: {}", ahip_last_error());

  eng_vdwl = eng;                                         // sum over local atoms only (:366-380)
  if (vflag) for (int k = 0; k < 6; k++) virial[k] = vir[k];    // xx,yy,zz,xy,xz,yz, no sign change (:387-392)

  if (!arith_note_printed) {                              // once: what the library's default arithmetic decided for this model (f16x2 kept, or the float32 instance and why)
    arith_note_printed = true;
    const char *note = ahip_arith_note(model);
    if (comm->me == 0 && note && *note) std::cout << "NequIP/Allegro: " << note << "\n";
  }
  if (debug_mode) ahip_debug_dump_edges(model, atom->tag);    // "Allegro edges: i j rij" (:562-565,620-633)
}

void PairAllegroHIP::add_custom_output(std::string name)    // :681-684
{
  custom_output_names.push_back(name);
  if (model && ahip_output_register(model, name.c_str()) != AHIP_OK) error->all(FLERR, "pair_allegro: {}", ahip_last_error());
}

std::vector<double> PairAllegroHIP::custom_output(const std::string &name)
{
  long long n = 0;
  if (ahip_output_get(model, name.c_str(), nullptr, 0, &n) != AHIP_OK) error->one(FLERR, "pair_allegro: {}", ahip_last_error());
  std::vector<double> v((size_t)n);
  if (n > 0 && ahip_output_get(model, name.c_str(), v.data(), n, &n) != AHIP_OK) error->one(FLERR, "pair_allegro: {}", ahip_last_error());
  return v;
}
