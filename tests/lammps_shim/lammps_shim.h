// TEST INFRASTRUCTURE ONLY: the few LAMMPS declarations pair_allegro_hip.cpp touches,
// just enough to compile the Pair subclass and drive it with the LAMMPS call sequence in tests.
// Not LAMMPS, not shipped, never used to build anything from /root/reference.
#pragma once
#include <mpi_stub.h>

#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

#define FLERR __FILE__, __LINE__
#define NEIGHMASK 0x1FFFFFFF

namespace LAMMPS_NS {
typedef int64_t bigint;
typedef int tagint;

struct LammpsAbort : std::runtime_error { using std::runtime_error::runtime_error; };

class Error {
 public:
  static std::string fmt(const std::string &f, const char *a) {
    std::string s = f; auto p = s.find("{}"); if (p != std::string::npos) s.replace(p, 2, a); return s;
  }
  [[noreturn]] void all(const char *, int, const std::string &m) { throw LammpsAbort(m); }
  [[noreturn]] void all(const char *, int, const std::string &m, const char *a) { throw LammpsAbort(fmt(m, a)); }
  [[noreturn]] void all(const char *, int, const std::string &m, const std::string &a) { throw LammpsAbort(fmt(m, a.c_str())); }
  [[noreturn]] void one(const char *, int, const std::string &m, const char *a) { throw LammpsAbort(fmt(m, a)); }
  void message(const char *, int, const std::string &m) { std::printf("%s\n", m.c_str()); }
};
class Memory {
 public:
  template <typename T> T **create(T **&a, int n1, int n2, const char *) {
    T *data = new T[(size_t)n1 * n2]();
    a = new T *[n1];
    for (int i = 0; i < n1; i++) a[i] = data + (size_t)i * n2;
    return a;
  }
  template <typename T> void destroy(T **&a) { if (a) { delete[] a[0]; delete[] a; a = nullptr; } }
  template <typename T> T *create(T *&a, int n, const char *) { a = new T[n](); return a; }
  template <typename T> void destroy(T *&a) { delete[] a; a = nullptr; }
};
enum ExecutionSpace { Host, Device };
class AtomKokkos; class MemoryKokkos; class KokkosLMP;
class Atom { public: virtual ~Atom() = default; int tag_enable = 1, ntypes = 1, nlocal = 0, nghost = 0, nmax = 0; double **x = nullptr, **f = nullptr; int *type = nullptr; tagint *tag = nullptr; };
class Compute;
class Pair;
class Comm {
 public:
  int me = 0, nprocs = 1;
  // single-rank periodic reverse communication: ghost rows are added to their owner (tag-1 == owner index here)
  tagint *tag = nullptr; int nlocal = 0, nghost = 0;
  inline void reverse_comm(Compute *c);
};
class Force { public: int newton_pair = 1; Pair *pair = nullptr; };
class Update { public: bigint ntimestep = 0; };
class NeighList { public: int inum = 0, gnum = 0; int *ilist = nullptr, *numneigh = nullptr; int **firstneigh = nullptr; };
namespace NeighConst { enum { REQ_FULL = 1, REQ_GHOST = 2 }; }
class NeighRequest { public: int kokkos_host = -1, kokkos_device = -1; void set_kokkos_host(int v) { kokkos_host = v; } void set_kokkos_device(int v) { kokkos_device = v; } };
class Neighbor { public: bigint lastcall = 0; int ago = 0; int requested = 0; NeighRequest request; void add_request(Pair *, int flags) { requested = flags; }
                 NeighRequest *find_request(Pair *) { return &request; } };

class LAMMPS {
 public:
  Atom *atom; Comm *comm; Force *force; Neighbor *neighbor; Error *error; Memory *memory; MPI_Comm world = 0;
  Update *update = nullptr;
  AtomKokkos *atomKK = nullptr; MemoryKokkos *memoryKK = nullptr; KokkosLMP *kokkos = nullptr;     // KOKKOS package (kokkos_shim.h)
};

class Pair {
 public:
  explicit Pair(LAMMPS *l) : lmp(l), atom(l->atom), comm(l->comm), force(l->force), neighbor(l->neighbor), error(l->error),
                             memory(l->memory), world(l->world), atomKK(l->atomKK), memoryKK(l->memoryKK) {}
  virtual ~Pair() = default;
  virtual void compute(int, int) = 0;
  virtual void settings(int, char **) = 0;
  virtual void coeff(int, char **) = 0;
  virtual double init_one(int, int) { return 0; }
  virtual void init_style() {}
  // state the subclass reads / writes
  int restartinfo = 1, manybody_flag = 0, no_virial_fdotr_compute = 0, allocated = 0, copymode = 0;
  int **setflag = nullptr; double **cutsq = nullptr;
  NeighList *list = nullptr;
  double eng_vdwl = 0, virial[6] = {0, 0, 0, 0, 0, 0};
  double *eatom = nullptr;
  int eflag_atom = 0, vflag_atom = 0, eflag_global = 0, vflag_global = 0, vflag_fdotr = 0;
  int respa_enable = 1, kokkosable = 0, maxeatom = 0;
  ExecutionSpace execution_space = Host; unsigned int datamask_read = 0, datamask_modify = 0;
  void ev_init(int eflag, int vflag, int /*alloc*/ = 1) {            // the part of Pair::ev_setup the subclass relies on
    if ((eflag & 2) && atom->nmax > maxeatom) maxeatom = atom->nmax;
    eflag_global = eflag & 1; eflag_atom = (eflag & 2) ? 1 : 0; vflag_global = vflag & 3; vflag_atom = (vflag & 4) ? 1 : 0;
    eng_vdwl = 0; for (double &v : virial) v = 0;
  }
 protected:
  LAMMPS *lmp; Atom *atom; Comm *comm; Force *force; Neighbor *neighbor; Error *error; Memory *memory; MPI_Comm world;
  AtomKokkos *atomKK; MemoryKokkos *memoryKK;
};

class Compute {
 public:
  Compute(LAMMPS *l, int, char **) : lmp(l), atom(l->atom), comm(l->comm), force(l->force), error(l->error), memory(l->memory),
                                     update(l->update), world(l->world) {}
  virtual ~Compute() = default;
  virtual void init() = 0;
  virtual void compute_vector() {}
  virtual void compute_peratom() {}
  virtual int pack_reverse_comm(int, int, double *) { return 0; }
  virtual void unpack_reverse_comm(int, int *, double *) {}
  int peratom_flag = 0, vector_flag = 0, extvector = 0, size_vector = 0, size_peratom_cols = 0, comm_reverse = 0, copymode = 0;
  bigint invoked_vector = -1, invoked_peratom = -1;
  double *vector = nullptr, *vector_atom = nullptr, **array_atom = nullptr;
 protected:
  LAMMPS *lmp; Atom *atom; Comm *comm; Force *force; Error *error; Memory *memory; Update *update; MPI_Comm world;
};
inline void Comm::reverse_comm(Compute *c) {
  std::vector<double> buf((size_t)nghost * c->comm_reverse);
  c->pack_reverse_comm(nghost, nlocal, buf.data());
  std::vector<int> owners(nghost);
  for (int k = 0; k < nghost; k++) owners[k] = tag[nlocal + k] - 1;
  c->unpack_reverse_comm(nghost, owners.data(), buf.data());
}
}    // namespace LAMMPS_NS
