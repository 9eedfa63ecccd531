// TEST INFRASTRUCTURE ONLY: drives PairAllegroHIP through the LAMMPS call sequence
// (settings -> coeff -> init_style -> init_one -> compute) on a system read from a flat binary file
// written by tests/test_lammps_cpp.py, and writes forces / energy / virial back.
#include "pair_allegro_hip.h"
#include "compute_allegro_hip.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

using namespace LAMMPS_NS;

template <typename T> static std::vector<T> rd(FILE *f, size_t n) { std::vector<T> v(n); if (n && fread(v.data(), sizeof(T), n, f) != n) { perror("read"); exit(3); } return v; }

// `plugin load` path: lammpsplugin_init must register pair allegro, compute allegro, compute allegro/atom
#include "lammpsplugin.h"
#include <string>
static std::vector<std::string> g_registered;
static lammpsplugin_factory1 *g_pair_factory = nullptr, *g_nequip_factory = nullptr;
static void collect_plugin(lammpsplugin_t *p, void *) {
  g_registered.push_back(std::string(p->style) + ":" + p->name);
  if (std::string(p->style) == "pair" && std::string(p->name) == "allegro") g_pair_factory = p->creator.v1;
  if (std::string(p->style) == "pair" && std::string(p->name) == "nequip") g_nequip_factory = p->creator.v1;
}

int main(int argc, char **argv) {
  if (argc == 2 && std::string(argv[1]) == "--plugin") {
    Atom atom; Comm comm; Force force; Neighbor neighbor; Error error; Memory memory; Update update;
    LAMMPS lmp{&atom, &comm, &force, &neighbor, &error, &memory};
    lmp.update = &update;
    lammpsplugin_init(&lmp, nullptr, (void *) &collect_plugin);
    for (auto &r : g_registered) printf("registered %s\n", r.c_str());
    Pair *p = g_pair_factory ? (Pair *) g_pair_factory(&lmp) : nullptr;      // the factory builds a working pair style object
    printf("pair object %s restartinfo=%d manybody=%d\n", p ? "ok" : "null", p ? p->restartinfo : -1, p ? p->manybody_flag : -1);
    delete p;
    try { if (g_nequip_factory) g_nequip_factory(&lmp); printf("pair_style nequip: no error raised\n"); }
    catch (const LammpsAbort &e) { printf("pair_style nequip -> error->all: %s\n", e.what()); }
    return g_registered.size() == 4 && p ? 0 : 1;
  }
  if (argc < 5) { fprintf(stderr, "usage: driver system.bin out.bin model names...\n"); return 2; }
  FILE *f = fopen(argv[1], "rb");
  int hdr[4];
  if (!f || fread(hdr, sizeof(int), 4, f) != 4) return 3;
  const int nlocal = hdr[0], nghost = hdr[1], ntypes = hdr[2], nneigh = hdr[3], nall = nlocal + nghost;
  auto x = rd<double>(f, (size_t)nall * 3);
  auto type = rd<int>(f, nall);
  auto tag = rd<int>(f, nall);
  auto numneigh = rd<int>(f, nall);
  auto flat = rd<int>(f, nneigh);
  fclose(f);
  std::vector<double> fr((size_t)nall * 3, 0.0), eatom(nall, 0.0);
  std::vector<double *> xp(nall), fp(nall);
  std::vector<int *> first(nall);
  std::vector<int> ilist(nlocal);
  size_t off = 0;
  for (int i = 0; i < nall; i++) { xp[i] = &x[3 * (size_t)i]; fp[i] = &fr[3 * (size_t)i]; first[i] = flat.data() + off; off += numneigh[i]; }
  for (int i = 0; i < nlocal; i++) ilist[i] = i;

  Atom atom; Comm comm; Force force; Neighbor neighbor; Error error; Memory memory; NeighList list; Update update;
  LAMMPS lmp{&atom, &comm, &force, &neighbor, &error, &memory};
  lmp.update = &update;
  atom.nmax = nall; comm.tag = tag.data(); comm.nlocal = nlocal; comm.nghost = nghost;
  atom.ntypes = ntypes; atom.nlocal = nlocal; atom.nghost = nghost; atom.x = xp.data(); atom.f = fp.data(); atom.type = type.data(); atom.tag = tag.data();
  list.inum = nlocal; list.gnum = nghost; list.ilist = ilist.data(); list.numneigh = numneigh.data(); list.firstneigh = first.data();
  int rc = 0;
  try {
    PairAllegroHIP pair(&lmp);
    pair.list = &list;
    pair.eatom = eatom.data();
    pair.settings(0, nullptr);
    std::vector<char *> args;
    char star[] = "*";
    args.push_back(star); args.push_back(star);
    for (int k = 3; k < argc; k++) args.push_back(argv[k]);
    pair.coeff((int)args.size(), args.data());
    force.pair = &pair;
    // `compute allegro` family, defined after the pair style like in a deck (only when the test asks: env DRIVER_COMPUTES)
    const bool with_computes = std::getenv("DRIVER_COMPUTES") != nullptr;
    char c_id[] = "c", c_all[] = "all", c_v[] = "allegro", c_a[] = "allegro/atom", q_vir[] = "virial", q_f[] = "forces", q_e[] = "atomic_energy",
         n9[] = "9", n3[] = "3", n1[] = "1", n0[] = "0";
    char *av[] = {c_id, c_all, c_v, q_vir, n9}, *af[] = {c_id, c_all, c_a, q_f, n3, n1}, *ae[] = {c_id, c_all, c_a, q_e, n1, n0};
    ComputeAllegroHIP<0> *cvir = with_computes ? new ComputeAllegroHIP<0>(&lmp, 5, av) : nullptr;
    ComputeAllegroHIP<1> *cfor = with_computes ? new ComputeAllegroHIP<1>(&lmp, 6, af) : nullptr;
    ComputeAllegroHIP<1> *cen = with_computes ? new ComputeAllegroHIP<1>(&lmp, 6, ae) : nullptr;
    pair.init_style();
    if (neighbor.requested != (NeighConst::REQ_FULL | NeighConst::REQ_GHOST)) { fprintf(stderr, "bad neighbor request\n"); return 4; }
    const double cut = pair.init_one(1, 1);
    neighbor.lastcall = 1;
    if (std::getenv("DRIVER_REBUILD_SAME_STEP")) {
      // `run 0` -> atoms displaced, list rebuilt with a different row order -> `run 0`, all at ONE timestep.
      // First "run" on a scrambled copy (positions shifted, rows reversed); the second run must not reuse that copy.
      std::vector<double> xs(x); std::vector<int> fl(flat);
      size_t o2 = 0;
      for (int i = 0; i < nall; i++) { std::reverse(fl.begin() + o2, fl.begin() + o2 + numneigh[i]); o2 += numneigh[i]; }
      for (int i = 0; i < nall; i++) { xs[3 * (size_t)i] += (i < nlocal) ? 0.05 * ((i * 7) % 5 - 2) : 0.0; }
      std::vector<double *> xsp(nall); std::vector<int *> fsp(nall);
      o2 = 0;
      for (int i = 0; i < nall; i++) { xsp[i] = &xs[3 * (size_t)i]; fsp[i] = fl.data() + o2; o2 += numneigh[i]; }
      std::vector<int> nn2(numneigh);
      for (int i = 0; i < nlocal; i++) nn2[i] = numneigh[i] / 2;   // ... and with other atoms in range (half of each row)
      atom.x = xsp.data(); list.firstneigh = fsp.data(); list.numneigh = nn2.data();
      neighbor.ago = 0;
      pair.compute(3, 2);
      std::fill(fr.begin(), fr.end(), 0.0);
      atom.x = xp.data(); list.firstneigh = first.data(); list.numneigh = numneigh.data();   // second run setup: same timestep, fresh list
      pair.init_style();
      neighbor.ago = 0;
    }
    pair.compute(3, 2);                     // eflag global+atom, vflag global
    neighbor.ago = 1;
    pair.compute(3, 2);                     // second step on the same list: forces must be ADDED again
    FILE *o = fopen(argv[2], "wb");
    fwrite(&cut, sizeof(double), 1, o);
    fwrite(&pair.eng_vdwl, sizeof(double), 1, o);
    fwrite(pair.virial, sizeof(double), 6, o);
    fwrite(fr.data(), sizeof(double), fr.size(), o);
    fwrite(eatom.data(), sizeof(double), eatom.size(), o);
    if (with_computes) {
      cvir->compute_vector(); cfor->compute_peratom(); cen->compute_peratom();
      fwrite(cvir->vector, sizeof(double), 9, o);
      for (int i = 0; i < nlocal; i++) fwrite(cfor->array_atom[i], sizeof(double), 3, o);      // ghost rows folded in (newton 1)
      fwrite(cen->vector_atom, sizeof(double), nlocal, o);
      delete cvir; delete cfor; delete cen;
    }
    fclose(o);
    printf("restartinfo=%d manybody=%d no_fdotr=%d setflag11=%d\n", pair.restartinfo, pair.manybody_flag, pair.no_virial_fdotr_compute, pair.setflag[1][1]);
  } catch (const LammpsAbort &e) { printf("LAMMPS error->all: %s\n", e.what()); rc = 10; }
  catch (const std::exception &e) { printf("exception: %s\n", e.what()); rc = 11; }
  return rc;
}
