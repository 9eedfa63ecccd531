// TEST INFRASTRUCTURE ONLY: drives PairAllegroHIPKokkos (pair_style allegro/kk) through the LAMMPS call sequence on a system
// read from the flat binary file of tests/test_lammps_cpp.py, with the KOKKOS package's data structures stood in by
// kokkos_shim.h: dual views for x / f / type / tag, a column-major padded neighbor table with special-bond bits set on some
// entries, atomKK->sync / modified masks.  Same output file as driver.cpp.
#include "pair_allegro_hip_kokkos.h"
#include "compute_allegro_hip.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using namespace LAMMPS_NS;

template <typename T> static std::vector<T> rd(FILE *f, size_t n) { std::vector<T> v(n); if (n && fread(v.data(), sizeof(T), n, f) != n) { perror("read"); exit(3); } return v; }

int main(int argc, char **argv) {
  if (argc < 5) { fprintf(stderr, "usage: driver_kk system.bin out.bin model names...\n"); return 2; }
  FILE *fi = fopen(argv[1], "rb");
  int hdr[4];
  if (!fi || fread(hdr, sizeof(int), 4, fi) != 4) return 3;
  const int nlocal = hdr[0], nghost = hdr[1], ntypes = hdr[2], nneigh = hdr[3], nall = nlocal + nghost;
  auto x = rd<double>(fi, (size_t)nall * 3);
  auto type = rd<int>(fi, nall);
  auto tag = rd<int>(fi, nall);
  auto numneigh = rd<int>(fi, nall);
  auto flat = rd<int>(fi, nneigh);
  fclose(fi);

  AtomKokkos atom; Comm comm; Force force; Neighbor neighbor; Error error; MemoryKokkos memory; Update update; KokkosLMP kokkos;
  LAMMPS lmp{&atom, &comm, &force, &neighbor, &error, &memory};
  lmp.update = &update; lmp.atomKK = &atom; lmp.memoryKK = &memory; lmp.kokkos = &kokkos;
  if (std::getenv("DRIVER_KK_NEIGH_FULL")) kokkos.neighflag = FULL;
  atom.nmax = nall + 7; atom.ntypes = ntypes; atom.nlocal = nlocal; atom.nghost = nghost;
  comm.tag = tag.data(); comm.nlocal = nlocal; comm.nghost = nghost;
  // per-atom arrays live in dual views of nmax rows; the host side is filled, the pair style must sync what it reads
  atom.k_x = DAT::tdual_x_array("atom:x", atom.nmax); atom.k_f = DAT::tdual_f_array("atom:f", atom.nmax);
  atom.k_type = DAT::tdual_int_1d("atom:type", atom.nmax); atom.k_tag = DAT::tdual_int_1d("atom:tag", atom.nmax);
  for (int i = 0; i < nall; i++) {
    for (int d = 0; d < 3; d++) { atom.k_x.h_view(i, d) = x[3 * (size_t)i + d]; atom.k_f.h_view(i, d) = 0.0; }
    atom.k_type.h_view(i) = type[i]; atom.k_tag.h_view(i) = tag[i];
  }
  atom.modified(Host, X_MASK | F_MASK | TYPE_MASK | TAG_MASK);

  // device neighbor table: rows for the local atoms only, padded to maxneighs (+3), column-major; bit 30 (a special-bond flag)
  // set on every fifth entry, stale garbage beyond numneigh
  NeighListKokkos<LMPDeviceType> list;
  int maxn = 1;
  for (int i = 0; i < nlocal; i++) maxn = std::max(maxn, numneigh[i]);
  maxn += 3;
  DAT::tdual_neighbors_2d k_nb("neigh:table", nlocal, maxn);
  DAT::tdual_int_1d k_il("neigh:ilist", nlocal), k_nn("neigh:numneigh", nall);
  size_t off = 0;
  for (int i = 0; i < nall; i++) {
    if (i < nlocal) {
      for (int jj = 0; jj < maxn; jj++) k_nb.h_view(i, jj) = jj < numneigh[i] ? (flat[off + jj] | ((off + jj) % 5 == 0 ? (1 << 30) : 0)) : 0x12345678;
      k_il.h_view(i) = i;
    }
    k_nn.h_view(i) = i < nlocal ? numneigh[i] : 0;
    off += numneigh[i];
  }
  k_nb.modify<LMPHostType>(); k_il.modify<LMPHostType>(); k_nn.modify<LMPHostType>();
  k_nb.sync<LMPDeviceType>(); k_il.sync<LMPDeviceType>(); k_nn.sync<LMPDeviceType>();
  list.inum = nlocal; list.gnum = nghost;
  list.d_neighbors = k_nb.d_view; list.d_ilist = k_il.d_view; list.d_numneigh = k_nn.d_view;

  int rc = 0;
  try {
    PairAllegroHIPKokkos pair(&lmp);
    pair.list = &list;
    pair.settings(0, nullptr);
    std::vector<char *> args;
    char star[] = "*";
    args.push_back(star); args.push_back(star);
    for (int k = 3; k < argc; k++) args.push_back(argv[k]);
    pair.coeff((int)args.size(), args.data());
    force.pair = &pair;
    // `compute allegro/atom forces 3 1` and `compute allegro virial 9` next to the /kk pair style (env DRIVER_COMPUTES)
    const bool with_computes = std::getenv("DRIVER_COMPUTES") != nullptr;
    char c_id[] = "c", c_all[] = "all", c_v[] = "allegro", c_a[] = "allegro/atom", q_vir[] = "virial", q_f[] = "forces", n9[] = "9", n3[] = "3", n1[] = "1";
    char *av[] = {c_id, c_all, c_v, q_vir, n9}, *af[] = {c_id, c_all, c_a, q_f, n3, n1};
    ComputeAllegroHIP<0> *cvir = with_computes ? new ComputeAllegroHIP<0>(&lmp, 5, av) : nullptr;
    ComputeAllegroHIP<1> *cfor = with_computes ? new ComputeAllegroHIP<1>(&lmp, 6, af) : nullptr;
    pair.init_style();
    if (neighbor.requested != (NeighConst::REQ_FULL | NeighConst::REQ_GHOST) || neighbor.request.kokkos_device != 1 || neighbor.request.kokkos_host != 0) {
      fprintf(stderr, "bad neighbor request\n"); return 4;
    }
    const double cut = pair.init_one(1, 1);
    neighbor.ago = 0;
    pair.compute(3, 2);                     // eflag global+atom, vflag global; list built this step
    const double eng1 = pair.eng_vdwl;
    double vir1[6];
    for (int k = 0; k < 6; k++) vir1[k] = pair.virial[k];
    neighbor.ago = 1;
    pair.compute(1, 0);                     // second step on the same list, energy only: forces must be ADDED again
    if (!(atom.synced_to_device & X_MASK) || !(atom.synced_to_device & TYPE_MASK) || !(atom.modified_on_device & F_MASK)) { fprintf(stderr, "sync/modified masks not set\n"); return 5; }
    atom.sync(Host, F_MASK);                // what a host-side fix would do before reading f
    std::vector<double> fr((size_t)nall * 3), ea(nall, 0.0);
    for (int i = 0; i < nall; i++) for (int d = 0; d < 3; d++) fr[3 * (size_t)i + d] = atom.k_f.h_view(i, d);
    // eatom of the FIRST call was synced to the host by the class; the second call had eflag_atom = 0 and must not touch it
    for (int i = 0; i < nlocal; i++) ea[i] = pair.eatom ? pair.eatom[i] : 0.0;
    FILE *o = fopen(argv[2], "wb");
    fwrite(&cut, sizeof(double), 1, o);
    fwrite(&eng1, sizeof(double), 1, o);
    fwrite(vir1, sizeof(double), 6, o);
    fwrite(fr.data(), sizeof(double), fr.size(), o);
    fwrite(ea.data(), sizeof(double), ea.size(), o);
    fwrite(&pair.eng_vdwl, sizeof(double), 1, o);
    if (with_computes) {
      cfor->compute_peratom();                  // forces of the LAST call, ghost rows folded in (newton 1)
      for (int i = 0; i < nlocal; i++) fwrite(cfor->array_atom[i], sizeof(double), 3, o);
      cvir->compute_vector();
      fwrite(cvir->vector, sizeof(double), 9, o);
      delete cvir; delete cfor;
    }
    fclose(o);
    printf("restartinfo=%d manybody=%d no_fdotr=%d respa=%d kokkosable=%d\n", pair.restartinfo, pair.manybody_flag, pair.no_virial_fdotr_compute, pair.respa_enable, pair.kokkosable);
  } catch (const LammpsAbort &e) { printf("LAMMPS error->all: %s\n", e.what()); rc = 10; }
  catch (const std::exception &e) { printf("exception: %s\n", e.what()); rc = 11; }
  return rc;
}
