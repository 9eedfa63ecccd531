// Stand-alone driver helpers: GPU cell-list full neighbor list (the list LAMMPS would hand to the
// pair style: REQ_FULL | REQ_GHOST request of /root/reference/pair_nequip_allegro.cpp:143-147,
// skin-inflated, centres = local atoms, neighbours = locals + ghosts) and the NVE half-steps of the
// test deck's `fix nve` (/root/reference/tests/test_python_repro_allegro.py:84-120).
#include <algorithm>
#include <cmath>

#include "engine.h"
#include "prims.h"

namespace ahip {

struct NbState {
  DevBuf bin_of, bin_cnt, bin_start, bin_fill, sorted, cnt, off, nlj, ilist, box;
  long long cap_j = 0;
};

struct BinGrid {
  double lo[3];
  double inv[3];
  int nb[3];
};

__device__ inline int bin_coord(double x, double lo, double inv, int nb) {
  int b = (int)floor((x - lo) * inv);
  return b < 0 ? 0 : (b >= nb ? nb - 1 : b);
}

__global__ void k_bin_count(int nall, const double *x, BinGrid g, int *bin_of, int *bin_cnt) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nall) return;
  int bx = bin_coord(x[3 * i], g.lo[0], g.inv[0], g.nb[0]);
  int by = bin_coord(x[3 * i + 1], g.lo[1], g.inv[1], g.nb[1]);
  int bz = bin_coord(x[3 * i + 2], g.lo[2], g.inv[2], g.nb[2]);
  int b = (bz * g.nb[1] + by) * g.nb[0] + bx;
  bin_of[i] = b;
  atomicAdd(&bin_cnt[b], 1);
}

__global__ void k_bin_fill(int nall, const int *bin_of, const int *bin_start, int *bin_fill, int *sorted) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nall) return;
  int b = bin_of[i];
  int slot = atomicAdd(&bin_fill[b], 1);
  sorted[bin_start[b] + slot] = (int)i;
}

// make the within-bin order deterministic (ascending atom index): insertion sort, bins are small
__global__ void k_bin_sort(int nbins, const int *bin_start, int *sorted) {
  long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nbins) return;
  int s = bin_start[b], e = bin_start[b + 1];
  for (int p = s + 1; p < e; ++p) {
    int v = sorted[p], q = p - 1;
    while (q >= s && sorted[q] > v) { sorted[q + 1] = sorted[q]; --q; }
    sorted[q + 1] = v;
  }
}

template <bool FILL>
__global__ void k_neigh_pass(int nlocal, const double *x, BinGrid g, const int *bin_start, const int *sorted,
                             double rcsq, int *cnt, const int *off, int *nlj) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nlocal) return;
  double xi = x[3 * i], yi = x[3 * i + 1], zi = x[3 * i + 2];
  int bx = bin_coord(xi, g.lo[0], g.inv[0], g.nb[0]);
  int by = bin_coord(yi, g.lo[1], g.inv[1], g.nb[1]);
  int bz = bin_coord(zi, g.lo[2], g.inv[2], g.nb[2]);
  int c = 0;
  long long w = FILL ? off[i] : 0;
  for (int dz = -1; dz <= 1; ++dz) {
    int z = bz + dz;
    if (z < 0 || z >= g.nb[2]) continue;
    for (int dy = -1; dy <= 1; ++dy) {
      int y = by + dy;
      if (y < 0 || y >= g.nb[1]) continue;
      int x0 = bx > 0 ? bx - 1 : 0, x1 = bx + 1 < g.nb[0] ? bx + 1 : g.nb[0] - 1;
      int rowbase = (z * g.nb[1] + y) * g.nb[0];
      int s = bin_start[rowbase + x0], e = bin_start[rowbase + x1 + 1];     // x-adjacent bins are contiguous
      for (int p = s; p < e; ++p) {
        int j = sorted[p];
        if (j == (int)i) continue;
        double ddx = x[3 * (long long)j] - xi, ddy = x[3 * (long long)j + 1] - yi, ddz = x[3 * (long long)j + 2] - zi;
        if (ddx * ddx + ddy * ddy + ddz * ddz <= rcsq) {
          if (FILL) nlj[w++] = j;
          ++c;
        }
      }
    }
  }
  if (!FILL) cnt[i] = c;
}

__global__ void k_iota(int n, int *p) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = (int)i;
}

void neigh_build(Model &m, int nlocal, int nall, const double *x_dev, const double *lo, const double *hi,
                 double rc_list, hipStream_t s) {
  if (!m.nb_state) m.nb_state = new NbState();
  NbState &st = *(NbState *)m.nb_state;
  BinGrid g;
  long long nbins = 1;
  for (int d = 0; d < 3; ++d) {
    double len = hi[d] - lo[d];
    if (!(len > 0)) throw ArgError("ahip_build_neighbors_dev: empty bounding box");
    int nb = (int)std::floor(len / rc_list);
    if (nb < 1) nb = 1;
    if (nb > 1024) nb = 1024;
    g.nb[d] = nb; g.lo[d] = lo[d]; g.inv[d] = nb / len;
    nbins *= nb;
  }
  const unsigned B = 256;
  auto grid = [&](long long n) { return dim3((unsigned)((n + B - 1) / B)); };
  st.bin_of.reserve((size_t)std::max(nall, 1) * sizeof(int));
  st.bin_cnt.reserve((size_t)(nbins + 1) * sizeof(int));
  st.bin_start.reserve((size_t)(nbins + 2) * sizeof(int));
  st.bin_fill.reserve((size_t)(nbins + 1) * sizeof(int));
  st.sorted.reserve((size_t)std::max(nall, 1) * sizeof(int));
  st.cnt.reserve((size_t)(nlocal + 1) * sizeof(int));
  st.off.reserve((size_t)(nlocal + 2) * sizeof(int));
  st.ilist.reserve((size_t)std::max(nlocal, 1) * sizeof(int));
  AHIP_CHECK(hipMemsetAsync(st.bin_cnt.p, 0, (size_t)(nbins + 1) * sizeof(int), s));
  AHIP_CHECK(hipMemsetAsync(st.bin_fill.p, 0, (size_t)(nbins + 1) * sizeof(int), s));
  if (nall > 0) hipLaunchKernelGGL(k_bin_count, grid(nall), dim3(B), 0, s, nall, x_dev, g, st.bin_of.as<int>(), st.bin_cnt.as<int>());
  AHIP_CHECK(prim_exclusive_scan_i32(m.prim, st.bin_cnt.as<int>(), st.bin_start.as<int>(), (int)nbins, s));
  if (nall > 0) hipLaunchKernelGGL(k_bin_fill, grid(nall), dim3(B), 0, s, nall, st.bin_of.as<int>(), st.bin_start.as<int>(), st.bin_fill.as<int>(), st.sorted.as<int>());
  hipLaunchKernelGGL(k_bin_sort, grid(nbins), dim3(B), 0, s, (int)nbins, st.bin_start.as<int>(), st.sorted.as<int>());
  const double rcsq = rc_list * rc_list;
  if (nlocal > 0) {
    hipLaunchKernelGGL(k_neigh_pass<false>, grid(nlocal), dim3(B), 0, s, nlocal, x_dev, g, st.bin_start.as<int>(), st.sorted.as<int>(), rcsq, st.cnt.as<int>(), (const int *)nullptr, (int *)nullptr);
    hipLaunchKernelGGL(k_iota, grid(nlocal), dim3(B), 0, s, nlocal, st.ilist.as<int>());
  }
  AHIP_CHECK(prim_exclusive_scan_i32(m.prim, st.cnt.as<int>(), st.off.as<int>(), nlocal, s));
  int tot = 0, maxrow = 0;
  st.box.reserve(64);
  AHIP_CHECK(prim_max_i32(st.cnt.as<int>(), nlocal, st.box.as<int>(), s));
  AHIP_CHECK(hipMemcpyAsync(&tot, st.off.as<int>() + nlocal, sizeof(int), hipMemcpyDeviceToHost, s));
  AHIP_CHECK(hipMemcpyAsync(&maxrow, st.box.as<int>(), sizeof(int), hipMemcpyDeviceToHost, s));
  AHIP_CHECK(hipStreamSynchronize(s));
  st.nlj.reserve((size_t)std::max(tot, 1) * sizeof(int));
  if (nlocal > 0)
    hipLaunchKernelGGL(k_neigh_pass<true>, grid(nlocal), dim3(B), 0, s, nlocal, x_dev, g, st.bin_start.as<int>(), st.sorted.as<int>(), rcsq, (int *)nullptr, st.off.as<int>(), st.nlj.as<int>());
  AHIP_CHECK(hipGetLastError());
  m.d_ilist = st.ilist.as<int>();
  m.d_nloff = st.off.as<int>();
  m.d_nlj = st.nlj.as<int>();
  m.inum = nlocal; m.nall = nall; m.nneigh = tot; m.have_list = true;
  m.max_list_row = maxrow;
  m.h_ilist.clear();
}

void neigh_free(Model &m) {
  if (!m.nb_state) return;
  NbState *st = (NbState *)m.nb_state;
  for (DevBuf *b : {&st->bin_of, &st->bin_cnt, &st->bin_start, &st->bin_fill, &st->sorted, &st->cnt, &st->off, &st->nlj, &st->ilist, &st->box})
    b->release();
  delete st;
  m.nb_state = nullptr;
}

struct MassTab { double inv_mass[16]; };

__global__ void k_nve(int mode, int n, double *x, double *v, const double *f, const int *mtype, MassTab mt, double dt, double dtf) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 3LL * n) return;
  long long i = t / 3;
  double dtfm = dtf * mt.inv_mass[mtype[i]];
  double vv = v[t] + dtfm * f[t];
  v[t] = vv;
  if (mode == 0) x[t] += dt * vv;
}

void nve_step(int mode, int n, double *x, double *v, const double *f, const int *mtype, const double *mass_host,
              int ntypes, double dt, double ftm2v, hipStream_t s) {
  if (ntypes > 16) throw UnsupportedError("nve: more than 16 model types");
  MassTab mt;
  for (int k = 0; k < 16; ++k) mt.inv_mass[k] = k < ntypes ? 1.0 / mass_host[k] : 0.0;
  if (n <= 0) return;
  const unsigned B = 256;
  hipLaunchKernelGGL(k_nve, dim3((unsigned)((3LL * n + B - 1) / B)), dim3(B), 0, s, mode, n, x, v, f, mtype, mt, dt, 0.5 * dt * ftm2v);
  AHIP_CHECK(hipGetLastError());
}

}  // namespace ahip
