// Generic (layer-at-a-time) HIP kernels of the Allegro model: forward and hand-derived backward.
//
// These are the correctness-first kernels: one thread per output element, activations in HBM,
// no LDS, no cross-lane traffic.  They run any model shape (any l_max <= 3, any widths, any
// number of edges per atom) in float32 or float64 and are the on-device cross-check of the fused
// MFMA kernel (fused_kernels.h).  The arithmetic is the model spec of DESIGN.md; the reference
// executes the same graph inside libtorch (/root/reference/pair_nequip_allegro.cpp:409-430).
//
// Layouts (row-major, edge-major):  scalars x[e][S];  tensors V[e][lm][u] (u fastest);
// per-l weights om[e][l][u];  environments env[centre][lm][u].
#pragma once
#include <hip/hip_runtime.h>

#include "cg_tables.h"

namespace ahip {

#define AHIP_GID() ((long long)blockIdx.x * (long long)blockDim.x + (long long)threadIdx.x)

template <typename T> __device__ inline T silu_f(T z) { return z / (T(1) + exp(-z)); }
template <typename T> __device__ inline T silu_df(T z) {
  T s = T(1) / (T(1) + exp(-z));
  return s * (T(1) + z * (T(1) - s));
}

// l of a flattened (l,m) index, l <= 2
__device__ inline int l_of_lm(int lm) { return lm == 0 ? 0 : (lm < 4 ? 1 : (lm < 9 ? 2 : 3)); }
__device__ inline int nl_of_D(int D) { return D == 1 ? 1 : (D == 4 ? 2 : (D == 9 ? 3 : 4)); }      // l_max + 1 from (l_max + 1)^2

// ---------------------------------------------------------------------------- edge build
// Pass 1 of preprocess() (pair_nequip_allegro.cpp:488-512): count neighbours within the
// per-type-pair cutoff; strict reference semantics rsq <= cut^2 in float64.
__global__ void k_count_edges(int inum, const int *ilist, const int *nl_off, const int *nl_j,
                              const double *x, const int *ftype, const double *cutsq, int nft,
                              int *cnt) {
  long long ii = AHIP_GID();
  if (ii >= inum) return;
  int i = ilist[ii];
  double xi = x[3 * (long long)i], yi = x[3 * (long long)i + 1], zi = x[3 * (long long)i + 2];
  const double *crow = cutsq + (long long)ftype[i] * nft;
  int c = 0;
  for (int p = nl_off[ii]; p < nl_off[ii + 1]; ++p) {
    int j = nl_j[p];
    double dx = xi - x[3 * (long long)j], dy = yi - x[3 * (long long)j + 1], dz = zi - x[3 * (long long)j + 2];
    double rsq = dx * dx + dy * dy + dz * dz;
    if (rsq <= crow[ftype[j]]) ++c;
  }
  cnt[ii] = c;
}

// Pass 2 (pair_nequip_allegro.cpp:566-629): emit edges grouped by centre, in list order.
template <typename T>
__global__ void k_fill_edges(int inum, const int *ilist, const int *nl_off, const int *nl_j,
                             const double *x, const int *ftype, const double *cutsq, int nft,
                             const int *eoff, int *e_ii, int *e_j, T *rvec) {
  long long ii = AHIP_GID();
  if (ii >= inum) return;
  int i = ilist[ii];
  double xi = x[3 * (long long)i], yi = x[3 * (long long)i + 1], zi = x[3 * (long long)i + 2];
  const double *crow = cutsq + (long long)ftype[i] * nft;
  long long e = eoff[ii];
  for (int p = nl_off[ii]; p < nl_off[ii + 1]; ++p) {
    int j = nl_j[p];
    double dx = x[3 * (long long)j] - xi, dy = x[3 * (long long)j + 1] - yi, dz = x[3 * (long long)j + 2] - zi;
    double rsq = dx * dx + dy * dy + dz * dz;
    if (rsq <= crow[ftype[j]]) {
      e_ii[e] = (int)ii;
      e_j[e] = j;
      rvec[3 * e] = (T)dx;          // neighbour - centre, computed in f64, stored in model dtype
      rvec[3 * e + 1] = (T)dy;
      rvec[3 * e + 2] = (T)dz;
      ++e;
    }
  }
}

// ---------------------------------------------------------------------------- geometry
struct GeomParams {
  int B, p, L, D, Tn;
  double r_max;
};

template <typename T> __device__ inline void sh_eval(int L, T nx, T ny, T nz, T *Y) {
  Y[0] = T(1);
  if (L >= 1) {
    const T s3 = T(1.7320508075688772);
    Y[1] = s3 * ny; Y[2] = s3 * nz; Y[3] = s3 * nx;
  }
  if (L >= 2) {
    const T s15 = T(3.872983346207417), s5h = T(1.118033988749895);
    Y[4] = s15 * nx * ny;
    Y[5] = s15 * ny * nz;
    Y[6] = s5h * (T(2) * nz * nz - nx * nx - ny * ny);
    Y[7] = s15 * nx * nz;
    Y[8] = T(0.5) * s15 * (nx * nx - ny * ny);
  }
  if (L >= 3) {            // homogeneous cubics of pair_allegro_amd/cg.py: real_sh (the layer-at-a-time kernels only: no fused kernel has l_max = 3)
    const T c70 = T(2.091650066335189), c105 = T(10.246950765959598), c42 = T(1.620185174601965), c7 = T(1.3228756555322954);
    const T x2 = nx * nx, y2 = ny * ny, z2 = nz * nz;
    Y[9] = c70 * ny * (T(3) * x2 - y2);
    Y[10] = c105 * nx * ny * nz;
    Y[11] = c42 * ny * (T(4) * z2 - x2 - y2);
    Y[12] = c7 * nz * (T(2) * z2 - T(3) * x2 - T(3) * y2);
    Y[13] = c42 * nx * (T(4) * z2 - x2 - y2);
    Y[14] = T(0.5) * c105 * nz * (x2 - y2);
    Y[15] = c70 * nx * (x2 - T(3) * y2);
  }
}

// G = sum_lm dY[lm] * grad_n Y_lm(n)  (homogeneous-polynomial gradients)
template <typename T> __device__ inline void sh_grad_dot(int L, T nx, T ny, T nz, const T *dY, T *G) {
  T gx = 0, gy = 0, gz = 0;
  if (L >= 1) {
    const T s3 = T(1.7320508075688772);
    gy += s3 * dY[1]; gz += s3 * dY[2]; gx += s3 * dY[3];
  }
  if (L >= 2) {
    const T s15 = T(3.872983346207417), s5h = T(1.118033988749895);
    gx += s15 * ny * dY[4]; gy += s15 * nx * dY[4];
    gy += s15 * nz * dY[5]; gz += s15 * ny * dY[5];
    gx += -T(2) * s5h * nx * dY[6]; gy += -T(2) * s5h * ny * dY[6]; gz += T(4) * s5h * nz * dY[6];
    gx += s15 * nz * dY[7]; gz += s15 * nx * dY[7];
    gx += s15 * nx * dY[8]; gy += -s15 * ny * dY[8];
  }
  if (L >= 3) {
    const T c70 = T(2.091650066335189), c105 = T(10.246950765959598), c42 = T(1.620185174601965), c7 = T(1.3228756555322954);
    const T x2 = nx * nx, y2 = ny * ny, z2 = nz * nz, xy = nx * ny, xz = nx * nz, yz = ny * nz;
    gx += c70 * T(6) * xy * dY[9];                  gy += c70 * (T(3) * x2 - T(3) * y2) * dY[9];
    gx += c105 * yz * dY[10];                       gy += c105 * xz * dY[10];                        gz += c105 * xy * dY[10];
    gx += -c42 * T(2) * xy * dY[11];                gy += c42 * (T(4) * z2 - x2 - T(3) * y2) * dY[11]; gz += c42 * T(8) * yz * dY[11];
    gx += -c7 * T(6) * xz * dY[12];                 gy += -c7 * T(6) * yz * dY[12];                  gz += c7 * (T(6) * z2 - T(3) * x2 - T(3) * y2) * dY[12];
    gx += c42 * (T(4) * z2 - T(3) * x2 - y2) * dY[13]; gy += -c42 * T(2) * xy * dY[13];              gz += c42 * T(8) * xz * dY[13];
    gx += c105 * xz * dY[14];                       gy += -c105 * yz * dY[14];                       gz += T(0.5) * c105 * (x2 - y2) * dY[14];
    gx += c70 * (T(3) * x2 - T(3) * y2) * dY[15];   gy += -c70 * T(6) * xy * dY[15];
  }
  G[0] = gx; G[1] = gy; G[2] = gz;
}

template <typename T> __device__ inline void cutoff_eval(int p, T x, T &f, T &df) {
  // f(x) = 1 - (p+1)(p+2)/2 x^p + p(p+2) x^(p+1) - p(p+1)/2 x^(p+2),  0 for x >= 1
  if (x >= T(1)) { f = 0; df = 0; return; }
  T xp1 = T(1);                       // x^(p-1)
  for (int k = 0; k < p - 1; ++k) xp1 *= x;
  T xp = xp1 * x;
  const T a = T(0.5) * T(p + 1) * T(p + 2), b = T(p) * T(p + 2), c = T(0.5) * T(p) * T(p + 1);
  f = T(1) - a * xp + b * xp * x - c * xp * x * x;
  df = -a * T(p) * xp1 + b * T(p + 1) * xp - c * T(p + 2) * xp * x;
}

// per edge: d, fc, bf[B] = bessel*fc, Y[D], two-body MLP input a = [onehot_i, onehot_j, bf]
template <typename T>
__global__ void k_geom_fwd(long long E, GeomParams gp, const T *rvec, const int *e_ii, const int *e_j,
                           const int *ilist, const int *mtype, const double *rcut_model,
                           T *fc, T *Y, T *a_in) {
  long long e = AHIP_GID();
  if (e >= E) return;
  T rx = rvec[3 * e], ry = rvec[3 * e + 1], rz = rvec[3 * e + 2];
  T d = sqrt(rx * rx + ry * ry + rz * rz);
  T inv = T(1) / d;
  int ti = mtype[ilist[e_ii[e]]], tj = mtype[e_j[e]];
  T rc = (T)rcut_model[ti * gp.Tn + tj];
  T xx = d / rc;
  T f, df;
  cutoff_eval<T>(gp.p, xx, f, df);
  fc[e] = f;
  T Yl[16];
  sh_eval<T>(gp.L, rx * inv, ry * inv, rz * inv, Yl);
  for (int k = 0; k < gp.D; ++k) Y[e * gp.D + k] = Yl[k];
  const int Ka = 2 * gp.Tn + gp.B;
  T *a = a_in + e * Ka;
  for (int t = 0; t < gp.Tn; ++t) { a[t] = (t == ti) ? T(1) : T(0); a[gp.Tn + t] = (t == tj) ? T(1) : T(0); }
  const T pref = T(2) / rc;
  for (int n = 1; n <= gp.B; ++n) a[2 * gp.Tn + n - 1] = pref * sin(T(3.14159265358979323846) * T(n) * xx) * inv * f;
}

// backward of the geometry: (dbf = da[2T:], dfc, dY) -> g = dE/dr
template <typename T>
__global__ void k_geom_bwd(long long E, GeomParams gp, const T *rvec, const int *e_ii, const int *e_j,
                           const int *ilist, const int *mtype, const double *rcut_model,
                           const T *da, const T *dfc, const T *dY, T *g) {
  long long e = AHIP_GID();
  if (e >= E) return;
  T rx = rvec[3 * e], ry = rvec[3 * e + 1], rz = rvec[3 * e + 2];
  T d = sqrt(rx * rx + ry * ry + rz * rz);
  T inv = T(1) / d;
  T nx = rx * inv, ny = ry * inv, nz = rz * inv;
  int ti = mtype[ilist[e_ii[e]]], tj = mtype[e_j[e]];
  T rc = (T)rcut_model[ti * gp.Tn + tj];
  T xx = d / rc;
  T f, df;
  cutoff_eval<T>(gp.p, xx, f, df);
  T dfdd = df / rc;
  // radial part
  T dd = dfc[e] * dfdd;
  const int Ka = 2 * gp.Tn + gp.B;
  const T pref = T(2) / rc;
  const T pi = T(3.14159265358979323846);
  for (int n = 1; n <= gp.B; ++n) {
    T arg = pi * T(n) * xx;
    T sn = sin(arg), cs = cos(arg);
    T b = pref * sn * inv;
    T db = pref * (cs * pi * T(n) / rc * inv - sn * inv * inv);
    dd += da[e * Ka + 2 * gp.Tn + n - 1] * (db * f + b * dfdd);
  }
  // angular part
  T G[3];
  sh_grad_dot<T>(gp.L, nx, ny, nz, dY + e * gp.D, G);
  T gn = G[0] * nx + G[1] * ny + G[2] * nz;
  g[3 * e] = dd * nx + (G[0] - gn * nx) * inv;
  g[3 * e + 1] = dd * ny + (G[1] - gn * ny) * inv;
  g[3 * e + 2] = dd * nz + (G[2] - gn * nz) * inv;
}

// ---------------------------------------------------------------------------- dense layers
// out[e][n] = sum_k in[e][k] W[k][n]
template <typename T>
__global__ void k_linear(long long E, int K, int N, const T *in, int ldin, const T *W, T *out, int ldout) {
  long long t = AHIP_GID();
  if (t >= E * N) return;
  long long e = t / N;
  int n = (int)(t - e * N);
  const T *row = in + e * ldin;
  T acc = 0;
  for (int k = 0; k < K; ++k) acc += row[k] * W[(long long)k * N + n];
  out[e * ldout + n] = acc;
}

// din[e][k] (+)= sum_n dout[e][n] W[k][n]
template <typename T>
__global__ void k_linear_bwd(long long E, int K, int N, const T *dout, int lddout, const T *W, T *din,
                             int lddin, int accumulate) {
  long long t = AHIP_GID();
  if (t >= E * K) return;
  long long e = t / K;
  int k = (int)(t - e * K);
  const T *row = dout + e * lddout;
  const T *w = W + (long long)k * N;
  T acc = 0;
  for (int n = 0; n < N; ++n) acc += row[n] * w[n];
  if (accumulate) din[e * lddin + k] += acc; else din[e * lddin + k] = acc;
}

template <typename T> __global__ void k_silu(long long n, const T *z, T *h) {
  long long t = AHIP_GID();
  if (t < n) h[t] = silu_f<T>(z[t]);
}
// dz = dh * silu'(z)   (in place on dh allowed)
template <typename T> __global__ void k_silu_bwd(long long n, const T *z, const T *dh, T *dz) {
  long long t = AHIP_GID();
  if (t < n) dz[t] = dh[t] * silu_df<T>(z[t]);
}

// xout[e][s] = a * xprev[e][s] + b * fc[e] * u[e][s]     (xprev may be NULL -> a term dropped)
template <typename T>
__global__ void k_latent_update(long long E, int S, const T *xprev, const T *u, const T *fc, const T *res,
                                T *xout) {
  long long t = AHIP_GID();
  if (t >= E * S) return;
  long long e = t / S;
  T v = res[1] * fc[e] * u[t];
  if (xprev) v += res[0] * xprev[t];
  xout[t] = v;
}
// backward: du = b fc dx ; dfc[e] += b sum_s u dx ; dxprev = a dx
template <typename T>
__global__ void k_latent_update_bwd(long long E, int S, const T *dx, const T *u, const T *fc, const T *res,
                                    T *du, T *dfc, T *dxprev) {
  long long e = AHIP_GID();
  if (e >= E) return;
  T acc = 0;
  const T b = res[1], a = res[0], f = fc[e];
  for (int s = 0; s < S; ++s) {
    T dxv = dx[e * S + s];
    acc += u[e * S + s] * dxv;
    du[e * S + s] = b * f * dxv;
    if (dxprev) dxprev[e * S + s] = a * dxv;
  }
  dfc[e] += b * acc;
}

// cat[e] = [x[e][0..S), Vp[e][lm=0][0..U)]
template <typename T>
__global__ void k_concat(long long E, int S, int U, const T *x, const T *Vp, int ldVp, T *cat) {
  long long t = AHIP_GID();
  int Wd = S + U;
  if (t >= E * Wd) return;
  long long e = t / Wd;
  int c = (int)(t - e * Wd);
  cat[t] = c < S ? x[e * S + c] : Vp[e * ldVp + (c - S)];
}

// ---------------------------------------------------------------------------- tensor track
// V[e][lm][u] = w[e][l(lm)][u] * Y[e][lm]
template <typename T>
__global__ void k_embed(long long E, int D, int U, const T *w, const T *Y, T *V) {
  long long t = AHIP_GID();
  if (t >= E * D * U) return;
  long long e = t / (D * U);
  int r = (int)(t - e * (long long)D * U);
  int lm = r / U, u = r - lm * U;
  int nl = nl_of_D(D);
  V[t] = w[e * nl * U + l_of_lm(lm) * U + u] * Y[e * D + lm];
}
// dw[e][l][u] = sum_{m in l} dV[e][lm][u] Y[e][lm]
template <typename T>
__global__ void k_embed_bwd_w(long long E, int D, int U, const T *dV, const T *Y, T *dw) {
  long long t = AHIP_GID();
  int nl = nl_of_D(D);
  if (t >= E * nl * U) return;
  long long e = t / (nl * U);
  int r = (int)(t - e * (long long)nl * U);
  int l = r / U, u = r - l * U;
  T acc = 0;
  for (int lm = l * l; lm < (l + 1) * (l + 1); ++lm) acc += dV[(e * D + lm) * U + u] * Y[e * D + lm];
  dw[t] = acc;
}
// dY[e][lm] += sum_u dV[e][lm][u] w[e][l(lm)][u]
template <typename T>
__global__ void k_embed_bwd_Y(long long E, int D, int U, const T *dV, const T *w, T *dY) {
  long long t = AHIP_GID();
  if (t >= E * D) return;
  long long e = t / D;
  int lm = (int)(t - e * D);
  int nl = nl_of_D(D);
  const T *wr = w + e * nl * U + l_of_lm(lm) * U;
  const T *dv = dV + (e * D + lm) * U;
  T acc = 0;
  for (int u = 0; u < U; ++u) acc += dv[u] * wr[u];
  dY[t] += acc;
}

// env[c][lm][u] = cenv * sum_{e in centre c} om[e][l][u] Y[e][lm]   (deterministic, one thread per output)
template <typename T>
__global__ void k_env_reduce(int nc, int c0, const int *eoff, long long e0, int D, int U, const T *om,
                             const T *Y, T cenv, T *env) {
  long long t = AHIP_GID();
  if (t >= (long long)nc * D * U) return;
  int c = (int)(t / (D * U));
  int r = (int)(t - (long long)c * D * U);
  int lm = r / U, u = r - lm * U;
  int nl = nl_of_D(D);
  int l = l_of_lm(lm);
  T acc = 0;
  for (long long e = eoff[c0 + c] - e0; e < eoff[c0 + c + 1] - e0; ++e) acc += om[e * nl * U + l * U + u] * Y[e * D + lm];
  env[t] = cenv * acc;
}

// out[c][f] = scale * sum_{e in centre c} in[e][f]
template <typename T>
__global__ void k_segment_sum(int nc, int c0, const int *eoff, long long e0, int F, const T *in, T scale, T *out) {
  long long t = AHIP_GID();
  if (t >= (long long)nc * F) return;
  int c = (int)(t / F);
  int f = (int)(t - (long long)c * F);
  T acc = 0;
  for (long long e = eoff[c0 + c] - e0; e < eoff[c0 + c + 1] - e0; ++e) acc += in[e * F + f];
  out[t] = scale * acc;
}

// Tensor product, one thread per (edge, channel).
//   Vp[e][lm3][u] = sum_entries pw[path][u] * c * V[e][i1][u] * env[centre(e)][i2][u]
template <typename T>
__global__ void k_tp_fwd(long long E, int D, int Dout, int U, const AhipCgEntry *cg, int ncg, const T *pw,
                         const T *V, const T *env, const int *e_ii, int c0, T *Vp) {
  long long t = AHIP_GID();
  if (t >= E * U) return;
  long long e = t / U;
  int u = (int)(t - e * U);
  const T *v = V + e * D * U + u;
  const T *en = env + (long long)(e_ii[e] - c0) * D * U + u;
  T out[16];
  for (int k = 0; k < Dout; ++k) out[k] = 0;
  for (int q = 0; q < ncg; ++q) {
    AhipCgEntry c = cg[q];
    out[c.i3] += pw[c.path * U + u] * (T)c.c * v[c.i1 * U] * en[c.i2 * U];
  }
  for (int k = 0; k < Dout; ++k) Vp[(e * Dout + k) * U + u] = out[k];
}

// backward: dV[e][i1][u] = sum pw c env[i2] dVp[i3];  denv_e[e][i2][u] = sum pw c V[i1] dVp[i3]
template <typename T>
__global__ void k_tp_bwd(long long E, int D, int Dout, int U, const AhipCgEntry *cg, int ncg, const T *pw,
                         const T *V, const T *env, const int *e_ii, int c0, const T *dVp, T *dV, T *denv_e) {
  long long t = AHIP_GID();
  if (t >= E * U) return;
  long long e = t / U;
  int u = (int)(t - e * U);
  const T *v = V + e * D * U + u;
  const T *en = env + (long long)(e_ii[e] - c0) * D * U + u;
  T a[16], b[16];
  for (int k = 0; k < D; ++k) { a[k] = 0; b[k] = 0; }
  for (int q = 0; q < ncg; ++q) {
    AhipCgEntry c = cg[q];
    T w = pw[c.path * U + u] * (T)c.c * dVp[(e * Dout + c.i3) * U + u];
    a[c.i1] += w * en[c.i2 * U];
    b[c.i2] += w * v[c.i1 * U];
  }
  for (int k = 0; k < D; ++k) { dV[(e * D + k) * U + u] = a[k]; denv_e[(e * D + k) * U + u] = b[k]; }
}

// dA[e][lm][u] = denv[centre][lm][u] (already scaled);  dom[e][l][u] = sum_m dA Y ;  (dY handled below)
template <typename T>
__global__ void k_env_bwd_om(long long E, int D, int U, const T *denv, const int *e_ii, int c0, const T *Y, T *dom) {
  long long t = AHIP_GID();
  int nl = nl_of_D(D);
  if (t >= E * nl * U) return;
  long long e = t / (nl * U);
  int r = (int)(t - e * (long long)nl * U);
  int l = r / U, u = r - l * U;
  const T *de = denv + (long long)(e_ii[e] - c0) * D * U;
  T acc = 0;
  for (int lm = l * l; lm < (l + 1) * (l + 1); ++lm) acc += de[lm * U + u] * Y[e * D + lm];
  dom[t] = acc;
}
template <typename T>
__global__ void k_env_bwd_Y(long long E, int D, int U, const T *denv, const int *e_ii, int c0, const T *om, T *dY) {
  long long t = AHIP_GID();
  if (t >= E * D) return;
  long long e = t / D;
  int lm = (int)(t - e * D);
  int nl = nl_of_D(D);
  const T *de = denv + (long long)(e_ii[e] - c0) * D * U + lm * U;
  const T *o = om + e * nl * U + l_of_lm(lm) * U;
  T acc = 0;
  for (int u = 0; u < U; ++u) acc += de[u] * o[u];
  dY[t] += acc;
}

// V[e][lm][v] = sum_u Vp[e][lm][u] mix[l][u][v]
template <typename T>
__global__ void k_mix(long long E, int D, int U, const T *Vp, const T *mix, T *V) {
  long long t = AHIP_GID();
  if (t >= E * D * U) return;
  long long e = t / (D * U);
  int r = (int)(t - e * (long long)D * U);
  int lm = r / U, v = r - lm * U;
  const T *m = mix + (long long)l_of_lm(lm) * U * U;
  const T *in = Vp + (e * D + lm) * U;
  T acc = 0;
  for (int u = 0; u < U; ++u) acc += in[u] * m[u * U + v];
  V[t] = acc;
}
// dVp[e][lm][u] = sum_v dV[e][lm][v] mix[l][u][v]   (+ ds[e][u] on lm = 0 when ds != NULL)
template <typename T>
__global__ void k_mix_bwd(long long E, int D, int U, const T *dV, const T *mix, const T *ds, int ldds, T *dVp) {
  long long t = AHIP_GID();
  if (t >= E * D * U) return;
  long long e = t / (D * U);
  int r = (int)(t - e * (long long)D * U);
  int lm = r / U, u = r - lm * U;
  T acc = 0;
  if (dV) {
    const T *m = mix + (long long)l_of_lm(lm) * U * U + (long long)u * U;
    const T *in = dV + (e * D + lm) * U;
    for (int v = 0; v < U; ++v) acc += in[v] * m[v];
  }
  if (lm == 0 && ds) acc += ds[e * ldds + u];
  dVp[t] = acc;
}

// ---------------------------------------------------------------------------- read-out
// deps[e] = scale[type(centre)] * cE
template <typename T>
__global__ void k_seed_deps(long long E, const int *e_ii, const int *ilist, const int *mtype, const T *scale,
                            T cE, T *deps) {
  long long e = AHIP_GID();
  if (e >= E) return;
  deps[e] = scale[mtype[ilist[e_ii[e]]]] * cE;
}

// Per-centre read-out of pair_nequip_allegro.cpp:366-380: E_i, F_i += sum_e g_e, F_j -= g_e,
// plus per-centre partial sums {E_i, virial[6]} (reduced afterwards by prim_sum_columns_f64).
//   virial = -sum_e sym(r_e (x) g_e), LAMMPS order xx,yy,zz,xy,xz,yz (pair_nequip_allegro.cpp:387-392)
template <typename T>
__global__ void k_readout(int nc, int c0, const int *eoff, long long e0, const int *ilist, const int *mtype,
                          const int *e_j, const T *rvec, const T *eps, const T *g, const T *scale,
                          const T *shift, T cE, double *f, double *eatom, double *partial /*[inum][7]*/) {
  long long c = AHIP_GID();
  if (c >= nc) return;
  int ii = c0 + (int)c;
  int i = ilist[ii];
  double es = 0, fx = 0, fy = 0, fz = 0, v0 = 0, v1 = 0, v2 = 0, v3 = 0, v4 = 0, v5 = 0;
  for (long long e = eoff[ii] - e0; e < eoff[ii + 1] - e0; ++e) {
    es += (double)eps[e];
    double gx = (double)g[3 * e], gy = (double)g[3 * e + 1], gz = (double)g[3 * e + 2];
    double rx = (double)rvec[3 * e], ry = (double)rvec[3 * e + 1], rz = (double)rvec[3 * e + 2];
    fx += gx; fy += gy; fz += gz;
    int j = e_j[e];
    atomicAdd(&f[3 * (long long)j], -gx);
    atomicAdd(&f[3 * (long long)j + 1], -gy);
    atomicAdd(&f[3 * (long long)j + 2], -gz);
    v0 -= rx * gx; v1 -= ry * gy; v2 -= rz * gz;
    v3 -= 0.5 * (rx * gy + ry * gx);
    v4 -= 0.5 * (rx * gz + rz * gx);
    v5 -= 0.5 * (ry * gz + rz * gy);
  }
  atomicAdd(&f[3 * (long long)i], fx);
  atomicAdd(&f[3 * (long long)i + 1], fy);
  atomicAdd(&f[3 * (long long)i + 2], fz);
  int ti = mtype[i];
  double ei = (double)((T)scale[ti] * ((T)es * cE) + shift[ti]);
  if (eatom) eatom[i] = ei;
  double *p = partial + 7 * (long long)ii;
  p[0] = ei; p[1] = v0; p[2] = v1; p[3] = v2; p[4] = v3; p[5] = v4; p[6] = v5;
}

// dst[e][c] += src[e][c], c < C, independent leading dimensions
template <typename T>
__global__ void k_add_cols(long long E, int C, const T *src, int ldsrc, T *dst, int lddst) {
  long long t = AHIP_GID();
  if (t >= E * C) return;
  long long e = t / C;
  int c = (int)(t - e * C);
  dst[e * lddst + c] += src[e * ldsrc + c];
}

template <typename T> __global__ void k_fill(long long n, T *p, T v) {
  long long t = AHIP_GID();
  if (t < n) p[t] = v;
}

}  // namespace ahip
