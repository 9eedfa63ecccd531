// Dense layers of the generic (any model shape) float32 path on the matrix cores:
//   C[e][n] (+)= sum_k A[e][k] * B(k, n),   B(k, n) = W[k][n]  (forward, x @ W)   or  W[n][k]  (backward, dy @ W^T).
// E is the edge count of a chunk (10^5..10^6), K and N are model widths (8..300), so this is a tall-skinny GEMM
// whose traffic is the activations: 4 E (K + N) bytes -- HBM-bound (intensity K N / (2 (K + N)) = 16 FLOP/B at 64x64).
// One workgroup = 64 rows x 64 columns, four waves of 16 rows each; K is walked in steps of 16 through LDS
// (A tile 64x16, B tile 16x64, padded rows: conflict-free operand reads); v_mfma_f32_16x16x4_f32, exact f32.
// Optional fused epilogues: silu_out (forward: also store h = silu(C)), dsilu_z (backward: C *= silu'(z), z with C's layout).
// Replaces the one-thread-per-output kernels k_linear / k_linear_bwd (generic_kernels.h), which stay for the float64
// debug build and the CPU-emulated tests: they ran at 80-400 GB/s (78 % of the generic path's time).
#include <utility>
#include <hip/hip_runtime.h>

#include <algorithm>

#include "cg_tables.h"
#include "engine.h"

namespace ahip {

typedef float f32x4g __attribute__((ext_vector_type(4)));

template <bool TRANSB, bool VECA>
__global__ void __launch_bounds__(256) k_gemm_f32(long long E, int K, int N, const float *__restrict__ A, int lda,
                                                   const float *__restrict__ W, int ldw, float *__restrict__ C, int ldc,
                                                   int accumulate, float *__restrict__ silu_out, const float *__restrict__ dsilu_z) {
  __shared__ float sA[64][17];
  __shared__ float sB[16][65];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, g = lane >> 4;
  const long long e0 = (long long)blockIdx.x * 64;
  const int n0 = blockIdx.y * 64;
  f32x4g acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = f32x4g{0.f, 0.f, 0.f, 0.f};
  const int ar = tid >> 2, ac = (tid & 3) * 4;          // A tile: row, first of 4 columns
  const int bk = tid >> 4, bn = (tid & 15) * 4;         // B tile: k row, first of 4 columns
  const long long arow = e0 + ar;
  for (int k0 = 0; k0 < K; k0 += 16) {
    if (VECA) {      // rows and K are multiples of 4 floats and 16-byte aligned: one 16-byte load per thread
      f32x4g v = {0.f, 0.f, 0.f, 0.f};
      if (arow < E && k0 + ac < K) v = *(const f32x4g *)(A + arow * lda + k0 + ac);
      sA[ar][ac] = v[0]; sA[ar][ac + 1] = v[1]; sA[ar][ac + 2] = v[2]; sA[ar][ac + 3] = v[3];
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int k = k0 + ac + i;
        sA[ar][ac + i] = (arow < E && k < K) ? A[arow * lda + k] : 0.f;
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = k0 + bk, n = n0 + bn + i;
      float v = 0.f;
      if (k < K && n < N) v = TRANSB ? W[(long long)n * ldw + k] : W[(long long)k * ldw + n];
      sB[bk][bn + i] = v;
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const float a = sA[16 * wave + j][4 * ks + g];
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, sB[4 * ks + g][16 * t + j], acc[t], 0, 0, 0);
    }
    __syncthreads();
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int n = n0 + 16 * t + j;
    if (n >= N) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long long e = e0 + 16 * wave + 4 * g + r;
      if (e < E) {
        float *p = C + e * ldc + n;
        float v = accumulate ? *p + acc[t][r] : acc[t][r];
        if (dsilu_z) {                                  // backward through the SiLU that produced this layer's input
          const float z = dsilu_z[e * ldc + n], sg = 1.f / (1.f + __expf(-z));
          v *= sg * (1.f + z * (1.f - sg));
        }
        *p = v;
        if (silu_out) silu_out[e * ldc + n] = v / (1.f + __expf(-v));     // forward: also h = silu(z)
      }
    }
  }
}

bool gemm_f32(hipStream_t s, long long E, int K, int N, const float *A, int lda, const float *W, int ldw, bool transB, float *C,
              int ldc, bool accumulate, float *silu_out, const float *dsilu_z) {
  if (E <= 0 || N <= 0) return true;
  const dim3 grid((unsigned)((E + 63) / 64), (unsigned)((N + 63) / 64));
  const bool veca = (lda % 4 == 0) && (K % 4 == 0) && (((size_t)A) % 16 == 0);
#define GEMM(TB, VA) hipLaunchKernelGGL((k_gemm_f32<TB, VA>), grid, dim3(256), 0, s, E, K, N, A, lda, W, ldw, C, ldc, accumulate ? 1 : 0, silu_out, dsilu_z)
  if (transB) { if (veca) GEMM(true, true); else GEMM(true, false); }
  else { if (veca) GEMM(false, true); else GEMM(false, false); }
#undef GEMM
  return true;
}

// ---- row-reduction kernels (one wave per edge row, coalesced, shuffle reduce) for the two element-wise backward steps
// that the one-thread-per-edge versions did with 256-byte strides (300 GB/s) ------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
// du = b fc dx ; dfc[e] += b sum_s u dx ; dxprev = a dx     (k_latent_update_bwd)
__global__ void __launch_bounds__(256) k_latent_update_bwd_rows(long long E, int S, const float *dx, const float *u, const float *fc,
                                                                 const float *res, float *du, float *dfc, float *dxprev) {
  const int lane = threadIdx.x & 63;
  const long long nw = (long long)gridDim.x * 4;
  const float a = res[0], b = res[1];
  for (long long e = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); e < E; e += nw) {
    const float f = fc[e];
    float acc = 0.f;
    for (int s = lane; s < S; s += 64) {
      const float dxv = dx[e * S + s];
      acc += u[e * S + s] * dxv;
      du[e * S + s] = b * f * dxv;
      if (dxprev) dxprev[e * S + s] = a * dxv;
    }
    acc = wave_sum(acc);
    if (lane == 0) dfc[e] += b * acc;
  }
}
// dY[e][lm] += sum_u dV[e][lm][u] w[e][l(lm)][u]            (k_embed_bwd_Y)
__global__ void __launch_bounds__(256) k_embed_bwd_Y_rows(long long E, int D, int U, const float *dV, const float *w, float *dY) {
  const int lane = threadIdx.x & 63;
  const long long nw = (long long)gridDim.x * 4;
  const int nl = D == 1 ? 1 : (D == 4 ? 2 : (D == 9 ? 3 : 4));
  for (long long e = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); e < E; e += nw) {
    for (int lm = 0; lm < D; ++lm) {
      const int l = lm == 0 ? 0 : (lm < 4 ? 1 : (lm < 9 ? 2 : 3));
      float acc = 0.f;
      for (int q = lane; q < U; q += 64) acc += dV[(e * D + lm) * U + q] * w[e * nl * U + l * U + q];
      acc = wave_sum(acc);
      if (lane == 0) dY[e * D + lm] += acc;
    }
  }
}
// ---- tensor product with the Clebsch-Gordan table unrolled at compile time (cg_tables.h is constexpr): every index is a
// constant, so V, env and the outputs stay in registers.  The portable kernels (k_tp_fwd / k_tp_bwd) index private arrays
// with table entries read at run time (scratch memory) and re-load V / env per entry: 45 % of the l_max = 2 path's time.
template <int L> struct CgTab;
template <> struct CgTab<1> { static constexpr const AhipCgEntry *tab = ahip_cg_l1; static constexpr int N = AHIP_CG_L1_N, NS = AHIP_CG_L1_NSCALAR, NP = AHIP_CG_L1_NPATHS; };
template <> struct CgTab<2> { static constexpr const AhipCgEntry *tab = ahip_cg_l2; static constexpr int N = AHIP_CG_L2_N, NS = AHIP_CG_L2_NSCALAR, NP = AHIP_CG_L2_NPATHS; };
template <> struct CgTab<3> { static constexpr const AhipCgEntry *tab = ahip_cg_l3; static constexpr int N = AHIP_CG_L3_N, NS = AHIP_CG_L3_NSCALAR, NP = AHIP_CG_L3_NPATHS; };      // round 6: 611 entries, 34 paths

template <int L, bool SCALAR>
__global__ void __launch_bounds__(256) k_tp_fwd_unrolled(long long E, int U, const float *__restrict__ pw, const float *__restrict__ V,
                                                          const float *__restrict__ env, const int *__restrict__ e_ii, int c0,
                                                          float *__restrict__ Vp) {
  constexpr int D = (L + 1) * (L + 1), DOUT = SCALAR ? 1 : D, N = SCALAR ? CgTab<L>::NS : CgTab<L>::N, NP = CgTab<L>::NP;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= E * U) return;
  const long long e = t / U;
  const int u = (int)(t - e * U);
  const float *v = V + e * D * U + u;
  const float *en = env + (long long)(e_ii[e] - c0) * D * U + u;
  float vv[D], ee[D], pp[NP], out[DOUT];
#pragma unroll
  for (int k = 0; k < D; ++k) { vv[k] = v[k * U]; ee[k] = en[k * U]; }
#pragma unroll
  for (int k = 0; k < NP; ++k) pp[k] = pw[k * U + u];
#pragma unroll
  for (int k = 0; k < DOUT; ++k) out[k] = 0.f;
#pragma unroll
  for (int q = 0; q < N; ++q) {
    constexpr const AhipCgEntry *tab = CgTab<L>::tab;
    out[tab[q].i3] += pp[tab[q].path] * (float)tab[q].c * vv[tab[q].i1] * ee[tab[q].i2];
  }
#pragma unroll
  for (int k = 0; k < DOUT; ++k) Vp[(e * DOUT + k) * U + u] = out[k];
}
// The backward table walk is unrolled by template expansion, not by `#pragma unroll`: the l_max = 3 body (611 entries) is over the
// compiler's pragma-unroll budget, which left a run-time loop of 47-entry blocks with the path weights in scratch and the
// accumulators selected by compare chains -- 47 instructions per entry, 5.8 ms per call on 300 k edges (40 % of an evaluation).
template <int L, int Q, int D, int DOUT, int NP>
__device__ __forceinline__ void tp_bwd_entry(const float (&pp)[NP], const float (&gg)[DOUT], const float (&vv)[D], const float (&ee)[D], float (&a)[D], float (&b)[D]) {
  constexpr AhipCgEntry t = CgTab<L>::tab[Q];
  const float wv = pp[t.path] * (float)t.c * gg[t.i3];
  a[t.i1] += wv * ee[t.i2];
  b[t.i2] += wv * vv[t.i1];
}
template <int L, int Q0, int D, int DOUT, int NP, int... I>
__device__ __forceinline__ void tp_bwd_block(std::integer_sequence<int, I...>, const float (&pp)[NP], const float (&gg)[DOUT], const float (&vv)[D], const float (&ee)[D], float (&a)[D], float (&b)[D]) {
  (tp_bwd_entry<L, Q0 + I, D, DOUT, NP>(pp, gg, vv, ee, a, b), ...);
}
template <int L, int N, int Q0, int D, int DOUT, int NP>
__device__ __forceinline__ void tp_bwd_walk(const float (&pp)[NP], const float (&gg)[DOUT], const float (&vv)[D], const float (&ee)[D], float (&a)[D], float (&b)[D]) {
  if constexpr (Q0 < N) {
    constexpr int B = N - Q0 < 64 ? N - Q0 : 64;
    tp_bwd_block<L, Q0, D, DOUT, NP>(std::make_integer_sequence<int, B>{}, pp, gg, vv, ee, a, b);
    tp_bwd_walk<L, N, Q0 + B, D, DOUT, NP>(pp, gg, vv, ee, a, b);
  }
}
template <int L, bool SCALAR>
__global__ void __launch_bounds__(256, 2) k_tp_bwd_unrolled(long long E, int U, const float *__restrict__ pw, const float *__restrict__ V,
                                                          const float *__restrict__ env, const int *__restrict__ e_ii, int c0,
                                                          const float *__restrict__ dVp, float *__restrict__ dV, float *__restrict__ denv_e) {
  constexpr int D = (L + 1) * (L + 1), DOUT = SCALAR ? 1 : D, N = SCALAR ? CgTab<L>::NS : CgTab<L>::N, NP = CgTab<L>::NP;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= E * U) return;
  const long long e = t / U;
  const int u = (int)(t - e * U);
  const float *v = V + e * D * U + u;
  const float *en = env + (long long)(e_ii[e] - c0) * D * U + u;
  float vv[D], ee[D], pp[NP], gg[DOUT], a[D], b[D];
#pragma unroll
  for (int k = 0; k < D; ++k) { vv[k] = v[k * U]; ee[k] = en[k * U]; a[k] = 0.f; b[k] = 0.f; }
#pragma unroll
  for (int k = 0; k < NP; ++k) pp[k] = pw[k * U + u];
#pragma unroll
  for (int k = 0; k < DOUT; ++k) gg[k] = dVp[(e * DOUT + k) * U + u];
  tp_bwd_walk<L, N, 0, D, DOUT, NP>(pp, gg, vv, ee, a, b);
#pragma unroll
  for (int k = 0; k < D; ++k) { dV[(e * D + k) * U + u] = a[k]; denv_e[(e * D + k) * U + u] = b[k]; }
}

bool tp_fwd_f32(hipStream_t s, long long E, int L, bool scalar_only, int U, const float *pw, const float *V, const float *env,
                const int *e_ii, int c0, float *Vp) {
  if (L < 1 || L > 3) return false;
  if (E <= 0) return true;
  const dim3 grid((unsigned)((E * U + 255) / 256));
#define TPF(LV, SV) hipLaunchKernelGGL((k_tp_fwd_unrolled<LV, SV>), grid, dim3(256), 0, s, E, U, pw, V, env, e_ii, c0, Vp)
  if (L == 1) { if (scalar_only) TPF(1, true); else TPF(1, false); }
  else if (L == 2) { if (scalar_only) TPF(2, true); else TPF(2, false); }
  else { if (scalar_only) TPF(3, true); else TPF(3, false); }
#undef TPF
  return true;
}
bool tp_bwd_f32(hipStream_t s, long long E, int L, bool scalar_only, int U, const float *pw, const float *V, const float *env,
                const int *e_ii, int c0, const float *dVp, float *dV, float *denv_e) {
  if (L < 1 || L > 3) return false;
  if (E <= 0) return true;
  const dim3 grid((unsigned)((E * U + 255) / 256));
#define TPB(LV, SV) hipLaunchKernelGGL((k_tp_bwd_unrolled<LV, SV>), grid, dim3(256), 0, s, E, U, pw, V, env, e_ii, c0, dVp, dV, denv_e)
  if (L == 1) { if (scalar_only) TPB(1, true); else TPB(1, false); }
  else if (L == 2) { if (scalar_only) TPB(2, true); else TPB(2, false); }
  else { if (scalar_only) TPB(3, true); else TPB(3, false); }
#undef TPB
  return true;
}

static unsigned row_grid(long long E) { return (unsigned)std::min<long long>((E + 3) / 4, 256LL * 64); }
// dY[e][lm] += sum_u denv[centre(e)][lm][u] om[e][l(lm)][u]                     (k_env_bwd_Y)
__global__ void __launch_bounds__(256) k_env_bwd_Y_rows(long long E, int D, int U, const float *denv, const int *e_ii, int c0,
                                                         const float *om, float *dY) {
  const int lane = threadIdx.x & 63;
  const long long nw = (long long)gridDim.x * 4;
  const int nl = D == 1 ? 1 : (D == 4 ? 2 : (D == 9 ? 3 : 4));
  for (long long e = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); e < E; e += nw) {
    const float *de = denv + (long long)(e_ii[e] - c0) * D * U;
    for (int lm = 0; lm < D; ++lm) {
      const int l = lm == 0 ? 0 : (lm < 4 ? 1 : (lm < 9 ? 2 : 3));
      float acc = 0.f;
      for (int q = lane; q < U; q += 64) acc += de[lm * U + q] * om[e * nl * U + l * U + q];
      acc = wave_sum(acc);
      if (lane == 0) dY[e * D + lm] += acc;
    }
  }
}
bool env_bwd_Y_f32(hipStream_t s, long long E, int D, int U, const float *denv, const int *e_ii, int c0, const float *om, float *dY) {
  if (E > 0) hipLaunchKernelGGL(k_env_bwd_Y_rows, dim3(row_grid(E)), dim3(256), 0, s, E, D, U, denv, e_ii, c0, om, dY);
  return true;
}

bool latent_update_bwd_f32(hipStream_t s, long long E, int S, const float *dx, const float *u, const float *fc, const float *res,
                           float *du, float *dfc, float *dxprev) {
  if (E > 0) hipLaunchKernelGGL(k_latent_update_bwd_rows, dim3(row_grid(E)), dim3(256), 0, s, E, S, dx, u, fc, res, du, dfc, dxprev);
  return true;
}
bool embed_bwd_Y_f32(hipStream_t s, long long E, int D, int U, const float *dV, const float *w, float *dY) {
  if (E > 0) hipLaunchKernelGGL(k_embed_bwd_Y_rows, dim3(row_grid(E)), dim3(256), 0, s, E, D, U, dV, w, dY);
  return true;
}

}  // namespace ahip
