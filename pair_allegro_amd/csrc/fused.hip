// Fused MFMA path: the whole Allegro model (forward + hand-derived backward) for a tile of centre
// atoms in ONE kernel launch, float32 compute on the gfx950 matrix cores.
//
// Mapping (DESIGN.md "Fused kernel"):
//  * tile  = consecutive centre atoms whose edges (<= 128) fit the 128 edge slots of a 256-thread
//            workgroup; wave w owns slots 32w..32w+31; lane = (slot = lane & 31, half h = lane >> 5).
//  * every per-edge feature vector lives in registers in the v_mfma_f32_32x32x2_f32 C/D layout:
//            tile t, register r of lane (slot, h)  <->  feature 32 t + (r & 3) + 8 (r >> 2) + 4 h.
//            With D = W^T-tile (rows = output features) x activations (cols = edges), register r of an
//            output tile is exactly the B operand of MFMA step r of the next layer: the MLP chains run
//            register-to-register with no LDS traffic and no shuffles.
//  * weights are pre-swizzled on the host into A-operand fragment order (one coalesced 1 KiB
//            dwordx4 load feeds 4 MFMAs); a transposed copy serves the backward pass.
//  * the only cross-edge coupling -- the per-centre environment sum and its gradient -- goes through
//            an LDS staging tile [128 slots][128 features] and a deterministic per-atom reduction.
//  * activations needed by the backward pass are spilled as raw register images to a per-wave
//            private scratch (written and re-read by the same wave within the same tile, so it lives
//            in L2 / Infinity Cache, not HBM).
//
// Supported model shape (others run the generic path): l_max = 1, 32 tensor features, 64 scalars,
// MLP 2 x 64, read-out 1 x 32, 8 Bessels, <= 3 layers, <= 4 types.  Reference graph:
// the TorchScript model executed at /root/reference/pair_nequip_allegro.cpp:409-430.
#include <hip/hip_runtime.h>

#include <cmath>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/allegro_hip.h"
#include "engine.h"
#include "prims.h"

namespace ahip {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

static constexpr int TILE_SLOTS = 128;
static constexpr int MAXA = 12;          // centre atoms per tile (LDS budget: env rows + the 64 KiB park region)
static constexpr int MAXNL = 3;
static constexpr int STG_LD = 129;       // staging leading dimension: all 128 features of a slot (odd -> conflict-free)
static constexpr int ENV_LD = 129;       // per-atom environment row
static constexpr int SEG = 512;          // atoms per sequential packing segment
static constexpr int ROW = 1024;         // floats per saved register image (16 regs x 64 lanes)

__host__ __device__ inline int feat_of(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

struct FusedArgs {
  // edge list
  const int *eoff, *e_ii, *e_j, *ilist, *mtype;
  const float *rvec;
  const double *rcut;            // [T*T]
  int T, NL, p;
  float cenv;
  // tiles
  const int *tile_a0, *ntiles;
  // weights (offsets in floats into wbase)
  const float *wbase;
  int wbytes;
  int o_tpl, o_pair, o_tb_wc, o_tb_w1, o_tb_w2, o_emb, o_out0, o_out1, o_scale, o_shift;
  int o_tb_wcT, o_tb_w1T, o_tb_w2T, o_embT, o_out0T;
  int o_env[MAXNL], o_lat0[MAXNL], o_lat1[MAXNL], o_lat2[MAXNL], o_mix[MAXNL], o_tp[MAXNL], o_res[MAXNL];
  int o_envT[MAXNL], o_lat0T[MAXNL], o_lat1T[MAXNL], o_lat2T[MAXNL], o_mixT[MAXNL];
  // scratch
  float *scratch;
  long long wg_scratch, wave_scratch;     // floats
  // outputs
  double *f, *eatom, *partial;            // partial [gridDim.x][7]
  long long *prof;                        // [PH_N] or unused
  float *dbg;                             // [E][8] per-edge diagnostics or null
};

struct __attribute__((aligned(16))) Lds {
  float stage[TILE_SLOTS * STG_LD];
  float env[MAXNL][MAXA * ENV_LD];
  float denv[MAXA * ENV_LD];
  float tp[MAXNL][5 * 32];                // tensor-product path weights [layer][path][u]
  float park[4][4 * ROW];                 // per-wave private park: 4 register images (V^{k+1} forward, dE/dV backward)
  float vir[4][8];                        // per-wave virial partials
  float ea[MAXA];
  int aoff[MAXA + 2];
};

// ---------------------------------------------------------------------------- device helpers
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x16 mfma(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// 16-byte buffer accesses: wave-uniform descriptor + scalar byte offset + per-lane byte offset
__device__ __forceinline__ f32x4 bload(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ void bstore(__amdgpu_buffer_rsrc_t r, int voff, int soff, f32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, soff, 0);
}

// out[NT] (+)= W-tiles x in[KT].  Fragment layout: [ot][kt][q][lane][4], r = 4 q + c;
// wo = offset of the fragment block in floats (wave-uniform), v16 = lane * 16.
// The MFMA chain is serial (64-cycle issue = latency), so the only thing to hide is the weight
// fragment latency: a PF-deep register ring keeps PF fragment loads (PF x 256 MFMA cycles) in
// flight; sched_barrier pins the load -> 4 MFMA order so the compiler cannot sink the loads back
// next to their use (it did: every load was followed by s_waitcnt vmcnt(0)).
template <int KT, int NT, int KQ_LAST = 4, bool ACC = false, int PF = 8>
__device__ __forceinline__ void linear(__amdgpu_buffer_rsrc_t W, int wo, const f32x16 (&in)[KT], f32x16 (&out)[NT], int v16) {
  constexpr int SPO = (KT - 1) * 4 + KQ_LAST;       // 4-MFMA steps per output tile
  constexpr int NS = NT * SPO;
  f32x4 ring[PF];
#pragma unroll
  for (int i = 0; i < PF; ++i)
    if (i < NS) ring[i] = bload(W, v16, (wo + (((i / SPO) * KT + (i % SPO) / 4) * 4 + (i % SPO) % 4) * 256) * 4);
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    const int ot = i / SPO, j = i % SPO, kt = j / 4, q = j % 4;
    if (j == 0) {
      if (ACC) acc = out[ot];
      else {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      }
    }
    const f32x4 a = ring[i % PF];
    if (i + PF < NS) {
      const int n = i + PF;
      ring[i % PF] = bload(W, v16, (wo + (((n / SPO) * KT + (n % SPO) / 4) * 4 + (n % SPO) % 4) * 256) * 4);
    }
    acc = mfma(a.x, in[kt][4 * q + 0], acc);
    acc = mfma(a.y, in[kt][4 * q + 1], acc);
    acc = mfma(a.z, in[kt][4 * q + 2], acc);
    acc = mfma(a.w, in[kt][4 * q + 3], acc);
    if (j == SPO - 1) out[ot] = acc;
    __builtin_amdgcn_sched_barrier(0);
  }
}

__device__ __forceinline__ float sigmoidf_fast(float z) { return __builtin_amdgcn_rcpf(1.f + __expf(-z)); }
__device__ __forceinline__ float silu1(float z) { return z * sigmoidf_fast(z); }
__device__ __forceinline__ float dsilu1(float z) {
  float s = sigmoidf_fast(z);
  return s * (1.f + z * (1.f - s));
}

template <int NT> __device__ __forceinline__ void silu_inplace(f32x16 (&z)[NT]) {
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) z[t][r] = silu1(z[t][r]);
}

// saved register images: row = 16 regs x 64 lanes, layout [q][lane][4]
template <int NT> __device__ __forceinline__ void save_rows(__amdgpu_buffer_rsrc_t S, int row0, const f32x16 (&v)[NT], int v16) {
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 x = {v[t][4 * q], v[t][4 * q + 1], v[t][4 * q + 2], v[t][4 * q + 3]};
      bstore(S, v16, ((row0 + t) * ROW + q * 256) * 4, x);
    }
}
template <int NT> __device__ __forceinline__ void load_rows(__amdgpu_buffer_rsrc_t S, int row0, f32x16 (&v)[NT], int v16) {
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 x = bload(S, v16, ((row0 + t) * ROW + q * 256) * 4);
      v[t][4 * q] = x.x; v[t][4 * q + 1] = x.y; v[t][4 * q + 2] = x.z; v[t][4 * q + 3] = x.w;
    }
}
__device__ __forceinline__ f32x4 rowq(__amdgpu_buffer_rsrc_t S, int row, int q, int v16) {
  return bload(S, v16, (row * ROW + q * 256) * 4);
}
__device__ __forceinline__ f32x4 hvecq(__amdgpu_buffer_rsrc_t W, int wo, int q, int h16) {
  return bload(W, h16, (wo + q * 8) * 4);
}
template <int NT> __device__ __forceinline__ void mul_dsilu(f32x16 (&d)[NT], const f32x16 (&z)[NT]) {
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) d[t][r] *= dsilu1(z[t][r]);
}
// z *= silu'(saved row)
template <int NT> __device__ __forceinline__ void mul_dsilu_rows(__amdgpu_buffer_rsrc_t S, int row0, f32x16 (&d)[NT], int v16) {
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 z = bload(S, v16, ((row0 + t) * ROW + q * 256) * 4);
      d[t][4 * q] *= dsilu1(z.x); d[t][4 * q + 1] *= dsilu1(z.y); d[t][4 * q + 2] *= dsilu1(z.z); d[t][4 * q + 3] *= dsilu1(z.w);
    }
}

// wave-private LDS park rows, same [q][lane][4] image as the scratch rows (16 B per lane, conflict-free)
__device__ __forceinline__ void park_store(float *pk, int row, const f32x16 &v, int lane) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    f32x4 x = {v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
    *(f32x4 *)(pk + row * ROW + q * 256 + lane * 4) = x;
  }
}
__device__ __forceinline__ void park_load(const float *pk, int row, f32x16 &v, int lane) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    f32x4 x = *(const f32x4 *)(pk + row * ROW + q * 256 + lane * 4);
    v[4 * q] = x.x; v[4 * q + 1] = x.y; v[4 * q + 2] = x.z; v[4 * q + 3] = x.w;
  }
}

// per-half small vectors stored as [q][h][4]; h16 = h * 16 bytes
__device__ __forceinline__ f32x16 load_hvec(__amdgpu_buffer_rsrc_t W, int wo, int h16) {
  f32x16 v;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    f32x4 x = bload(W, h16, (wo + q * 8) * 4);
    v[4 * q] = x.x; v[4 * q + 1] = x.y; v[4 * q + 2] = x.z; v[4 * q + 3] = x.w;
  }
  return v;
}

__device__ __forceinline__ void cutoff_poly(int p, float x, float &f, float &df) {
  if (x >= 1.f) { f = 0.f; df = 0.f; return; }
  float xp1 = 1.f;
  for (int k = 0; k < p - 1; ++k) xp1 *= x;
  const float xp = xp1 * x;
  const float a = 0.5f * (p + 1) * (p + 2), b = (float)p * (p + 2), c = 0.5f * p * (p + 1);
  f = 1.f - a * xp + b * xp * x - c * xp * x * x;
  df = -a * p * xp1 + b * (p + 1) * xp - c * (p + 2) * xp * x;
}

// scratch row map (per wave)
__device__ __host__ constexpr int R_Z1TB() { return 0; }
__device__ __host__ constexpr int R_Z2TB() { return 2; }
__device__ __host__ constexpr int R_U0() { return 4; }
__device__ __host__ constexpr int R_W0() { return 6; }
__device__ __host__ constexpr int R_DV() { return 8; }                          // parked dV (4 rows)
__device__ __host__ constexpr int R_LAYER(int kk) { return 12 + 12 * kk; }      // OM 2, Z1 2, Z2 2, U 2, VIN 4
__device__ __host__ constexpr int R_TOTAL(int NL) { return 12 + 12 * NL; }

static constexpr float C_S3 = 1.7320508075688772f;
static constexpr float C_P1 = 0.5773502691896258f;     // (1,1,0): sqrt(1) * w3j = 1/sqrt(3)
static constexpr float C_P4 = 0.7071067811865476f;     // (1,1,1): sqrt(3) * w3j = eps_ijk / sqrt(2)

// ---- streamed linear: continuous weight-fragment ring + epilogue under the next tile's MFMAs ----
// The sequence of linears of a tile is static, so the 4-deep fragment ring never drains: while the last
// 4 steps of one linear issue their MFMAs the ring is refilled with the FIRST 4 fragments of the next
// linear (wo_next; they always sit at wo_next + j*256 because every streamed shape has >= 4 steps per
// output tile).  Every streamed linear has a step count that is a multiple of 4, so the ring phase is 0 at
// every call boundary (also across the runtime layer loop).  The element-wise epilogue of output tile t
// (SiLU, save, scaling, SiLU') is executed in small chunks between the MFMA groups of tile t+1.
struct EpiNone {
  __device__ __forceinline__ void tile_done(int, const f32x16 &) const {}
  __device__ __forceinline__ float apply(int, int, float v) const { return v; }
};
struct EpiSave {             // raw rows to scratch, value unchanged
  __amdgpu_buffer_rsrc_t S; int row0, v16;
  __device__ __forceinline__ void tile_done(int ot, const f32x16 &acc) const {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 x = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
      bstore(S, v16, ((row0 + ot) * ROW + q * 256) * 4, x);
    }
  }
  __device__ __forceinline__ float apply(int, int, float v) const { return v; }
};
struct EpiSavePark : EpiSave {  // raw rows to scratch (for the backward pass) and to the LDS park (next layer's forward)
  float *pk; int prow, lane;
  __device__ __forceinline__ void tile_done(int ot, const f32x16 &acc) const {
    EpiSave::tile_done(ot, acc);
    park_store(pk, prow + ot, acc, lane);
  }
};
struct EpiSiluSave : EpiSave {   // raw rows to scratch, out = silu
  __device__ __forceinline__ float apply(int, int, float v) const { return silu1(v); }
};
struct EpiSaveScale : EpiSave {  // raw rows to scratch, out = c * v
  float c;
  __device__ __forceinline__ float apply(int, int, float v) const { return c * v; }
};
template <int NT> struct EpiMulDsilu {       // out = v * silu'(z)
  const f32x16 (&z)[NT];
  __device__ __forceinline__ void tile_done(int, const f32x16 &) const {}
  __device__ __forceinline__ float apply(int ot, int r, float v) const { return v * dsilu1(z[ot][r]); }
};
template <int NT> struct EpiResidual : EpiSave {   // raw u rows to scratch, out = ra * xold + rbf * u
  const f32x16 (&xold)[NT]; float ra, rbf;
  __device__ __forceinline__ float apply(int ot, int r, float v) const { return ra * xold[ot][r] + rbf * v; }
};

template <int KT, int NT, bool ACC, bool HAS_NEXT, class Epi>
__device__ __forceinline__ void linear_s(__amdgpu_buffer_rsrc_t W, int wo, int wo_next, const f32x16 (&in)[KT],
                                         f32x16 (&out)[NT], int v16, f32x4 (&ring)[4], const Epi &epi) {
  constexpr int SPO = KT * 4, NS = NT * SPO;
  f32x16 acc, prev;
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    const int ot = i / SPO, j = i % SPO, kt = j / 4, q = j % 4;
    if (j == 0) {
      if (ACC) acc = out[ot];
      else {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      }
    }
    const f32x4 a = ring[i % 4];
    {
      const int n = i + 4;
      if (n < NS) ring[i % 4] = bload(W, v16, (wo + (((n / SPO) * KT + (n % SPO) / 4) * 4 + (n % SPO) % 4) * 256) * 4);
      else if (HAS_NEXT) ring[i % 4] = bload(W, v16, (wo_next + (n - NS) * 256) * 4);
    }
    // The 4 MFMAs of a step form a dependent chain (64 cycles each): the wave stalls on each one, so the
    // epilogue of the previous output tile is fed one register at a time INTO the gaps of the chain
    // (one element every KT MFMAs), where it executes in the shadow of the MFMA just issued.
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      acc = mfma(a[c], in[kt][4 * q + c], acc);
      if (ot > 0) {
        const int idx = j * 4 + c;                 // MFMA index inside this output tile
        if (idx % KT == 0) {
          const int r = idx / KT;
          out[ot - 1][r] = epi.apply(ot - 1, r, prev[r]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (j == SPO - 1) {
      epi.tile_done(ot, acc);
      if (ot == NT - 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) out[ot][r] = epi.apply(ot, r, acc[r]);
      } else prev = acc;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}
// first 4 fragments of the linear at wo into the ring
__device__ __forceinline__ void ring_prime(__amdgpu_buffer_rsrc_t W, int wo, int v16, f32x4 (&ring)[4]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) ring[j] = bload(W, v16, (wo + j * 256) * 4);
}

// Per-centre sum of the staged tile: dst[a][f] = scale * sum_{slots of a} stage[slot][f], f < 128.
// 256 threads = 2 atoms x 128 features per pass; 4 independent accumulators keep 4 LDS reads in flight.
__device__ __forceinline__ void reduce_stage(const Lds &lds, float *dst, int na, float scale, int tid) {
  const int fidx = tid & 127;
  for (int a = tid >> 7; a < na; a += 2) {
    const int s0 = lds.aoff[a], s1 = lds.aoff[a + 1];
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
    int sl = s0;
    for (; sl + 4 <= s1; sl += 4) {
      acc0 += lds.stage[sl * STG_LD + fidx];
      acc1 += lds.stage[(sl + 1) * STG_LD + fidx];
      acc2 += lds.stage[(sl + 2) * STG_LD + fidx];
      acc3 += lds.stage[(sl + 3) * STG_LD + fidx];
    }
    for (; sl < s1; ++sl) acc0 += lds.stage[sl * STG_LD + fidx];
    dst[a * ENV_LD + fidx] = scale * ((acc0 + acc1) + (acc2 + acc3));
  }
}

// ---------------------------------------------------------------------------- the kernel
// PROF: opt-in phase timing (s_memtime stamps per wave, summed into A.prof[phase]); AHIP_FUSED_PROF=1.
enum { PH_GEOM = 0, PH_TB, PH_EMB, PH_ENV, PH_TP, PH_MIX, PH_LAT, PH_OUT, PH_BLAT, PH_BMIX, PH_BTP, PH_BENV, PH_BEMB, PH_BTB, PH_FIN, PH_N };
#define PHASE(id) do { if (PROF) { long long _t = clock64(); pacc[id] += _t - tprev; tprev = _t; } } while (0)

template <bool PROF>
__global__ void __launch_bounds__(256, 1) k_fused(FusedArgs A) {
  __shared__ Lds lds;
  const int tid = threadIdx.x, lane = tid & 63, slot = lane & 31, h = lane >> 5, wave = tid >> 6;
  const int v16 = lane * 16;
  // wave-uniform buffer descriptors (made provably uniform with readfirstlane)
  __amdgpu_buffer_rsrc_t SB, WB;
  {
    unsigned long long b = (unsigned long long)(A.scratch + (size_t)blockIdx.x * A.wg_scratch + (size_t)wave * A.wave_scratch);
    unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    SB = __builtin_amdgcn_make_buffer_rsrc((void *)(((unsigned long long)hi << 32) | lo), 0, (int)(A.wave_scratch * 4), 0x00020000);
    WB = __builtin_amdgcn_make_buffer_rsrc((void *)A.wbase, 0, A.wbytes, 0x00020000);
  }
  const float *__restrict__ Wb = A.wbase;
  for (int k = tid; k < A.NL * 160; k += 256) lds.tp[k / 160][k % 160] = Wb[A.o_tpl + k];
  const int ntiles = *A.ntiles;
  const int NL = A.NL;
  double acc_part = 0.0;       // thread 0: energy; threads 64..69: virial components
  long long pacc[PH_N];
  long long tprev = 0;
  if (PROF) {
#pragma unroll
    for (int k = 0; k < PH_N; ++k) pacc[k] = 0;
    tprev = clock64();
  }
  float *const pk = lds.park[wave];
  f32x4 ring[4];                               // the weight-fragment stream (see linear_s)
  ring_prime(WB, A.o_tb_w1, v16, ring);

  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int a0 = A.tile_a0[tile], a1 = A.tile_a0[tile + 1];
    const int na = a1 - a0;
    const int e0 = A.eoff[a0], e1 = A.eoff[a1];
    if (tid <= na) lds.aoff[tid] = A.eoff[a0 + tid] - e0;
    const int s = wave * 32 + slot;
    const int e = e0 + s;
    const bool valid = e < e1;
    float *const st = lds.stage + s * STG_LD;

    // ---------------- geometry ----------------
    float rx = 1.f, ry = 0.f, rz = 0.f;
    int aloc = 0, ti = 0, tj = 0, jat = 0;
    if (valid) {
      rx = A.rvec[3 * (size_t)e]; ry = A.rvec[3 * (size_t)e + 1]; rz = A.rvec[3 * (size_t)e + 2];
      const int ii = A.e_ii[e];
      aloc = ii - a0;
      ti = A.mtype[A.ilist[ii]];
      jat = A.e_j[e];
      tj = A.mtype[jat];
    }
    const float d = sqrtf(rx * rx + ry * ry + rz * rz);
    const float inv = 1.f / d;
    const float nx = rx * inv, ny = ry * inv, nz = rz * inv;
    const float rc = (float)A.rcut[ti * A.T + tj];
    const float xx = d / rc;
    float fc, dfc_dx;
    cutoff_poly(A.p, xx, fc, dfc_dx);
    if (!valid) { fc = 0.f; dfc_dx = 0.f; }
    const float Y1 = C_S3 * ny, Y2 = C_S3 * nz, Y3 = C_S3 * nx;
    const float pref = 2.f / rc;
    const float PI = 3.14159265358979323846f;
    const float *const envrow = lds.env[0] + aloc * ENV_LD;        // + kk * MAXA*ENV_LD
    const float *const denvrow = lds.denv + aloc * ENV_LD;
    PHASE(PH_GEOM);

    // ---------------- two-body MLP ----------------
    f32x16 x[2];
    {
      f32x16 z[2], z2[2];
      {
        const float *pt = Wb + A.o_pair + (size_t)(ti * A.T + tj) * 64;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            f32x4 v = *(const f32x4 *)(pt + t * 32 + (q * 2 + h) * 4);
            z[t][4 * q] = v.x; z[t][4 * q + 1] = v.y; z[t][4 * q + 2] = v.z; z[t][4 * q + 3] = v.w;
          }
      }
      f32x16 bfin[1];
#pragma unroll
      for (int r = 0; r < 16; ++r) bfin[0][r] = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float n = (float)(r + 4 * h + 1);
        bfin[0][r] = pref * __builtin_amdgcn_sinf(0.5f * n * xx) * inv * fc;    // revolutions: sin(pi n x)
      }
      linear<1, 2, 1, true, 2>(WB, A.o_tb_wc, bfin, z, v16);          // 2 steps only: own loads, not streamed
      save_rows<2>(SB, R_Z1TB(), z, v16);
      silu_inplace<2>(z);
      linear_s<2, 2, false, true>(WB, A.o_tb_w1, A.o_tb_w2, z, z2, v16, ring, EpiSiluSave{{SB, R_Z2TB(), v16}});
      linear_s<2, 2, false, true>(WB, A.o_tb_w2, A.o_emb, z2, x, v16, ring, EpiSaveScale{{SB, R_U0(), v16}, fc});
    }
    PHASE(PH_TB);
    // ---------------- tensor embedding weights (V^0 = w0 (x) Y is rebuilt where needed) -------------
    {
      f32x16 w0[2];
      linear_s<2, 2, false, true>(WB, A.o_emb, A.o_env[0], x, w0, v16, ring, EpiSave{SB, R_W0(), v16});
    }
    __syncthreads();          // aoff visible; previous tile's LDS users done
    PHASE(PH_EMB);

    // ---------------- layers, forward ----------------
    for (int kk = 0; kk < NL; ++kk) {
      const bool last = (kk == NL - 1);
      const int RL = R_LAYER(kk);
      float *const envk = lds.env[0] + kk * (MAXA * ENV_LD);
      f32x16 V[4], V0a[1], V1a[1];
      {
        f32x16 om[2];
        linear_s<2, 2, false, true>(WB, A.o_env[kk], last ? A.o_lat0[kk] : A.o_mix[kk], x, om, v16, ring,
                                    EpiSave{SB, RL + 0, v16});
        // prefetch V^{kk} (or w0 for the first layer) now: it lands while the environment is reduced
        if (kk == 0) { load_rows<1>(SB, R_W0(), V0a, v16); load_rows<1>(SB, R_W0() + 1, V1a, v16); }
        __builtin_amdgcn_sched_barrier(0);
        // environment sum over the centre's edges
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int fidx = feat_of(r, h);
          st[fidx] = om[0][r];
          st[32 + fidx] = om[1][r] * Y1;
          st[64 + fidx] = om[1][r] * Y2;
          st[96 + fidx] = om[1][r] * Y3;
        }
        __syncthreads();
        reduce_stage(lds, envk, na, A.cenv, tid);
        __syncthreads();
      }
      if (kk == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float w1 = V1a[0][r];
          V[0][r] = V0a[0][r]; V[1][r] = w1 * Y1; V[2][r] = w1 * Y2; V[3][r] = w1 * Y3;
        }
      } else {
#pragma unroll
        for (int lm = 0; lm < 4; ++lm) park_load(pk, lm, V[lm], lane);      // V^{kk} parked by the previous layer's mix
      }
      __builtin_amdgcn_sched_barrier(0);
      PHASE(PH_ENV);
      // tensor product (4 registers per scheduling group)
      f32x16 Vp[4];
      {
        const float *en = envrow + kk * (MAXA * ENV_LD);
        const float *tp = lds.tp[kk];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const int r = 4 * q + c;
            const int fidx = feat_of(r, h);
            const float e0v = en[fidx], e1v = en[32 + fidx], e2v = en[64 + fidx], e3v = en[96 + fidx];
            const float v0 = V[0][r], v1 = V[1][r], v2 = V[2][r], v3 = V[3][r];
            Vp[0][r] = tp[fidx] * v0 * e0v + tp[32 + fidx] * C_P1 * (v1 * e1v + v2 * e2v + v3 * e3v);
            if (!last) {
              const float p2 = tp[64 + fidx], p3 = tp[96 + fidx], c4 = tp[128 + fidx] * C_P4;
              Vp[1][r] = p2 * v0 * e1v + p3 * v1 * e0v + c4 * (v2 * e3v - v3 * e2v);
              Vp[2][r] = p2 * v0 * e2v + p3 * v2 * e0v + c4 * (v3 * e1v - v1 * e3v);
              Vp[3][r] = p2 * v0 * e3v + p3 * v3 * e0v + c4 * (v1 * e2v - v2 * e1v);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      PHASE(PH_TP);
      // channel mixing -> V^{kk+1}, parked in the next layer's VIN rows
      if (!last) {
        const int mx = A.o_mix[kk];
        f32x16 in1[1], out1[1];
#pragma unroll
        for (int lm = 0; lm < 4; ++lm) {
          in1[0] = Vp[lm];
          linear_s<1, 1, false, true>(WB, lm == 0 ? mx : mx + 1024, lm == 3 ? A.o_lat0[kk] : mx + 1024, in1, out1, v16, ring,
                                      EpiSavePark{{SB, R_LAYER(kk + 1) + 8 + lm, v16}, pk, lm, lane});
        }
      }
      PHASE(PH_MIX);
      // latent MLP
      {
        f32x16 cat[3], z[2], z2[2];
        cat[0] = x[0]; cat[1] = x[1]; cat[2] = Vp[0];
        linear_s<3, 2, false, true>(WB, A.o_lat0[kk], A.o_lat1[kk], cat, z, v16, ring, EpiSiluSave{{SB, RL + 2, v16}});
        linear_s<2, 2, false, true>(WB, A.o_lat1[kk], A.o_lat2[kk], z, z2, v16, ring, EpiSiluSave{{SB, RL + 4, v16}});
        const float ra = Wb[A.o_res[kk]], rbf = Wb[A.o_res[kk] + 1] * fc;
        f32x16 xn[2];
        linear_s<2, 2, false, true>(WB, A.o_lat2[kk], last ? A.o_out0 : A.o_env[last ? kk : kk + 1], z2, xn, v16, ring,
                                    EpiResidual<2>{{SB, RL + 6, v16}, x, ra, rbf});
        x[0] = xn[0]; x[1] = xn[1];
      }
      PHASE(PH_LAT);
    }

    // ---------------- read-out ----------------
    f32x16 zr[1];
    linear_s<2, 1, false, true>(WB, A.o_out0, A.o_out0T, x, zr, v16, ring, EpiNone{});
    const f32x16 wo1 = load_hvec(WB, A.o_out1, h * 16);
    float eps = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) eps += silu1(zr[0][r]) * wo1[r];
    eps += __shfl_xor(eps, 32, 64);

    // =========================== backward ===========================
    const float deps = valid ? Wb[A.o_scale + ti] * A.cenv : 0.f;
    f32x16 dx[2], upre[2];
    load_rows<2>(SB, R_LAYER(NL - 1) + 6, upre, v16);     // u of the last layer, lands under the out0^T MFMAs
    __builtin_amdgcn_sched_barrier(0);
    {
      f32x16 dzr[1];
#pragma unroll
      for (int r = 0; r < 16; ++r) dzr[0][r] = deps * wo1[r] * dsilu1(zr[0][r]);
      linear_s<1, 2, false, true>(WB, A.o_out0T, A.o_lat2T[NL - 1], dzr, dx, v16, ring, EpiNone{});
    }
    float dfc_part = 0.f, dY1 = 0.f, dY2 = 0.f, dY3 = 0.f;
    PHASE(PH_OUT);

    for (int kk = NL - 1; kk >= 0; --kk) {
      const bool last = (kk == NL - 1);
      const int RL = R_LAYER(kk);
      f32x16 dVp[4], Vk[4], V0b[1], V1b[1];
      {
        f32x16 du[2], dh[2];
        {
          const float ra = Wb[A.o_res[kk]], rb = Wb[A.o_res[kk] + 1];
          float acc = 0.f;
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              acc += upre[t][r] * dx[t][r];
              du[t][r] = rb * fc * dx[t][r];
              dx[t][r] = ra * dx[t][r];
            }
          dfc_part += rb * acc;
        }
        f32x16 zt[2], zt1[2];
        load_rows<2>(SB, RL + 4, zt, v16);                   // z2, lands under the next 64 MFMAs
        load_rows<2>(SB, RL + 2, zt1, v16);                  // z1
        __builtin_amdgcn_sched_barrier(0);
        linear_s<2, 2, false, true>(WB, A.o_lat2T[kk], A.o_lat1T[kk], du, dh, v16, ring, EpiMulDsilu<2>{zt});
        // prefetch V^{kk} (input of this layer's tensor product) under the MFMAs that follow
        if (kk > 0) load_rows<4>(SB, RL + 8, Vk, v16);
        else { load_rows<1>(SB, R_W0(), V0b, v16); load_rows<1>(SB, R_W0() + 1, V1b, v16); }
        __builtin_amdgcn_sched_barrier(0);
        linear_s<2, 2, false, true>(WB, A.o_lat1T[kk], A.o_lat0T[kk], dh, du, v16, ring, EpiMulDsilu<2>{zt1});
        f32x16 dcat[3];
        linear_s<2, 3, false, true>(WB, A.o_lat0T[kk], last ? A.o_envT[kk] : A.o_mixT[kk], du, dcat, v16, ring, EpiNone{});
#pragma unroll
        for (int r = 0; r < 16; ++r) { dx[0][r] += dcat[0][r]; dx[1][r] += dcat[1][r]; }
        dVp[0] = dcat[2];                                   // ds
      }
      if (kk == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float w1 = V1b[0][r];
          Vk[0][r] = V0b[0][r]; Vk[1][r] = w1 * Y1; Vk[2][r] = w1 * Y2; Vk[3][r] = w1 * Y3;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      PHASE(PH_BLAT);
      if (!last) {
        const int mx = A.o_mixT[kk];
        f32x16 in1[1], out1[1];
#pragma unroll
        for (int lm = 0; lm < 4; ++lm) {
          park_load(pk, lm, in1[0], lane);
          linear_s<1, 1, false, true>(WB, lm == 0 ? mx : mx + 1024, lm == 3 ? A.o_envT[kk] : mx + 1024, in1, out1, v16, ring, EpiNone{});
          if (lm == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) dVp[0][r] += out1[0][r];
          } else dVp[lm] = out1[0];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      PHASE(PH_BMIX);
      // tensor-product backward: dV (w.r.t. V^{kk}, parked) and the per-edge environment gradient
      f32x16 om[2];
      {
        const float *en = envrow + kk * (MAXA * ENV_LD);
        const float *tp = lds.tp[kk];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 o0, o1, o2, o3;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const int r = 4 * q + c;
            const int fidx = feat_of(r, h);
            const float e0v = en[fidx], e1v = en[32 + fidx], e2v = en[64 + fidx], e3v = en[96 + fidx];
            const float v0 = Vk[0][r], v1 = Vk[1][r], v2 = Vk[2][r], v3 = Vk[3][r];
            const float g0 = dVp[0][r];
            const float q0 = tp[fidx] * g0, q1 = tp[32 + fidx] * C_P1 * g0;
            float a0v = q0 * e0v, a1v = q1 * e1v, a2v = q1 * e2v, a3v = q1 * e3v;       // dV
            float b0v = q0 * v0, b1v = q1 * v1, b2v = q1 * v2, b3v = q1 * v3;           // denv_e
            if (!last) {
              const float g1 = dVp[1][r], g2 = dVp[2][r], g3 = dVp[3][r];
              const float q2 = tp[64 + fidx], q3 = tp[96 + fidx], c4 = tp[128 + fidx] * C_P4;
              a0v += q2 * (e1v * g1 + e2v * g2 + e3v * g3);
              b0v += q3 * (v1 * g1 + v2 * g2 + v3 * g3);
              a1v += q3 * e0v * g1 + c4 * (e2v * g3 - e3v * g2);      // (e x g)_1
              a2v += q3 * e0v * g2 + c4 * (e3v * g1 - e1v * g3);
              a3v += q3 * e0v * g3 + c4 * (e1v * g2 - e2v * g1);
              b1v += q2 * v0 * g1 + c4 * (g2 * v3 - g3 * v2);         // (g x v)_1
              b2v += q2 * v0 * g2 + c4 * (g3 * v1 - g1 * v3);
              b3v += q2 * v0 * g3 + c4 * (g1 * v2 - g2 * v1);
            }
            o0[c] = a0v; o1[c] = a1v; o2[c] = a2v; o3[c] = a3v;
            st[fidx] = b0v; st[32 + fidx] = b1v; st[64 + fidx] = b2v; st[96 + fidx] = b3v;
          }
          *(f32x4 *)(pk + 0 * ROW + q * 256 + lane * 4) = o0;
          *(f32x4 *)(pk + 1 * ROW + q * 256 + lane * 4) = o1;
          *(f32x4 *)(pk + 2 * ROW + q * 256 + lane * 4) = o2;
          *(f32x4 *)(pk + 3 * ROW + q * 256 + lane * 4) = o3;
          __builtin_amdgcn_sched_barrier(0);
        }
        load_rows<2>(SB, RL + 0, om, v16);                   // omega of this layer, lands during the reduction
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        reduce_stage(lds, lds.denv, na, A.cenv, tid);
        __syncthreads();
      }
      __builtin_amdgcn_sched_barrier(0);
      PHASE(PH_BTP);
      {
        f32x16 dom[2];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const int r = 4 * q + c;
            const int fidx = feat_of(r, h);
            const float d0 = denvrow[fidx], d1 = denvrow[32 + fidx], d2 = denvrow[64 + fidx], d3 = denvrow[96 + fidx];
            dom[0][r] = d0;
            dom[1][r] = d1 * Y1 + d2 * Y2 + d3 * Y3;
            dY1 += d1 * om[1][r]; dY2 += d2 * om[1][r]; dY3 += d3 * om[1][r];
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        if (kk > 0) load_rows<2>(SB, R_LAYER(kk - 1) + 6, upre, v16);     // next iteration's u rows
        __builtin_amdgcn_sched_barrier(0);
        linear_s<2, 2, true, true>(WB, A.o_envT[kk], kk > 0 ? A.o_lat2T[kk > 0 ? kk - 1 : 0] : A.o_embT, dom, dx, v16, ring, EpiNone{});
      }
      PHASE(PH_BENV);
    }
    // ---------------- embedding backward ----------------
    {
      f32x16 dV[4], w0[2], dw0[2];
      load_rows<2>(SB, R_W0(), w0, v16);
#pragma unroll
      for (int lm = 0; lm < 4; ++lm) park_load(pk, lm, dV[lm], lane);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        dw0[0][r] = dV[0][r];
        dw0[1][r] = dV[1][r] * Y1 + dV[2][r] * Y2 + dV[3][r] * Y3;
        dY1 += dV[1][r] * w0[1][r]; dY2 += dV[2][r] * w0[1][r]; dY3 += dV[3][r] * w0[1][r];
      }
      linear_s<2, 2, true, true>(WB, A.o_embT, A.o_tb_w2T, dw0, dx, v16, ring, EpiNone{});
    }
    PHASE(PH_BEMB);
    // ---------------- two-body MLP backward ----------------
    float dd_part = 0.f;
    {
      f32x16 du[2], dh[2];
      float acc = 0.f;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 u = bload(SB, v16, ((R_U0() + t) * ROW + q * 256) * 4);
          const float uu[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
          for (int c = 0; c < 4; ++c) { acc += uu[c] * dx[t][4 * q + c]; du[t][4 * q + c] = fc * dx[t][4 * q + c]; }
        }
      dfc_part += acc;
      f32x16 zt[2], zt1[2];
      load_rows<2>(SB, R_Z2TB(), zt, v16);
      load_rows<2>(SB, R_Z1TB(), zt1, v16);
      __builtin_amdgcn_sched_barrier(0);
      linear_s<2, 2, false, true>(WB, A.o_tb_w2T, A.o_tb_w1T, du, dh, v16, ring, EpiMulDsilu<2>{zt});
      linear_s<2, 2, false, true>(WB, A.o_tb_w1T, A.o_tb_wcT, dh, du, v16, ring, EpiMulDsilu<2>{zt1});
      f32x16 dbf[1];
      linear_s<2, 1, false, true>(WB, A.o_tb_wcT, A.o_tb_w1, du, dbf, v16, ring, EpiNone{});   // next = next tile's first linear
      const float dfdd = dfc_dx / rc;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float n = (float)(r + 4 * h + 1);
        // argument in revolutions for v_sin/v_cos (|arg| <= 4): abs error ~1e-6, far inside the force budget
        const float sn = __builtin_amdgcn_sinf(0.5f * n * xx), cs = __builtin_amdgcn_cosf(0.5f * n * xx);
        const float b = pref * sn * inv;
        const float db = pref * (cs * PI * n / rc * inv - sn * inv * inv);
        dd_part += dbf[0][r] * (db * fc + b * dfdd);
      }
    }
    PHASE(PH_BTB);
    // ---------------- geometry backward, outputs ----------------
    {
      const float dfc_tot = dfc_part + __shfl_xor(dfc_part, 32, 64);
      const float dd = dfc_tot * (dfc_dx / rc) + dd_part + __shfl_xor(dd_part, 32, 64);
      const float y1 = dY1 + __shfl_xor(dY1, 32, 64), y2 = dY2 + __shfl_xor(dY2, 32, 64), y3 = dY3 + __shfl_xor(dY3, 32, 64);
      const float Gx = C_S3 * y3, Gy = C_S3 * y1, Gz = C_S3 * y2;
      const float gn = Gx * nx + Gy * ny + Gz * nz;
      const float gx = dd * nx + (Gx - gn * nx) * inv;
      const float gy = dd * ny + (Gy - gn * ny) * inv;
      const float gz = dd * nz + (Gz - gn * nz) * inv;
      if (A.dbg && valid && h == 0) {
        float *dp = A.dbg + 8 * (size_t)e;
        dp[0] = gx; dp[1] = gy; dp[2] = gz; dp[3] = dd; dp[4] = dfc_tot; dp[5] = y1; dp[6] = y2; dp[7] = y3;
      }
      const float m = valid ? 1.f : 0.f;
      if (h == 0) {
        st[0] = m * gx; st[1] = m * gy; st[2] = m * gz; st[3] = m * eps;
        if (valid) {
          atomicAdd(&A.f[3 * (size_t)jat], -(double)gx);
          atomicAdd(&A.f[3 * (size_t)jat + 1], -(double)gy);
          atomicAdd(&A.f[3 * (size_t)jat + 2], -(double)gz);
        }
      }
      // virial of this wave's 32 edges: butterfly over the slot lanes, lane 0 publishes 6 partials
      float w6[6] = {-m * rx * gx, -m * ry * gy, -m * rz * gz, -m * 0.5f * (rx * gy + ry * gx),
                     -m * 0.5f * (rx * gz + rz * gx), -m * 0.5f * (ry * gz + rz * gy)};
#pragma unroll
      for (int c = 0; c < 6; ++c) {
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) w6[c] += __shfl_xor(w6[c], off, 64);
      }
      if (lane == 0) {
#pragma unroll
        for (int c = 0; c < 6; ++c) lds.vir[wave][c] = w6[c];
      }
    }
    __syncthreads();
    {
      // per-centre sums of (g, eps): 16 lanes per atom = 4 columns x 4 row-parts
      const int a = tid >> 4, col = tid & 3, part = (tid >> 2) & 3;
      float sum = 0.f;
      if (a < na)
        for (int sl = lds.aoff[a] + part; sl < lds.aoff[a + 1]; sl += 4) sum += lds.stage[sl * STG_LD + col];
      sum += __shfl_xor(sum, 4, 64);
      sum += __shfl_xor(sum, 8, 64);
      if (a < na && part == 0) {
        const int i = A.ilist[a0 + a];
        if (col < 3) atomicAdd(&A.f[3 * (size_t)i + col], (double)sum);
        else {
          const int t = A.mtype[i];
          const float ei = Wb[A.o_scale + t] * (sum * A.cenv) + Wb[A.o_shift + t];
          if (A.eatom) A.eatom[i] = (double)ei;
          lds.ea[a] = ei;
        }
      }
    }
    __syncthreads();
    if (tid == 0) {
      double se = 0;
      for (int a = 0; a < na; ++a) se += lds.ea[a];
      acc_part += se;
    } else if (tid >= 64 && tid < 70) {
      const int c = tid - 64;
      acc_part += (double)((lds.vir[0][c] + lds.vir[1][c]) + (lds.vir[2][c] + lds.vir[3][c]));
    }
    __syncthreads();
    PHASE(PH_FIN);
  }
  if (PROF && lane == 0) {
#pragma unroll
    for (int k = 0; k < PH_N; ++k) atomicAdd((unsigned long long *)&A.prof[k], (unsigned long long)pacc[k]);
  }
  if (tid == 0) A.partial[7 * (size_t)blockIdx.x] = acc_part;
  if (tid >= 64 && tid < 70) A.partial[7 * (size_t)blockIdx.x + 1 + (tid - 64)] = acc_part;
}

// ---------------------------------------------------------------------------- tile packing
// Greedy packing of consecutive centre atoms into tiles (<= 128 edges, <= MAXA atoms), done
// sequentially inside independent segments of SEG atoms so it parallelises.
template <bool FILL>
__global__ void k_pack_tiles(int inum, const int *eoff, int nseg, int *seg_count, const int *seg_base, int *tile_a0) {
  int sg = blockIdx.x * blockDim.x + threadIdx.x;
  if (sg >= nseg) return;
  int a = sg * SEG, end = min(inum, a + SEG);
  int nt = 0, cur_e = 0, cur_a = 0;
  int base = FILL ? seg_base[sg] : 0;
  if (FILL && a < end) tile_a0[base] = a;
  for (int at = a; at < end; ++at) {
    int deg = eoff[at + 1] - eoff[at];
    if (cur_a == MAXA || cur_e + deg > TILE_SLOTS) {
      ++nt; cur_e = 0; cur_a = 0;
      if (FILL) tile_a0[base + nt] = at;
    }
    cur_e += deg; ++cur_a;
  }
  if (!FILL) seg_count[sg] = (a < end) ? nt + 1 : 0;
}
__global__ void k_pack_finish(int inum, int nseg, const int *seg_base, int *tile_a0, int *ntiles) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    int n = seg_base[nseg];
    tile_a0[n] = inum;
    *ntiles = n;
  }
}

// ---------------------------------------------------------------------------- host side
struct FusedState {
  DevBuf wbuf, scratch, seg_count, seg_base, tile_a0, ntiles, partial;
  FusedArgs args;
  bool ready = false, prof_on = false, dbg_on = false;
  DevBuf prof, dbg;
  int ncu = 256;
  int grid = 256;
  int occ = 2;                 // resident workgroups per CU (AHIP_FUSED_OCC=1|2)
};

// A-operand fragments of W [K][N] (row-major, x @ W): [ot][kt][q][lane][c], r = 4q+c,
//   value = W[32 kt + feat(r, lane>>5)][32 ot + (lane & 31)], zero padded.
static void append_frag(std::vector<float> &out, const double *W, int K, int N, int ldw) {
  const int KT = (K + 31) / 32, NT = (N + 31) / 32;
  for (int ot = 0; ot < NT; ++ot)
    for (int kt = 0; kt < KT; ++kt)
      for (int q = 0; q < 4; ++q)
        for (int lane = 0; lane < 64; ++lane)
          for (int c = 0; c < 4; ++c) {
            int r = 4 * q + c;
            int k = 32 * kt + feat_of(r, lane >> 5), n = 32 * ot + (lane & 31);
            out.push_back((k < K && n < N) ? (float)W[(size_t)k * ldw + n] : 0.f);
          }
}
static std::vector<double> transpose(const double *W, int K, int N) {
  std::vector<double> t((size_t)K * N);
  for (int k = 0; k < K; ++k)
    for (int n = 0; n < N; ++n) t[(size_t)n * K + k] = W[(size_t)k * N + n];
  return t;
}
// per-half vector [q][h][4] of v[0..32)
static void append_hvec(std::vector<float> &out, const double *v) {
  for (int q = 0; q < 4; ++q)
    for (int h = 0; h < 2; ++h)
      for (int c = 0; c < 4; ++c) out.push_back((float)v[feat_of(4 * q + c, h)]);
}

bool fused_model_supported(const Model &m, std::string *why) {
  const HostModel &h = m.hm;
  auto no = [&](const char *msg) { if (why) *why = msg; return false; };
  if (h.l_max != 1) return no("fused kernels need l_max = 1");
  if (h.U != 32 || h.S != 64 || h.mlp_width != 64 || h.readout_width != 32) return no("fused kernels need U=32, S=64, MLP width 64, read-out width 32");
  if (h.mlp_depth != 2 || h.readout_depth != 1) return no("fused kernels need MLP depth 2 and read-out depth 1");
  if (h.num_bessels != 8) return no("fused kernels need 8 Bessel functions");
  if (h.num_layers < 1 || h.num_layers > MAXNL) return no("fused kernels need 1..3 layers");
  if (h.num_types > 4) return no("fused kernels support at most 4 model types");
  return true;
}

static void fused_prepare(Model &m) {
  if (!m.fused_state) m.fused_state = new FusedState();
  FusedState &st = *(FusedState *)m.fused_state;
  if (st.ready) return;
  const HostModel &h = m.hm;
  const int T = h.num_types, NL = h.num_layers;
  std::vector<float> w;
  FusedArgs &A = st.args;
  std::memset(&A, 0, sizeof(A));
  auto mark = [&]() { while (w.size() % 4) w.push_back(0.f); return (int)w.size(); };
  // two-body: pair table + Bessel block
  {
    const HostTensor &w0 = h.get("tb.w0");          // [2T+8][64]
    A.o_pair = mark();
    for (int ti = 0; ti < T; ++ti)
      for (int tj = 0; tj < T; ++tj) {
        std::vector<double> row(64);
        for (int n = 0; n < 64; ++n) row[n] = w0.data[(size_t)ti * 64 + n] + w0.data[(size_t)(T + tj) * 64 + n];
        append_hvec(w, row.data());
        append_hvec(w, row.data() + 32);
      }
    const double *wc = w0.data.data() + (size_t)2 * T * 64;       // [8][64]
    A.o_tb_wc = mark(); append_frag(w, wc, 8, 64, 64);
    std::vector<double> wcT = transpose(wc, 8, 64);               // [64][8]
    A.o_tb_wcT = mark(); append_frag(w, wcT.data(), 64, 8, 8);
  }
  auto both = [&](const std::string &name, int K, int N, int &of, int &ofT) {
    const HostTensor &t = h.get(name);
    of = mark(); append_frag(w, t.data.data(), K, N, N);
    std::vector<double> tt = transpose(t.data.data(), K, N);
    ofT = mark(); append_frag(w, tt.data(), N, K, K);
  };
  both("tb.w1", 64, 64, A.o_tb_w1, A.o_tb_w1T);
  both("tb.w2", 64, 64, A.o_tb_w2, A.o_tb_w2T);
  both("emb.w", 64, 64, A.o_emb, A.o_embT);
  for (int k = 0; k < NL; ++k) {
    const std::string lk = "l" + std::to_string(k + 1);
    both(lk + ".env", 64, 64, A.o_env[k], A.o_envT[k]);
    both(lk + ".lat.w0", 96, 64, A.o_lat0[k], A.o_lat0T[k]);
    both(lk + ".lat.w1", 64, 64, A.o_lat1[k], A.o_lat1T[k]);
    both(lk + ".lat.w2", 64, 64, A.o_lat2[k], A.o_lat2T[k]);
    const HostTensor &tp = h.get(lk + ".tp");
    A.o_tp[k] = mark();
    for (int p = 0; p < 5; ++p) {
      std::vector<double> row(32, 0.0);
      if (p < tp.shape[0]) for (int u = 0; u < 32; ++u) row[u] = tp.data[(size_t)p * 32 + u];
      append_hvec(w, row.data());
    }
    const HostTensor &res = h.get(lk + ".res");
    A.o_res[k] = mark(); w.push_back((float)res.data[0]); w.push_back((float)res.data[1]);
    if (k < NL - 1) {
      const HostTensor &mx = h.get(lk + ".mix");           // [2][32][32]
      A.o_mix[k] = mark();
      for (int l = 0; l < 2; ++l) append_frag(w, mx.data.data() + (size_t)l * 1024, 32, 32, 32);
      A.o_mixT[k] = mark();
      for (int l = 0; l < 2; ++l) { auto t = transpose(mx.data.data() + (size_t)l * 1024, 32, 32); append_frag(w, t.data(), 32, 32, 32); }
    }
  }
  A.o_tpl = mark();
  for (int k = 0; k < NL; ++k) {
    const HostTensor &tp = h.get("l" + std::to_string(k + 1) + ".tp");
    for (int p = 0; p < 5; ++p)
      for (int u = 0; u < 32; ++u) w.push_back(p < tp.shape[0] ? (float)tp.data[(size_t)p * 32 + u] : 0.f);
  }
  both("out.w0", 64, 32, A.o_out0, A.o_out0T);
  A.o_out1 = mark(); append_hvec(w, h.get("out.w1").data.data());
  A.o_scale = mark(); for (int t = 0; t < T; ++t) w.push_back((float)h.get("scale").data[t]);
  A.o_shift = mark(); for (int t = 0; t < T; ++t) w.push_back((float)h.get("shift").data[t]);
  st.wbuf.reserve(w.size() * sizeof(float));
  AHIP_CHECK(hipMemcpy(st.wbuf.p, w.data(), w.size() * sizeof(float), hipMemcpyHostToDevice));
  A.wbase = st.wbuf.as<float>();
  A.wbytes = (int)(w.size() * sizeof(float));
  A.T = T; A.NL = NL; A.p = h.poly_p;
  A.cenv = (float)(1.0 / std::sqrt(h.avg_num_neighbors));
  hipDeviceProp_t prop;
  AHIP_CHECK(hipGetDeviceProperties(&prop, m.device));
  st.ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  // One persistent workgroup per CU = one wave per SIMD with the whole 512-entry register file.
  // (Two workgroups per CU at <= 256 registers was measured: the tensor-product phases spill and
  //  the kernel is 1.7x slower -- profiles/r01_b.)
  st.occ = 1;
  st.grid = st.ncu;
  A.wave_scratch = (long long)R_TOTAL(NL) * ROW;
  A.wg_scratch = 4 * A.wave_scratch;
  st.scratch.reserve((size_t)st.grid * A.wg_scratch * sizeof(float));
  A.scratch = st.scratch.as<float>();
  st.partial.reserve((size_t)st.grid * 7 * sizeof(double));
  st.ntiles.reserve(64);
  st.prof.reserve(64 * sizeof(long long));
  const char *pe = std::getenv("AHIP_FUSED_PROF");
  st.prof_on = pe && pe[0] == '1';
  const char *de = std::getenv("AHIP_FUSED_DBG");
  st.dbg_on = de && de[0] == '1';
  st.ready = true;
}

bool fused_run(Model &m, const ComputeArgs &a, std::string *why) {
  if (m.last_max_deg > TILE_SLOTS) {
    if (why) *why = "an atom has " + std::to_string(m.last_max_deg) + " edges (> 128 per tile)";
    return false;
  }
  if (m.edges_T_size != 4) { if (why) *why = "edge vectors are not float32"; return false; }
  fused_prepare(m);
  FusedState &st = *(FusedState *)m.fused_state;
  hipStream_t s = a.stream;
  const int inum = m.inum;
  const int nseg = (inum + SEG - 1) / SEG;
  st.seg_count.reserve((size_t)(nseg + 1) * sizeof(int));
  st.seg_base.reserve((size_t)(nseg + 2) * sizeof(int));
  st.tile_a0.reserve((size_t)(inum + nseg + 2) * sizeof(int));
  {
    StageTimer tm(m, "tile_pack", s);
    const unsigned B = 64;
    hipLaunchKernelGGL(k_pack_tiles<false>, dim3((nseg + B - 1) / B), dim3(B), 0, s, inum, m.b_eoff.as<int>(), nseg, st.seg_count.as<int>(), (const int *)nullptr, (int *)nullptr);
    AHIP_CHECK(prim_exclusive_scan_i32(st.seg_count.as<int>(), st.seg_base.as<int>(), nseg, s));
    hipLaunchKernelGGL(k_pack_tiles<true>, dim3((nseg + B - 1) / B), dim3(B), 0, s, inum, m.b_eoff.as<int>(), nseg, (int *)nullptr, st.seg_base.as<int>(), st.tile_a0.as<int>());
    hipLaunchKernelGGL(k_pack_finish, dim3(1), dim3(1), 0, s, inum, nseg, st.seg_base.as<int>(), st.tile_a0.as<int>(), st.ntiles.as<int>());
  }
  FusedArgs A = st.args;
  A.eoff = m.b_eoff.as<int>(); A.e_ii = m.b_eii.as<int>(); A.e_j = m.b_ej.as<int>();
  A.ilist = m.d_ilist; A.mtype = a.mtype; A.rvec = m.b_rvec.as<float>(); A.rcut = m.rcut_model_dev;
  A.tile_a0 = st.tile_a0.as<int>(); A.ntiles = st.ntiles.as<int>();
  A.f = a.f; A.eatom = a.eatom; A.partial = st.partial.as<double>();
  {
    StageTimer tm(m, "model_fused", s);
    if (st.dbg_on) {
      st.dbg.reserve((size_t)std::max<long long>(m.nedges, 1) * 8 * sizeof(float));
      AHIP_CHECK(hipMemsetAsync(st.dbg.p, 0, (size_t)m.nedges * 8 * sizeof(float), s));
      A.dbg = st.dbg.as<float>();
    }
    if (st.prof_on) {
      AHIP_CHECK(hipMemsetAsync(st.prof.p, 0, 64 * sizeof(long long), s));
      A.prof = st.prof.as<long long>();
      hipLaunchKernelGGL(k_fused<true>, dim3(st.grid), dim3(256), 0, s, A);
    } else {
      hipLaunchKernelGGL(k_fused<false>, dim3(st.grid), dim3(256), 0, s, A);
    }
  }
  AHIP_CHECK(hipGetLastError());
  AHIP_CHECK(prim_sum_columns_f64(st.partial.as<double>(), st.grid, 7, a.engvir, s));
  if (st.prof_on) {
    long long hp[PH_N];
    AHIP_CHECK(hipMemcpyAsync(hp, st.prof.p, sizeof(hp), hipMemcpyDeviceToHost, s));
    AHIP_CHECK(hipStreamSynchronize(s));
    static const char *names[PH_N] = {"geom", "tb_mlp", "embed", "env+reduce", "tp", "mix", "latent_mlp", "readout", "b_latent", "b_mix", "b_tp+reduce", "b_env", "b_embed", "b_tb", "finish"};
    double tot = 0;
    for (int k = 0; k < PH_N; ++k) tot += (double)hp[k];
    std::fprintf(stderr, "[ahip fused prof] wave-cycles by phase (sum over %d waves):", st.grid * 4);
    for (int k = 0; k < PH_N; ++k) std::fprintf(stderr, " %s=%.1f%%", names[k], 100.0 * hp[k] / tot);
    std::fprintf(stderr, " | total=%.3g cycles\n", tot);
  }
  return true;
}

void fused_free(Model &m) {
  if (!m.fused_state) return;
  FusedState *st = (FusedState *)m.fused_state;
  for (DevBuf *b : {&st->wbuf, &st->scratch, &st->seg_count, &st->seg_base, &st->tile_a0, &st->ntiles, &st->partial, &st->prof}) b->release();
  delete st;
  m.fused_state = nullptr;
}

// ---------------------------------------------------------------------------- diagnostic
// Single-wave self-test of the register-chain linear primitive: out[32][N] = in[32][K] @ W[K][N].
template <int KT, int NT, int KQ>
__global__ void __launch_bounds__(64) k_selftest_linear(const float *Wf, const float *in, int K, float *out, int N) {
  const int lane = threadIdx.x, slot = lane & 31, h = lane >> 5;
  f32x16 a[KT], o[NT];
#pragma unroll
  for (int t = 0; t < KT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int k = 32 * t + feat_of(r, h);
      a[t][r] = k < K ? in[slot * K + k] : 0.f;
    }
  __amdgpu_buffer_rsrc_t WB = __builtin_amdgcn_make_buffer_rsrc((void *)Wf, 0, KT * NT * 4096, 0x00020000);
  linear<KT, NT, KQ>(WB, 0, a, o, lane * 16);
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int n = 32 * t + feat_of(r, h);
      if (n < N) out[slot * N + n] = o[t][r];
    }
}

}  // namespace ahip

using namespace ahip;

// Diagnostic (AHIP_FUSED_DBG=1): per-edge {g[3], dd, dfc, dY[3]} of the last fused compute.
extern "C" int ahip_debug_fused_edges(ahip_model *mh, float *out, long long nedges) {
  ahip::Model *m = (ahip::Model *)mh;
  if (!m || !m->fused_state) return AHIP_ERR_STATE;
  FusedState &st = *(FusedState *)m->fused_state;
  if (!st.dbg_on || !st.dbg.p || nedges != m->nedges) return AHIP_ERR_STATE;
  if (hipDeviceSynchronize() != hipSuccess) return AHIP_ERR_DEVICE;
  return hipMemcpy(out, st.dbg.p, (size_t)nedges * 8 * sizeof(float), hipMemcpyDeviceToHost) == hipSuccess ? 0 : AHIP_ERR_DEVICE;
}

extern "C" int ahip_debug_fused_linear(int K, int N, const double *W, const float *in, float *out) {
  try {
    std::vector<float> frag;
    append_frag(frag, W, K, N, N);
    float *dW = nullptr, *din = nullptr, *dout = nullptr;
    AHIP_CHECK(hipMalloc((void **)&dW, frag.size() * sizeof(float)));
    AHIP_CHECK(hipMalloc((void **)&din, (size_t)32 * K * sizeof(float)));
    AHIP_CHECK(hipMalloc((void **)&dout, (size_t)32 * N * sizeof(float)));
    AHIP_CHECK(hipMemcpy(dW, frag.data(), frag.size() * sizeof(float), hipMemcpyHostToDevice));
    AHIP_CHECK(hipMemcpy(din, in, (size_t)32 * K * sizeof(float), hipMemcpyHostToDevice));
    const int KT = (K + 31) / 32, NT = (N + 31) / 32;
    bool ok = true;
#define CASE(kt, nt, kq) hipLaunchKernelGGL((k_selftest_linear<kt, nt, kq>), dim3(1), dim3(64), 0, 0, dW, din, K, dout, N)
    if (K == 8 && NT == 2) CASE(1, 2, 1);
    else if (KT == 1 && NT == 1) CASE(1, 1, 4);
    else if (KT == 1 && NT == 2) CASE(1, 2, 4);
    else if (KT == 2 && NT == 1) CASE(2, 1, 4);
    else if (KT == 2 && NT == 2) CASE(2, 2, 4);
    else if (KT == 3 && NT == 2) CASE(3, 2, 4);
    else if (KT == 2 && NT == 3) CASE(2, 3, 4);
    else ok = false;
#undef CASE
    if (ok) {
      AHIP_CHECK(hipDeviceSynchronize());
      AHIP_CHECK(hipMemcpy(out, dout, (size_t)32 * N * sizeof(float), hipMemcpyDeviceToHost));
    }
    (void)hipFree(dW); (void)hipFree(din); (void)hipFree(dout);
    return ok ? 0 : AHIP_ERR_UNSUPPORTED;
  } catch (const std::exception &) { return AHIP_ERR_DEVICE; }
}
