// Wave-pair fused MFMA kernel for l_max = 2, 64 tensor features (BASELINE config 5's model L): the same model graph and the same
// hand-derived backward as fused_lx.hip, restructured so that TWO waves run per SIMD.
//
// fused_lx.hip keeps the whole edge tensor V[lm][u] (9 x 64 floats per edge = 144 registers per lane) in one wave, which then needs
// the whole register file of its SIMD: a lone wave exposes every memory, LDS and barrier wait (profiles/r02_c_fused_lx_modelL.md).
// Here a workgroup is 8 waves = 4 wave PAIRS; a pair shares 16 edge slots and each wave of the pair owns 32 of the 64 tensor
// channels (2 of the 4 K-tiles): 72 registers of V per lane, <= 256 registers per wave, two waves per SIMD.
//
//  register budget per lane (f32 registers), forward / backward peak:
//     V (own 2 K-tiles, parked in AGPRs)                72
//     latent x or its gradient (all 64, replicated)     16
//     weight-fragment ring (8 x 16 B)                   32   (dropped in the tensor-product phases)
//     Y, dY, geometry, indices                          ~40
//     linear inputs / accumulators / epilogue rows      ~60
//     tensor product half pass (9+9+9 half rows)        ~54  (instead of the ring)
//
//  * Channel-wise work needs no exchange: the embedding / environment linears are split by OUTPUT tile (each wave computes the
//    weights of its own channels from the replicated latent x), the environment staging, the tensor product and its gradient act on
//    the wave's own K-tiles.
//  * Channel-coupling work exchanges 16-feature register images through LDS (xch: 2 x 2 tiles per wave, double-buffered; write,
//    workgroup barrier, read the partner's):  the latent MLP is split by output tile (hand-overs: tensor-product scalars, z1, z2,
//    new x), the channel mixing V[lm] <- V[lm] M_l likewise (one hand-over of the row's two own tiles per (l, m)), the transposed
//    environment / embedding linears are split by INPUT tile: each wave accumulates a partial latent gradient P (sum over the pair
//    = the gradient), combined by a reduce-scatter + all-gather of two tiles each.  Both waves hold the latent in "own tiles first"
//    order (local tile k = global tile (k + 2 half) mod 4); the host lays out each half's weight stream accordingly, so every
//    register index in the kernel is static.
//  * The read-out MLP (1 % of the MFMAs) is evaluated by both waves instead of exchanged; per-edge scalars that are sums over channels
//    (dE/dY, the cutoff and distance derivatives) are linear in the partials and meet in LDS once per tile; the first wave of a pair
//    writes the forces.
// Reference graph: the TorchScript model executed at /root/reference/pair_nequip_allegro.cpp:409-430; oracle: oracle/allegro_torch.py.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/allegro_hip.h"
#ifndef AHIP_ROW_AUX
#define AHIP_ROW_AUX 0          // saved rows: default cache policy.  (fused_lx.hip streams them non-temporally, which paid while that kernel spilled 1.4 KB per lane; here, with 50 dwords
                                // of spill traffic per wave-tile, the default policy is 4 % faster: 25.65 vs 26.8 ms on the 41k-atom water box; sc0 26.65, sc0 + nt 26.9)
#endif
#define AHIP_NO_ACC_PARK 1
#include "engine.h"
#include "fused_lx_common.h"
#include "prims.h"

namespace ahip {

struct ShapeP {
  static constexpr int L = 2, D = 9, NLP = 3, UT = 4, HT = 2, U = 64, EWH = NLP * HT;    // EWH: own 16-feature tiles of an (l, u) weight vector
  static constexpr int NW = 8, SLOTS = 64, MAXA = 4;
  static constexpr int STG_LD = D * 16 + 4;             // one K-tile of a slot: [lm][16] + pad
  static constexpr int ENVA = D * U + 16;               // environment row of one centre: [lm][u] + pad.  ENVA = 16 (mod 64): the rows of the (<= 4) centres of a tile and the four lane groups (+ 4 g)
                                                        // start in 16 distinct 4-bank groups (round 6; with + 4 centre 1 / group 0 met centre 0 / group 1: 229.1 -> 227.7 ms on config 5)
  static constexpr int NP = CgX<2>::NP;
  // scratch rows (per wave, 1 KiB each): d x0/dd 4 | w0 EWH | per layer: omega EWH, silu'(z1) 2, silu'(z2) 2, u 2, V_in D*HT
  static constexpr int R_DX0 = 0, R_W0 = 4, LSZ = EWH + 6 + D * HT;
  __host__ __device__ static constexpr int R_LAYER(int kk) { return 4 + EWH + kk * LSZ; }
  __host__ __device__ static constexpr int R_TOTAL(int NL) { return 4 + EWH + NL * LSZ; }
  static constexpr int O_OM = 0, O_Z1 = EWH, O_Z2 = EWH + 2, O_U = EWH + 4, O_VIN = EWH + 6;
};

struct __attribute__((aligned(16))) LdsP {
  using S = ShapeP;
  float stage[2][S::SLOTS * S::STG_LD];      // [wave half][slot][lm][16]: one K-tile of each half
  float zero16[4];                           // zeros: where the per-centre reduction's reads past a centre's last slot land
  float env[LX_MAXNL][S::MAXA * S::ENVA];
  float denv[S::MAXA * S::ENVA];
  float tp[LX_MAXNL][S::NP * S::U];          // tensor-product path weights [layer][path][u]
  float xch[S::NW][2][2 * ROW];              // pair hand-over: [wave][buffer][2 register images]
  int aloc[64];                              // centre slot of every edge slot (parked here instead of a register that lives a whole tile)
  float ych[4][16][12];                      // per-edge channel sums of the second wave of a pair: dY[1..8], cutoff and distance parts
  double eacc[S::MAXA];
  double virw[4][6];
  int aoff[2][S::MAXA + 2];
  float rc[256];                             // model cutoff table [T*T], T <= 16 (the edge build packs a type in 4 bits)
  float scale[16], shift[16];
  float res[LX_MAXNL][2];
  int chunk[2];
};
static_assert(sizeof(LdsP) <= 160 * 1024, "LDS budget of one CU");

__device__ __forceinline__ unsigned lds_addr(const void *p) { return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const void *)p; }
// Per-centre sum of the two staged K-tiles (one per wave half): env[a][lm][16 (2 h + t) + f] = scale * sum_{slots of a} stage[h][slot][lm][f].
// Work item = (centre, half, 4-feature column); 4 lanes per item take every 4th slot (see reduce_stage_x in fused_lx.hip for the lane mapping).
// A lane's reads past its centre's last slot go to a 16-byte block of zeros behind the staging area (one select per read) instead of
// being masked out of the sum value by value: 105 instead of 216 vector instructions per call, twelve calls per tile.
__device__ __forceinline__ void reduce_stage_p(const float *stg, const float *zero16, const int *aoff, float *dst, int na, float scale, int t, int uwave) {
  using S = ShapeP;
  constexpr int LPI = 4, NC1 = S::D * 4, NC = 2 * NC1, PER_ROUND = S::NW * 64 / LPI, NRD = S::SLOTS / LPI;
  const int lane = fresh_lane();
  const int p = lane >> 4;
  for (int it = uwave * 16 + (lane & 15); it < ((na * NC + PER_ROUND - 1) / PER_ROUND) * PER_ROUND; it += PER_ROUND) {   // whole waves iterate together
    const bool live = it < na * NC;
    const int a = live ? it / NC : 0, c2 = live ? it - a * NC : 0;
    const int h = c2 >= NC1 ? 1 : 0, c = c2 - h * NC1;
    const int s0 = aoff[a] + p, s1 = live ? aoff[a + 1] : 0;
    const float *sh = stg + h * (S::SLOTS * S::STG_LD) + 4 * c + s0 * S::STG_LD;
    const int cnt = (s1 - s0 + LPI - 1) >> 2;          // reads of this lane that fall inside the centre's slots (<= 0: none)
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    // batches of RB reads in flight, issued and waited for by hand: left to itself the scheduler (short of registers here) waits for
    // every read before it issues the next one, 16 LDS round trips in a row with all eight waves of the CU in this same phase
    constexpr int RB = 4;
    const unsigned shb = lds_addr(sh), zb = lds_addr(zero16);
#pragma unroll
    for (int b = 0; b < NRD / RB; ++b) {
      f32x4 v[RB];
      unsigned ad[RB];
#pragma unroll
      for (int k = 0; k < RB; ++k) {
        const int kk = k + b * RB;
        ad[k] = kk < cnt ? shb + kk * (LPI * S::STG_LD * 4) : zb;
      }
      // the four reads AND their wait in ONE statement, early-clobber outputs: as separate statements the compiler was free to copy or
      // spill a destination between a read and the wait -- i.e. to read the register before the LDS data had landed -- and only the allocation of the day kept it from doing so
      static_assert(RB == 4, "the statement below names four reads");
      asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %7\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3])
                   : "v"(ad[0]), "v"(ad[1]), "v"(ad[2]), "v"(ad[3])
                   : "memory");
#pragma unroll
      for (int k = 0; k < RB; ++k) acc += v[k];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v1 = acc[r];
      v1 += __shfl_xor(v1, 16, 64);
      v1 += __shfl_xor(v1, 32, 64);
      acc[r] = v1 * scale;
    }
    if (live && p == 0) *(f32x4 *)(dst + a * S::ENVA + (c >> 2) * S::U + 16 * (2 * h + t) + 4 * (c & 3)) = acc;
  }
}

// ---- pair hand-over through LDS ----
// send: own register images into this wave's buffer xb; after a workgroup barrier the partner's images are read from its buffer.
// The buffers alternate (xb ^= 1 per hand-over): a buffer is rewritten two hand-overs later, i.e. after a barrier that the partner
// passes only with its reads of the older contents complete (s_waitcnt lgkmcnt(0) precedes every s_barrier).
// (A pair-only synchronisation -- per-wave counters in LDS, spin on the partner's -- was measured instead of the workgroup barrier:
// 29.8 vs 29.3 ms on the 41k-atom water box, i.e. the eight waves arrive together anyway; dropped.)
struct Xch {
  float *mine;            // lds.xch[wave]
  const float *theirs;    // lds.xch[partner wave]
};
template <int NT> __device__ __forceinline__ void x_send(const Xch &X, int xb, const f32x4 (&v)[NT]) {
  static_assert(NT <= 2, "two images per buffer");
  const int lane4 = fresh_lane() * 4;
#pragma unroll
  for (int t = 0; t < NT; ++t) *(f32x4 *)(X.mine + (xb * 2 + t) * ROW + lane4) = v[t];
}
__device__ __forceinline__ void x_wait(const Xch &) {
#ifndef ABL_NOSYNC
  __syncthreads();
#endif
}
template <int NT> __device__ __forceinline__ void x_recv(const Xch &X, int xb, f32x4 (&v)[NT]) {
  const int lane4 = fresh_lane() * 4;
#pragma unroll
  for (int t = 0; t < NT; ++t) v[t] = *(const f32x4 *)(X.theirs + (xb * 2 + t) * ROW + lane4);
}
template <int NT> __device__ __forceinline__ void x_swap(const Xch &X, int &xb, const f32x4 (&out)[NT], f32x4 (&in)[NT]) {
  x_send<NT>(X, xb, out);
  x_wait(X);
  x_recv<NT>(X, xb, in);
  xb ^= 1;
}

// Channel mixing with the row hand-over, one (l, m) row at a time, in place on the parked half tensor:
//   V[lm][own] <- ([V[lm][own], V[lm][partner's]] @ M_l)[own]   (forward: the output rows are saved as the next layer's V_in)
//   V[lm][own] <- the same with M_l^T (+ ds on the scalar row)  (backward)
// Row lm + 1 is handed over before row lm's MFMAs are issued, so the partner's images are in LDS when the next barrier falls.
template <int LM, bool FWD, int AR>
__device__ __forceinline__ void mix_rows_p(float (&V)[9][2][4], __amdgpu_buffer_rsrc_t WB, int &wp, LxRing<AR> &ring,
                                           __amdgpu_buffer_rsrc_t SB, int row0, const f32x4 (&ds)[2], const Xch &X, int &xb) {
  if constexpr (LM < 9) {
    f32x4 in[4], o[2];
    in[0] = acc_get4(V[LM][0]); in[1] = acc_get4(V[LM][1]);
    if constexpr (LM == 0) {
      f32x4 me[2] = {in[0], in[1]};
      x_send<2>(X, xb, me);
    }
    x_wait(X);
    {
      f32x4 pr[2];
      x_recv<2>(X, xb, pr);
      in[2] = pr[0]; in[3] = pr[1];
    }
    xb ^= 1;
    if constexpr (LM + 1 < 9) {
      f32x4 nx[2] = {acc_get4(V[LM + 1][0]), acc_get4(V[LM + 1][1])};
      x_send<2>(X, xb, nx);
    }
    const int v16 = fresh_lane() << 4;
    if constexpr (FWD) lx_lin<AR, 4, 2, false>(WB, wp, in, o, v16, ring, EpiSaveN<2>{SB, row0 + LM * 2, v16});
    else lx_lin<AR, 4, 2, false>(WB, wp, in, o, v16, ring, EpiNone{});
#pragma unroll
    for (int t = 0; t < 2; ++t) acc_put4(V[LM][t], (!FWD && LM == 0) ? o[t] + ds[t] : o[t]);
    __builtin_amdgcn_sched_barrier(0);
    mix_rows_p<LM + 1, FWD, AR>(V, WB, wp, ring, SB, row0, ds, X, xb);
  }
}

enum { PP_GEOM = 0, PP_EMB, PP_ENV, PP_TP, PP_LAT, PP_MIX, PP_OUT, PP_BLAT, PP_BMIX, PP_BTP, PP_BENV, PP_BEMB, PP_FIN, PP_N };
#define PHASEP(id) do { if (PROF) { long long _t = clock64(); pacc[id] += _t - tprev; tprev = _t; } } while (0)

// ---------------------------------------------------------------------------- the kernel
template <int NLT, bool PROF, int AR>
__global__ void __launch_bounds__(512, 1) k_fused_lx2(FusedLxArgs A) {
  using S = ShapeP;
  constexpr int NTHREADS = 512, D = S::D, U = S::U, HT = S::HT, EWH = S::EWH, MAXA = S::MAXA, STG_LD = S::STG_LD, ENVA = S::ENVA, NP = S::NP, L = S::L;
  constexpr bool SAVEZ = AR == 3;          // f16x2: raw pre-activation rows of the last hidden layer instead of silu' rows, no u rows (see fused.hip)
  __shared__ LdsP lds;
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 15, g = lane >> 4, wave = tid >> 6;
  const int uwave = __builtin_amdgcn_readfirstlane(wave);
  const int hf = uwave >> 2, q = uwave & 3;            // wave half (channels 32 hf .. 32 hf + 31) and pair index (edge slots 16 q .. 16 q + 15)
  // Lane-derived addresses are RECOMPUTED where they are used (from the hardware lane counter and wave-uniform values) instead of kept in
  // registers across the tile: a register that lives that long is spilled, and its reload -- `s_waitcnt vmcnt(0)`, loads return in order --
  // waits for every saved row requested ahead from HBM.  The centre slot of the lane's edge is parked in LDS for the same reason.
  auto V16 = [&]() { return fresh_lane() << 4; };
  __amdgpu_buffer_rsrc_t SB, WB;
  {
    unsigned long long b = (unsigned long long)(A.scratch + (size_t)blockIdx.x * A.wg_scratch + (size_t)wave * A.wave_scratch);
    unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    SB = __builtin_amdgcn_make_buffer_rsrc((void *)(((unsigned long long)hi << 32) | lo), 0, (int)(A.wave_scratch * 4), 0x00020000);
    WB = __builtin_amdgcn_make_buffer_rsrc((void *)A.wbase, 0, A.wbytes, 0x00020000);
  }
  const float *__restrict__ Wb = A.wbase;
  constexpr int NL = NLT;
  for (int k = tid; k < NL * NP * U; k += NTHREADS) lds.tp[k / (NP * U)][k % (NP * U)] = Wb[A.o_tpl + k];
  const int ntiles = *A.ntiles;
  double acc_part = 0.0;
  long long pacc[PP_N];
  long long tprev = 0;
  if (PROF) {
#pragma unroll
    for (int k = 0; k < PP_N; ++k) pacc[k] = 0;
    tprev = clock64();
  }
  LxRing<AR> ring;
  const int wp0 = hf ? A.o_stream_hi : A.o_stream;
  int wp = wp0;
  lx_prime<AR>(WB, wp, V16(), ring);
  if (tid < MAXA) lds.eacc[tid] = 0.0;
  if (tid < 4) lds.zero16[tid] = 0.f;
  if (hf == 0 && lane < 6) lds.virw[q][lane] = 0.0;
  if (tid < A.T * A.T) lds.rc[tid] = (float)A.rcut[tid];
  if (tid < A.T) { lds.scale[tid] = Wb[A.o_scale + tid]; lds.shift[tid] = Wb[A.o_shift + tid]; }
  if (tid < 2 * NL) lds.res[tid >> 1][tid & 1] = Wb[A.o_res[tid >> 1] + (tid & 1)];

  const int s = q * 16 + j;                    // this lane's edge slot
  const int ca = tid >> 4;                     // centre slot served by this thread in the per-centre output step
  const int choff = 32 * hf;                   // first own channel
  Xch X;
  X.mine = &lds.xch[uwave][0][0];
  X.theirs = &lds.xch[uwave ^ 4][0][0];
  int xb = 0;
  if (tid == 0) lds.chunk[0] = (int)atomicAdd(A.tile_counter, (unsigned)A.tchunk);
  __syncthreads();
  int par = 0, cpar = 0, ck = 0;
  int cbase = __builtin_amdgcn_readfirstlane(lds.chunk[0]);

  for (;;) {
    const int tile = cbase + ck;
    if (tile >= ntiles) break;
    int claimed = 0;
    if (ck == 0 && tid == 0) claimed = (int)atomicAdd(A.tile_counter, (unsigned)A.tchunk);
    const int a0 = A.tile_a0[tile], a1 = A.tile_a0[tile + 1], e0 = A.tile_e0[tile], e1 = A.tile_e0[tile + 1];
    const int na = a1 - a0;
    if (e1 - e0 > S::SLOTS) {
      // a single centre with more edges than the tile has slots: evaluated by the layer-at-a-time kernels afterwards (heavy_generic)
      if (ck == 0 && tid == 0) lds.chunk[cpar ^ 1] = claimed;
      __syncthreads();
      if (++ck == A.tchunk) { ck = 0; cpar ^= 1; cbase = __builtin_amdgcn_readfirstlane(lds.chunk[cpar]); }
      continue;
    }
    par ^= 1;
    int *const aoffp = lds.aoff[par];
    const int e = e0 + s;
    const bool valid = e < e1;
    float rx = 1.f, ry = 0.f, rz = 0.f;
    int aloc = 0, ti = 0, tj = 0, jat = 0, c_i = 0, c_t = 0;
    if (valid) {
      rx = A.rvec[3 * (size_t)e]; ry = A.rvec[3 * (size_t)e + 1]; rz = A.rvec[3 * (size_t)e + 2];
      aloc = A.e_ii[e] - a0;
      jat = A.e_j[e];
      const int tt = A.e_tt[e];
      ti = tt >> 4; tj = tt & 15;
    }
    if (ca < na) { const int2 ci = A.centre[a0 + ca]; c_i = ci.x; c_t = ci.y; }
    if (tid <= na) aoffp[tid] = A.eoff[a0 + tid] - e0;

    // ---------------- geometry ----------------
    const float d = sqrtf(rx * rx + ry * ry + rz * rz);
    const float inv = 1.f / d;
    const float nx = rx * inv, ny = ry * inv, nz = rz * inv;
    const float rc = lds.rc[ti * A.T + tj];
    const float xx = d / rc;
    float fc, dfc_dx;
    cutoff_poly_c(A.p, A.cp, xx, fc, dfc_dx);
    if (!valid) { fc = 0.f; dfc_dx = 0.f; }
    constexpr float C3 = 1.7320508075688772f, C15 = 3.872983346207417f, C5H = 1.118033988749895f;
    float Y[D];
    Y[0] = 1.f;
    Y[1] = C3 * ny; Y[2] = C3 * nz; Y[3] = C3 * nx;
    Y[4] = C15 * nx * ny; Y[5] = C15 * ny * nz; Y[6] = C5H * (2.f * nz * nz - nx * nx - ny * ny);
    Y[7] = C15 * nx * nz; Y[8] = 0.5f * C15 * (nx * nx - ny * ny);
    if (hf == 0 && g == 0) lds.aloc[s] = aloc;           // read back by ENVOFF() after the barrier that follows the embedding
    // this lane's offset into an environment row set: + kk * MAXA*ENVA (layer) + lm * U + 16 t
    auto ENVOFF = [&]() { const int ln = fresh_lane(); return lds.aloc[q * 16 + (ln & 15)] * ENVA + 4 * (ln >> 4) + choff; };
    // this lane's staging row in its half's K-tile
    auto STW = [&]() { const int ln = fresh_lane(); return lds.stage[0] + hf * (S::SLOTS * STG_LD) + (q * 16 + (ln & 15)) * STG_LD + 4 * (ln >> 4); };
    PHASEP(PP_GEOM);

    // ---------------- two-body embedding x0(d; type pair) from the spline table; local tile k = global tile (k + 2 hf) & 3 ----------------
    f32x4 x[4];
    {
      lx_prime<AR>(WB, wp, V16(), ring);        // the ring is not carried through the finish / geometry phases of the tile boundary
      const float tb_invh = (float)A.tb_nk / rc;
      const float sft = d * tb_invh;
      const int kq = min((int)sft, A.tb_nk - 1);
      const float tb_t = sft - (float)kq;
      const int tb_off = (A.o_tbtab + ((ti * A.T + tj) * A.tb_nk + kq) * 256 + 4 * g) * 4;      // byte offset inside the weight buffer: 32 bits per lane
      const float vm = (valid && xx < 1.f) ? 1.f : 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int te = (((t + 2 * hf) & 3) * 4) * 64;                 // wave-uniform: the instruction's scalar offset
        const f32x4 c0 = bload_w(WB, tb_off, te), c1 = bload_w(WB, tb_off, te + 64), c2 = bload_w(WB, tb_off, te + 128), c3 = bload_w(WB, tb_off, te + 192);
        x[t] = (c0 + tb_t * (c1 + tb_t * (c2 + tb_t * c3))) * vm;
        bstore(SB, V16(), (S::R_DX0 + t) * ROW * 4, (c1 + tb_t * (2.f * c2 + (3.f * tb_t) * c3)) * (vm * tb_invh));
      }
    }
    // ---------------- tensor embedding V^0[lm][u] = w0[l][u] Y[lm], own channels ----------------
    float V[D][HT][4];       // own half of the edge tensor, forward; of its gradient, backward: parked in AGPRs (acc_park)
    {
      f32x4 w0[EWH];
      lx_lin<AR, 4, EWH, false>(WB, wp, x, w0, V16(), ring, EpiSave{SB, S::R_W0, V16()});
#pragma unroll
      for (int lm = 0; lm < D; ++lm)
#pragma unroll
        for (int t = 0; t < HT; ++t) acc_put4(V[lm][t], lm == 0 ? w0[t] : w0[l_of_lm(lm) * HT + t] * Y[lm]);
    }
    if (ck == 0 && tid == 0) lds.chunk[cpar ^ 1] = claimed;
    __syncthreads();          // aoff visible; previous tile's LDS users done
    PHASEP(PP_EMB);

    // ---------------- layers, forward ----------------
#pragma unroll
    for (int kk = 0; kk < NL; ++kk) {
      const bool last = (kk == NL - 1);
      const int RL = S::R_LAYER(kk);
      float *const envk = lds.env[kk];
      {
        f32x4 om[EWH];
        lx_lin<AR, 4, EWH, false>(WB, wp, x, om, V16(), ring, EpiSaveFrom<HT>{{SB, RL + S::O_OM, V16()}});        // the backward pass reads the l >= 1 rows only
        // environment sum over the centre's edges: both halves stage one own K-tile, all waves reduce both
#pragma unroll
        for (int t = 0; t < HT; ++t) {
          float *const stw = STW();
#pragma unroll
          for (int lm = 0; lm < D; ++lm) *(f32x4 *)(stw + lm * 16) = lm == 0 ? om[t] : om[l_of_lm(lm) * HT + t] * Y[lm];
          __syncthreads();
          #ifndef ABL_NOREDUCE
          reduce_stage_p(lds.stage[0], lds.zero16, aoffp, envk, na, A.cenv, t, uwave);
#endif
          __builtin_amdgcn_sched_barrier(0);
          __syncthreads();
        }
      }
      PHASEP(PP_ENV);
      // tensor product, in place per own K-tile; the weight-fragment ring is dropped and requested again under the last half pass
      f32x4 sc[HT];            // scalar outputs (l3 = 0) of the tensor product, own channels
      {
        const float *en = envk + ENVOFF();
        const float *tp = lds.tp[kk] + 4 * g + choff;
        if (!last) {
#pragma unroll
          for (int t = 0; t < HT; ++t) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              f32x2 vin[D], out[D];
#pragma unroll
              for (int lm = 0; lm < D; ++lm) vin[lm] = acc_get2(V[lm][t], h);
              if (t == HT - 1 && h == 1) lx_prime<AR>(WB, wp, V16(), ring);
              f32x2 ee[D];
#pragma unroll
              for (int lm = 0; lm < D; ++lm) ee[lm] = *(const f32x2 *)(en + 16 * t + 2 * h + lm * U);
              tp_g<L, 0, false, U>(vin, ee, tp + 16 * t + 2 * h, out);
#pragma unroll
              for (int lm = 0; lm < D; ++lm) acc_put2(V[lm][t], h, out[lm]);
              __builtin_amdgcn_sched_barrier(0);
            }
            sc[t] = acc_get4(V[0][t]);
          }
        } else {
#pragma unroll
          for (int t = 0; t < HT; ++t) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              f32x2 vin[D], out[1];
#pragma unroll
              for (int lm = 0; lm < D; ++lm) vin[lm] = acc_get2(V[lm][t], h);
              if (t == HT - 1 && h == 1) lx_prime<AR>(WB, wp, V16(), ring);
              f32x2 ee[D];
#pragma unroll
              for (int lm = 0; lm < D; ++lm) ee[lm] = *(const f32x2 *)(en + 16 * t + 2 * h + lm * U);
              tp_g<L, 0, true, U>(vin, ee, tp + 16 * t + 2 * h, out);
              set_half(sc[t], h, out[0]);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        }
      }
      PHASEP(PP_TP);
      // latent MLP on [x, scalars], split by output tile: hand-overs of the scalars, z1, z2 and the new x
      {
        f32x4 cat[8], z[2], zin[4], z2[2], xn[2], pr[2];
#pragma unroll
        for (int t = 0; t < 4; ++t) cat[t] = x[t];
        x_swap<2>(X, xb, sc, pr);
        cat[4] = sc[0]; cat[5] = sc[1]; cat[6] = pr[0]; cat[7] = pr[1];
        lx_lin<AR, 8, 2, false>(WB, wp, cat, z, V16(), ring, EpiSiluSaveD{SB, RL + S::O_Z1, V16()});
        x_swap<2>(X, xb, z, pr);
        zin[0] = z[0]; zin[1] = z[1]; zin[2] = pr[0]; zin[3] = pr[1];
        if constexpr (SAVEZ) lx_lin<AR, 4, 2, false>(WB, wp, zin, z2, V16(), ring, EpiSiluSaveZ{SB, RL + S::O_Z2, V16()});
        else lx_lin<AR, 4, 2, false>(WB, wp, zin, z2, V16(), ring, EpiSiluSaveD{SB, RL + S::O_Z2, V16()});
        x_swap<2>(X, xb, z2, pr);
        zin[0] = z2[0]; zin[1] = z2[1]; zin[2] = pr[0]; zin[3] = pr[1];
        const float ra = lds.res[kk][0], rbf = lds.res[kk][1] * fc;
        f32x4 xo[2] = {x[0], x[1]};
        if constexpr (SAVEZ) lx_lin<AR, 4, 2, false>(WB, wp, zin, xn, V16(), ring, EpiResidualNS<2>{xo, ra, rbf});
        else lx_lin<AR, 4, 2, false>(WB, wp, zin, xn, V16(), ring, EpiResidual<2>{{SB, RL + S::O_U, V16()}, xo, ra, rbf});
        x_swap<2>(X, xb, xn, pr);
        x[0] = xn[0]; x[1] = xn[1]; x[2] = pr[0]; x[3] = pr[1];
      }
      PHASEP(PP_LAT);
      // channel mixing, in place per (l, m) row -> V^{kk+1}, saved as the next layer's V_in rows
      if (!last) mix_rows_p<0, true, AR>(V, WB, wp, ring, SB, S::R_LAYER(kk + 1) + S::O_VIN, sc, X, xb);
      PHASEP(PP_MIX);
    }

    // ---------------- read-out (both waves of a pair evaluate it) ----------------
    f32x4 upre[2], zt[2], w0pre[L * HT];
    if constexpr (!SAVEZ) load_rows<2>(SB, S::R_LAYER(NL - 1) + S::O_U, upre, V16());
    load_rows<2>(SB, S::R_LAYER(NL - 1) + S::O_Z2, zt, V16());
    __builtin_amdgcn_sched_barrier(0);
    f32x4 zr[2];
    lx_lin<AR, 4, 2, false>(WB, wp, x, zr, V16(), ring, EpiNone{});
    f32x4 wo1[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) wo1[t] = *(const f32x4 *)(Wb + A.o_out1 + 16 * t + 4 * g);
    float eps = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) eps += silu1(zr[t][r]) * wo1[t][r];
    eps = gsum(eps);
    pin(eps);

    // =========================== backward ===========================
    // f16x2: the backward pass runs scaled by a power of two (fused_h.h) -- per CENTRE TYPE since round 6: the exponent of this centre type's own upstream
    // gradient (every edge of a centre shares it); its inverse waits in the slot's pad floats of the staging tile (never staged, never reduced)
    float bsc = 1.f;
    if constexpr (AR == 3) {
      int bex;
      (void)frexpf(lds.scale[ti] * A.cenv, &bex);
      bsc = ldexpf(1.f, -bex);
      if (g == 0) lds.stage[0][s * STG_LD + D * 16] = ldexpf(1.f, bex);
    }
    const float deps = valid ? lds.scale[ti] * A.cenv * bsc : 0.f;
    f32x4 dx[4];             // dE/dx, all 64 features, own tiles first (replicated in the pair)
    {
      f32x4 dzr[2];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) dzr[t][r] = deps * wo1[t][r] * dsilu1(zr[t][r]);
      lx_lin<AR, 2, 4, false>(WB, wp, dzr, dx, V16(), ring, EpiNone{});
    }
    float dfc_part = 0.f;    // partial sums over the own channels / own latent tiles: they meet in lds.ych at the end of the tile
    float dY[D];
#pragma unroll
    for (int lm = 0; lm < D; ++lm) dY[lm] = 0.f;
    PHASEP(PP_OUT);

#pragma unroll
    for (int kk = NL - 1; kk >= 0; --kk) {
      const bool last = (kk == NL - 1);
      const int RL = S::R_LAYER(kk);
      f32x4 ds[HT];
      f32x4 P[4];            // partial latent gradient: dE/dx^{kk-1} = P(this wave) + P(partner)
      {
        f32x4 du[4], dh[2], din[4], pr[2];
        f32x4 rows1[2];
        load_rows<2>(SB, RL + S::O_Z1, rows1, V16());            // silu'(z1), own tiles: first used one linear from here
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (SAVEZ) {
          // the u rows are not saved (fused.hip: SAVEZ): <u, g> over this wave's hidden tiles falls out of the epilogue of the first backward linear, fed with the unscaled gradient
          const float ra = lds.res[kk][0], rb = lds.res[kk][1];
          float ug = 0.f;
          lx_lin<AR, 4, 2, false>(WB, wp, dx, dh, V16(), ring, EpiMulSiluZ<2>{zt, rb * fc, ug});
          P[0] = ra * dx[0]; P[1] = ra * dx[1];
          P[2] = f32x4{0.f, 0.f, 0.f, 0.f}; P[3] = f32x4{0.f, 0.f, 0.f, 0.f};
          dfc_part += rb * ug;
          pin(dfc_part);
        } else {
          const float ra = lds.res[kk][0], rb = lds.res[kk][1];
          const f32x4 accv = upre[0] * dx[0] + upre[1] * dx[1];       // u is split by output tile: own tiles only
          const float rbfc = rb * fc;
#pragma unroll
          for (int t = 0; t < 4; ++t) du[t] = rbfc * dx[t];
          P[0] = ra * dx[0]; P[1] = ra * dx[1];
          P[2] = f32x4{0.f, 0.f, 0.f, 0.f}; P[3] = f32x4{0.f, 0.f, 0.f, 0.f};
          dfc_part += rb * hsum4(accv);
          pin(dfc_part);
          lx_lin<AR, 4, 2, false>(WB, wp, du, dh, V16(), ring, EpiMulRows<2>{zt});
        }
        x_swap<2>(X, xb, dh, pr);
        din[0] = dh[0]; din[1] = dh[1]; din[2] = pr[0]; din[3] = pr[1];
        lx_lin<AR, 4, 2, false>(WB, wp, din, dh, V16(), ring, EpiMulRows<2>{rows1});
        x_swap<2>(X, xb, dh, pr);
        din[0] = dh[0]; din[1] = dh[1]; din[2] = pr[0]; din[3] = pr[1];
        f32x4 dcat[4];       // own x tiles (2), own scalar tiles (2)
        lx_lin<AR, 4, 4, false>(WB, wp, din, dcat, V16(), ring, EpiNone{});
        P[0] += dcat[0]; P[1] += dcat[1];
        ds[0] = dcat[2]; ds[1] = dcat[3];
      }
      PHASEP(PP_BLAT);
      // mix^T in place per (l, m) row: V holds dE/dV^{kk+1}, becomes dE/dV' (tensor-product output gradient)
      if (!last) mix_rows_p<0, false, AR>(V, WB, wp, ring, SB, 0, ds, X, xb);
      PHASEP(PP_BMIX);
      // tensor-product gradient in place per own K-tile; the per-edge environment gradient goes through the stage
      {
        const float *en = lds.env[kk] + ENVOFF();
        const float *tp = lds.tp[kk] + 4 * g + choff;
        // Saved input rows V^{kk}[.][t] (w0 rows for the first layer) of half pass i = 2 t + h: requested while half pass i - 1 computes its
        // environment gradient, consumed by half pass i's second table pass.  Register budget of a half pass: environment rows 18 + output
        // gradient 18 + dE/dV 18 in the first table pass (dE/dV goes back into V before the second), input rows 18 + output gradient 18 +
        // per-edge environment gradient 18 + the next half pass's rows 18 in the second.
        f32x2 vnext[D];
        f32x4 omall[L * HT];      // own omega rows of l >= 1 for the step after this one: requested before the last reduction
        auto request_vin = [&](int i) {
          const int t = i >> 1, h = i & 1;
          if (kk > 0) {
#pragma unroll
            for (int lm = 0; lm < D; ++lm) vnext[lm] = bload_half(SB, V16() + 8 * h, (RL + S::O_VIN + lm * HT + t) * ROW * 4);
          } else {
#pragma unroll
            for (int l = 0; l <= L; ++l) vnext[l] = bload_half(SB, V16() + 8 * h, (S::R_W0 + l * HT + t) * ROW * 4);
          }
        };
        request_vin(0);
#pragma unroll
        for (int t = 0; t < HT; ++t) {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            f32x2 vin[D], b[D];
#pragma unroll
            for (int lm = 0; lm < D; ++lm) vin[lm] = vnext[lm];
            const float *tph = tp + 16 * t + 2 * h;
            if (!last) {
              f32x2 gg[D];
#pragma unroll
              for (int lm = 0; lm < D; ++lm) gg[lm] = acc_get2(V[lm][t], h);
              {
                f32x2 a[D], ee[D];
#pragma unroll
                for (int lm = 0; lm < D; ++lm) ee[lm] = *(const f32x2 *)(en + 16 * t + 2 * h + lm * U);
                tp_g<L, 1, false, U>(gg, ee, tph, a);
#pragma unroll
                for (int lm = 0; lm < D; ++lm) acc_put2(V[lm][t], h, a[lm]);
              }
              __builtin_amdgcn_sched_barrier(0);
              asm volatile("" ::: "memory");      // the second table pass reads the path weights from LDS again (merged loads = 30 registers held across the first)
              if (2 * t + h + 1 < 2 * HT) request_vin(2 * t + h + 1);
              if (kk == 0) {
#pragma unroll
                for (int lm = D - 1; lm >= 1; --lm) vin[lm] = vin[l_of_lm(lm)] * Y[lm];
              }
              tp_g<L, 2, false, U>(gg, vin, tph, b);
            } else {
              f32x2 gg[1];
              gg[0] = half_of(ds[t], h);
              {
                f32x2 a[D], ee[D];
#pragma unroll
                for (int lm = 0; lm < D; ++lm) ee[lm] = *(const f32x2 *)(en + 16 * t + 2 * h + lm * U);
                tp_g<L, 1, true, U>(gg, ee, tph, a);
#pragma unroll
                for (int lm = 0; lm < D; ++lm) acc_put2(V[lm][t], h, a[lm]);
              }
              __builtin_amdgcn_sched_barrier(0);
              asm volatile("" ::: "memory");      // the second table pass reads the path weights from LDS again (merged loads = 30 registers held across the first)
              if (2 * t + h + 1 < 2 * HT) request_vin(2 * t + h + 1);
              if (kk == 0) {
#pragma unroll
                for (int lm = D - 1; lm >= 1; --lm) vin[lm] = vin[l_of_lm(lm)] * Y[lm];
              }
              tp_g<L, 2, true, U>(gg, vin, tph, b);
            }
            {
              float *const stw = STW();
#pragma unroll
              for (int lm = 0; lm < D; ++lm) *(f32x2 *)(stw + lm * 16 + 2 * h) = b[lm];
            }
            __builtin_amdgcn_sched_barrier(0);
          }
          if (t == HT - 1) load_rows<L * HT>(SB, RL + S::O_OM + HT, omall, V16());
          __syncthreads();
          #ifndef ABL_NOREDUCE
          reduce_stage_p(lds.stage[0], lds.zero16, aoffp, lds.denv, na, A.cenv, t, uwave);
#endif
          __builtin_amdgcn_sched_barrier(0);
          __syncthreads();
        }
        PHASEP(PP_BTP);
        // environment weights backward, own channels: d omega[l][u] = sum_m denv[lm][u] Y[lm];  dY[lm] += sum_u denv[lm][u] omega[l][u]
        f32x4 dom[EWH];
        const float *dn = lds.denv + ENVOFF();
#pragma unroll
        for (int t = 0; t < HT; ++t) {
          f32x4 omr[L + 1];
#pragma unroll
          for (int l = 1; l <= L; ++l) omr[l] = omall[(l - 1) * HT + t];
#pragma unroll
          for (int l = 0; l <= L; ++l) dom[l * HT + t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int lm = 0; lm < D; ++lm) {
            const f32x4 dv = *(const f32x4 *)(dn + lm * U + 16 * t);
            if (lm == 0) dom[t] = dv;
            else {
              dom[l_of_lm(lm) * HT + t] += dv * Y[lm];
              dY[lm] += hsum4(dv * omr[l_of_lm(lm)]);
            }
          }
#pragma unroll
          for (int lm = 1; lm < D; ++lm) pin(dY[lm]);
          __builtin_amdgcn_sched_barrier(0);
        }
        lx_prime<AR>(WB, wp, V16(), ring);
        if (kk > 0) {
          if constexpr (!SAVEZ) load_rows<2>(SB, S::R_LAYER(kk - 1) + S::O_U, upre, V16());
          load_rows<2>(SB, S::R_LAYER(kk - 1) + S::O_Z2, zt, V16());
        } else load_rows<L * HT>(SB, S::R_W0 + HT, w0pre, V16());
        __builtin_amdgcn_sched_barrier(0);
        lx_lin<AR, EWH, 4, true>(WB, wp, dom, P, V16(), ring, EpiNone{});      // split by input tile: partial sums over the own channels
      }
      if (kk > 0) {
        // dE/dx^{kk-1} = P + P(partner): reduce-scatter (each wave completes its own two tiles), then all-gather
        f32x4 snd[2] = {P[2], P[3]}, pr[2];
        x_swap<2>(X, xb, snd, pr);
        dx[0] = P[0] + pr[0]; dx[1] = P[1] + pr[1];
        snd[0] = dx[0]; snd[1] = dx[1];
        x_swap<2>(X, xb, snd, pr);
        dx[2] = pr[0]; dx[3] = pr[1];
      } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) dx[t] = P[t];        // stays partial: everything downstream of it is linear
      }
      PHASEP(PP_BENV);
    }
    // ---------------- embedding backward: V holds dE/dV^0 (own channels); dx = partial latent gradient ----------------
    {
      f32x4 dw0[EWH];
#pragma unroll
      for (int t = 0; t < HT; ++t) {
        f32x4 w0r[L + 1];
#pragma unroll
        for (int l = 1; l <= L; ++l) w0r[l] = w0pre[(l - 1) * HT + t];
#pragma unroll
        for (int l = 0; l <= L; ++l) dw0[l * HT + t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int lm = 0; lm < D; ++lm) {
          const f32x4 dv = acc_get4(V[lm][t]);
          if (lm == 0) dw0[t] = dv;
          else {
            dw0[l_of_lm(lm) * HT + t] += dv * Y[lm];
            dY[lm] += hsum4(dv * w0r[l_of_lm(lm)]);
          }
        }
#pragma unroll
        for (int lm = 1; lm < D; ++lm) pin(dY[lm]);
        __builtin_amdgcn_sched_barrier(0);
      }
      lx_lin<AR, EWH, 4, true>(WB, wp, dw0, dx, V16(), ring, EpiNone{});
      wp = wp0;                                                            // last linear of the tile (wrap-around copy follows it)
    }
    PHASEP(PP_BEMB);
    // ---------------- two-body embedding backward (partial: linear in dx) ----------------
    float dd_part;
    {
      f32x4 rows[4];
      load_rows<4>(SB, S::R_DX0, rows, V16());
      f32x4 accv = dx[0] * rows[0];
#pragma unroll
      for (int t = 1; t < 4; ++t) accv += dx[t] * rows[t];
      dd_part = hsum4(accv);
      pin(dd_part);
    }
    // ---------------- channel sums of the pair meet; geometry backward, outputs (first wave of the pair) ----------------
    {
      float yv[D];
#pragma unroll
      for (int lm = 1; lm < D; ++lm) yv[lm] = gsum(dY[lm]);
      float dfc_tot = gsum(dfc_part), dd_tot = gsum(dd_part);
      float *const yc = &lds.ych[q][j][0];
      if (hf == 1 && g == 0) {
        *(f32x4 *)(yc) = f32x4{yv[1], yv[2], yv[3], yv[4]};
        *(f32x4 *)(yc + 4) = f32x4{yv[5], yv[6], yv[7], yv[8]};
        *(f32x4 *)(yc + 8) = f32x4{dfc_tot, dd_tot, 0.f, 0.f};
      }
      __syncthreads();
      if (hf == 0) {
        const f32x4 y0 = *(const f32x4 *)(yc), y1 = *(const f32x4 *)(yc + 4), y2 = *(const f32x4 *)(yc + 8);
        yv[1] += y0[0]; yv[2] += y0[1]; yv[3] += y0[2]; yv[4] += y0[3];
        yv[5] += y1[0]; yv[6] += y1[1]; yv[7] += y1[2]; yv[8] += y1[3];
        dfc_tot += y2[0]; dd_tot += y2[1];
        if (AR == 3) {
          const float ibs = lds.stage[0][s * STG_LD + D * 16];      // this slot's inverse backward scale (per centre type, written at the start of the backward pass)
          dfc_tot *= ibs; dd_tot *= ibs;
#pragma unroll
          for (int lm = 1; lm < D; ++lm) yv[lm] *= ibs;
        }
        const float dd = dfc_tot * (dfc_dx / rc) + dd_tot;
        // G = sum_lm dE/dY_lm * dY_lm/dn (n treated as a free vector), then projected onto the sphere
        float Gx = C3 * yv[3], Gy = C3 * yv[1], Gz = C3 * yv[2];
        Gx += C15 * (yv[4] * ny + yv[7] * nz + yv[8] * nx) - 2.f * C5H * yv[6] * nx;
        Gy += C15 * (yv[4] * nx + yv[5] * nz - yv[8] * ny) - 2.f * C5H * yv[6] * ny;
        Gz += C15 * (yv[5] * ny + yv[7] * nx) + 4.f * C5H * yv[6] * nz;
        const float gn = Gx * nx + Gy * ny + Gz * nz;
        const float gx = dd * nx + (Gx - gn * nx) * inv;
        const float gy = dd * ny + (Gy - gn * ny) * inv;
        const float gz = dd * nz + (Gz - gn * nz) * inv;
        const float m = valid ? 1.f : 0.f;
        if (AR == 3 && valid && !(fabsf(gx) + fabsf(gy) + fabsf(gz) + fabsf(eps) < 3.0e38f)) *A.err = 1;      // inf / NaN: an operand left float16's range
        float *const st = lds.stage[0] + s * STG_LD;
        if (g == 0) {
          st[0] = m * gx; st[1] = m * gy; st[2] = m * gz; st[3] = m * eps;
          if (valid) {
            atomicAdd(&A.f[3 * (size_t)jat], -(double)gx);
            atomicAdd(&A.f[3 * (size_t)jat + 1], -(double)gy);
            atomicAdd(&A.f[3 * (size_t)jat + 2], -(double)gz);
          }
        }
        float w6[6] = {-m * rx * gx, -m * ry * gy, -m * rz * gz, -m * 0.5f * (rx * gy + ry * gx),
                       -m * 0.5f * (rx * gz + rz * gx), -m * 0.5f * (ry * gz + rz * gy)};
#pragma unroll
        for (int c = 0; c < 6; ++c) {
#pragma unroll
          for (int off = 8; off > 0; off >>= 1) w6[c] += __shfl_xor(w6[c], off, 64);
        }
        if (lane < 6) {
          const float mine = lane == 0 ? w6[0] : lane == 1 ? w6[1] : lane == 2 ? w6[2] : lane == 3 ? w6[3] : lane == 4 ? w6[4] : w6[5];
          lds.virw[q][lane] += (double)mine;
        }
      }
    }
    __syncthreads();
    {
      const int col = tid & 3, part = (tid >> 2) & 3;
      float sum = 0.f;
      if (ca < na)
        for (int sl = aoffp[ca] + part; sl < aoffp[ca + 1]; sl += 4) sum += lds.stage[0][sl * STG_LD + col];
      sum += __shfl_xor(sum, 4, 64);
      sum += __shfl_xor(sum, 8, 64);
      if (ca < na && part == 0) {
        if (col < 3) atomicAdd(&A.f[3 * (size_t)c_i + col], (double)sum);
        else {
          const float ei = lds.scale[c_t] * (sum * A.cenv) + lds.shift[c_t];
          if (A.eatom) A.eatom[c_i] = (double)ei;
          lds.eacc[ca] += (double)ei;
        }
      }
    }
    // no trailing barrier: the next tile's first staging write sits behind the barrier after its embedding linear
    PHASEP(PP_FIN);
    if (++ck == A.tchunk) { ck = 0; cpar ^= 1; cbase = __builtin_amdgcn_readfirstlane(lds.chunk[cpar]); }
  }
  __syncthreads();
  if (tid == 0) {
    for (int a = 0; a < MAXA; ++a) acc_part += lds.eacc[a];
  } else if (tid >= 64 && tid < 70) {
    for (int w = 0; w < 4; ++w) acc_part += lds.virw[w][tid - 64];
  }
  if (PROF && lane == 0) {
#pragma unroll
    for (int k = 0; k < PP_N; ++k) atomicAdd((unsigned long long *)&A.prof[k], (unsigned long long)pacc[k]);
  }
  if (tid == 0) A.partial[7 * (size_t)blockIdx.x] = acc_part;
  if (tid >= 64 && tid < 70) A.partial[7 * (size_t)blockIdx.x + 1 + (tid - 64)] = acc_part;
}

// ---------------------------------------------------------------------------- host side
// dense copy of 16 x 16 tiles of W (row-major, leading dimension ldw): rows rt[], columns ct[]
static std::vector<double> gather_tiles(const double *W, int ldw, const std::vector<int> &rt, const std::vector<int> &ct) {
  const int K = 16 * (int)rt.size(), N = 16 * (int)ct.size();
  std::vector<double> o((size_t)K * N);
  for (int a = 0; a < (int)rt.size(); ++a)
    for (int r = 0; r < 16; ++r)
      for (int b = 0; b < (int)ct.size(); ++b)
        for (int c = 0; c < 16; ++c) o[(size_t)(16 * a + r) * N + 16 * b + c] = W[(size_t)(16 * rt[a] + r) * ldw + 16 * ct[b] + c];
  return o;
}

static void fusedlx2_prepare(Model &m) {
  using S = ShapeP;
  if (!m.fusedlx2_state) m.fusedlx2_state = new FusedLxState();
  FusedLxState &st = *(FusedLxState *)m.fusedlx2_state;
  if (st.ready) return;
  st.L = 2; st.UT = 4;
  const HostModel &h = fused_host_model(m);          // at the kernel's fixed widths (zero-padded when the model is narrower: 33..63 tensor features, ...)
  const int T = h.num_types, NL = h.num_layers, U = S::U, D = S::D, UT = S::UT;
  std::vector<float> w;
  FusedLxArgs &A = st.args;
  std::memset(&A, 0, sizeof(A));
  auto mark = [&]() { while (w.size() % 64) w.push_back(0.f); return (int)w.size(); };
  auto T_ = [&](const std::string &name) -> const double * { return h.get(name).data.data(); };
  st.arith = lx_arith_of(m);
  int h_flags = 0;        // float16 range findings over the weight stream (engine.h: H_RANGE_*)
  // ---- one weight stream per wave half, in the order a tile consumes it (see k_fused_lx2) ----
  for (int hf = 0; hf < 2; ++hf) {
    auto own = [&](int t) { return 2 * hf + t; };
    auto par = [&](int t) { return 2 * (1 - hf) + t; };
    const std::vector<int> xo = {own(0), own(1), par(0), par(1)};          // latent / 64-wide hidden tiles: own first
    const std::vector<int> ownt = {own(0), own(1)};
    std::vector<int> lu;                                                    // own tiles of an (l, u) weight vector: [l][own t]
    for (int l = 0; l <= S::L; ++l)
      for (int t = 0; t < S::HT; ++t) lu.push_back(l * UT + own(t));
    std::vector<int> cat_rows = xo;                                         // latent MLP input [x, scalars]: x own-first, own scalars, partner's
    for (int t = 0; t < 2; ++t) cat_rows.push_back(4 + own(t));
    for (int t = 0; t < 2; ++t) cat_rows.push_back(4 + par(t));
    const std::vector<int> cat_cols = {own(0), own(1), 4 + own(0), 4 + own(1)};
    auto put = [&](const double *W, int ldw, const std::vector<int> &rt, const std::vector<int> &ct) {
      auto sub = gather_tiles(W, ldw, rt, ct);
      if (st.arith == 3) h_flags |= append_frag_h(w, sub.data(), 16 * (int)rt.size(), 16 * (int)ct.size(), 16 * (int)ct.size()) & H_RANGE_OVERFLOW;      // (sub-blocks: the tiny-linear finding is taken per matrix below)
      else append_frag(w, sub.data(), 16 * (int)rt.size(), 16 * (int)ct.size(), 16 * (int)ct.size());
    };
    auto putT = [&](const double *W, int K, int N, const std::vector<int> &rt, const std::vector<int> &ct) {    // tiles of W^T ([N][K])
      auto t = transpose(W, K, N);
      put(t.data(), K, rt, ct);
    };
    const int o_stream = mark();
    if (hf == 0) A.o_stream = o_stream; else A.o_stream_hi = o_stream;
    const size_t stream0 = w.size();
    put(T_("emb.w"), U * 3, xo, lu);
    for (int k = 0; k < NL; ++k) {
      const std::string lk = "l" + std::to_string(k + 1);
      put(T_(lk + ".env"), U * 3, xo, lu);
      put(T_(lk + ".lat.w0"), 64, cat_rows, ownt);
      put(T_(lk + ".lat.w1"), 64, xo, ownt);
      put(T_(lk + ".lat.w2"), 64, xo, ownt);
      if (k < NL - 1) {
        const double *mx = T_(lk + ".mix");            // [L+1][U][U]; block l serves its 2l+1 components
        for (int lm = 0; lm < D; ++lm) put(mx + (size_t)l_of_lm(lm) * U * U, U, xo, ownt);
      }
    }
    put(T_("out.w0"), 32, xo, {0, 1});
    putT(T_("out.w0"), 64, 32, {0, 1}, xo);
    for (int k = NL - 1; k >= 0; --k) {
      const std::string lk = "l" + std::to_string(k + 1);
      putT(T_(lk + ".lat.w2"), 64, 64, xo, ownt);
      putT(T_(lk + ".lat.w1"), 64, 64, xo, ownt);
      putT(T_(lk + ".lat.w0"), 64 + U, 64, xo, cat_cols);
      if (k < NL - 1) {
        const double *mx = T_(lk + ".mix");
        for (int lm = 0; lm < D; ++lm) putT(mx + (size_t)l_of_lm(lm) * U * U, U, U, xo, ownt);
      }
      putT(T_(lk + ".env"), 64, U * 3, lu, xo);
    }
    putT(T_("emb.w"), 64, U * 3, lu, xo);
    for (size_t i = 0; i < (size_t)RING * 256; ++i) w.push_back(w[stream0 + i]);      // wrap-around copy
  }
  // two-body table
  A.tb_nk = 512;
  A.o_tbtab = mark();
  append_two_body_table(w, h, m.rcut_model_host, A.tb_nk);
  // small tables: path weights (last layer: only the scalar paths, the rest zero)
  A.o_tpl = mark();
  for (int k = 0; k < NL; ++k) {
    const HostTensor &tp = h.get("l" + std::to_string(k + 1) + ".tp");
    for (int p = 0; p < S::NP; ++p)
      for (int u = 0; u < U; ++u) w.push_back(p < tp.shape[0] ? (float)(tp.data[(size_t)p * U + u] * ahip_cg_l2_cbase[p]) : 0.f);   // x the path's base |c| (tp_g)
  }
  for (int k = 0; k < NL; ++k) {
    const HostTensor &res = h.get("l" + std::to_string(k + 1) + ".res");
    A.o_res[k] = mark(); w.push_back((float)res.data[0]); w.push_back((float)res.data[1]);
  }
  A.o_out1 = mark(); for (int u = 0; u < 32; ++u) w.push_back((float)h.get("out.w1").data[u]);
  A.o_scale = mark(); for (int t = 0; t < T; ++t) w.push_back((float)h.get("scale").data[t]);
  A.o_shift = mark(); for (int t = 0; t < T; ++t) w.push_back((float)h.get("shift").data[t]);
  mark();
  st.wbuf.reserve(w.size() * sizeof(float));
  copy_h2d(st.wbuf.p, w.data(), w.size() * sizeof(float));       // staged: see engine.h
  A.wbase = st.wbuf.as<float>();
  A.wbytes = (int)(w.size() * sizeof(float));
  A.T = T; A.NL = NL; A.p = h.poly_p;
  A.cenv = (float)(1.0 / std::sqrt(h.avg_num_neighbors));
  {
    const float pf = (float)h.poly_p, ca = 0.5f * (pf + 1) * (pf + 2), cb = pf * (pf + 2), cc = 0.5f * pf * (pf + 1);      // the expressions of cutoff_poly
    A.cp[0] = ca; A.cp[1] = cb; A.cp[2] = cc; A.cp[3] = ca * pf; A.cp[4] = cb * (pf + 1); A.cp[5] = cc * (pf + 2);
  }
  if (st.arith == 3) {
    arith_range_verdict(m, h_flags | model_tiny_linear(h));      // auto: ArithDegraded (run_model falls back to the f32 instance); explicit f16x2: an overflow is an error
    A.err = alarm_word(m);
  }
  A.wave_scratch = (long long)S::R_TOTAL(NL) * ROW;
  hipDeviceProp_t prop;
  AHIP_CHECK(hipGetDeviceProperties(&prop, m.device));
  st.ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  st.scratch.reserve((size_t)st.ncu * S::NW * st.args.wave_scratch * sizeof(float));
  st.args.scratch = st.scratch.as<float>();
  st.partial.reserve((size_t)st.ncu * 7 * sizeof(double));
  st.ntiles.reserve(64);
  st.prof.reserve(64 * sizeof(long long));
  const char *pe = std::getenv("AHIP_FUSED_PROF");
  st.prof_on = pe && pe[0] == '1';
  st.ready = true;
}

bool fusedlx2_run(Model &m, const ComputeArgs &a, std::string *why) {
  using S = ShapeP;
  constexpr int NW = S::NW, SLOTS = S::SLOTS;
  fusedlx2_prepare(m);
  FusedLxState &st = *(FusedLxState *)m.fusedlx2_state;
  m.last_fused_arith = st.arith;
  hipStream_t s = a.stream;
  const int inum = m.inum;
  const int grid = std::max(1, st.ncu - (m.reserve_wgs + 1) / 2);      // see fused.hip: slots left free for the exchange kernels
  lx_pack_tiles(m, st, a, SLOTS, S::MAXA);
  FusedLxArgs A = st.args;
  A.wg_scratch = NW * A.wave_scratch;
  A.eoff = m.b_eoff.as<int>(); A.e_ii = m.b_eii.as<int>(); A.e_j = m.b_ej.as<int>();
  A.e_tt = m.b_ett.as<unsigned char>(); A.rvec = m.b_rvec.as<float>(); A.rcut = m.rcut_model_dev;
  lx_tile_args(m, st, A, SLOTS, S::MAXA);
  A.tchunk = (lx_nedges_estimate(m) / SLOTS > (long long)grid * 256) ? TCHUNK : 1;
  A.f = a.f; A.eatom = a.eatom; A.partial = st.partial.as<double>();
  (void)inum;
  {
    StageTimer tm(m, "model_fused", s);
#define LX2_LAUNCH(NLV, PROFV) do { if (st.arith == 3) hipLaunchKernelGGL((k_fused_lx2<NLV, PROFV, 3>), dim3(grid), dim3(NW * 64), 0, s, A); \
                                    else hipLaunchKernelGGL((k_fused_lx2<NLV, PROFV, 0>), dim3(grid), dim3(NW * 64), 0, s, A); } while (0)
    if (st.prof_on && A.NL == 3) {
      AHIP_CHECK(hipMemsetAsync(st.prof.p, 0, 64 * sizeof(long long), s));
      A.prof = st.prof.as<long long>();
      LX2_LAUNCH(3, true);
    } else if (A.NL == 3) LX2_LAUNCH(3, false);
    else if (A.NL == 2) LX2_LAUNCH(2, false);
    else LX2_LAUNCH(1, false);
#undef LX2_LAUNCH
  }
  AHIP_CHECK(hipGetLastError());
  AHIP_CHECK(prim_sum_columns_f64(m.prim, st.partial.as<double>(), grid, 7, a.engvir, s));
  if (st.prof_on && A.NL == 3) {
    std::vector<long long> hp(PP_N);
    AHIP_CHECK(hipMemcpyAsync(hp.data(), st.prof.p, hp.size() * sizeof(long long), hipMemcpyDeviceToHost, s));
    AHIP_CHECK(hipStreamSynchronize(s));
    static const char *names[PP_N] = {"geom+tb", "embed", "env+reduce", "tp", "latent_mlp", "mix", "readout", "b_latent", "b_mix", "b_tp+reduce", "b_env", "b_embed", "finish"};
    double tot = 0;
    for (int k = 0; k < PP_N; ++k) tot += (double)hp[k];
    std::fprintf(stderr, "[ahip fused_lx2 prof] wave-cycles by phase (sum over %d waves):", grid * NW);
    for (int k = 0; k < PP_N; ++k) std::fprintf(stderr, " %s=%.1f%%", names[k], 100.0 * hp[k] / tot);
    std::fprintf(stderr, " | total=%.3g cycles\n", tot);
  }
  (void)why;
  return true;
}

void fusedlx2_free(Model &m) {
  if (!m.fusedlx2_state) return;
  FusedLxState *st = (FusedLxState *)m.fusedlx2_state;
  for (DevBuf *b : {&st->wbuf, &st->scratch, &st->seg_count, &st->seg_base, &st->tile_a0, &st->tile_e0, &st->centre, &st->ntiles, &st->partial, &st->prof}) b->release();
  delete st;
  m.fusedlx2_state = nullptr;
}

}  // namespace ahip
