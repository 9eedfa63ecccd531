// Fused MFMA path for the wider Allegro shapes: l_max = 1 or 2, 32 or 64 tensor features (BASELINE config 5's model L:
// l_max = 2, U = 64, 3 layers; the reference test YAML's shape: l_max = 2, U = 32, 3 layers,
// /root/reference/tests/test_data/test_repro_allegro.yaml:89-99).  Same mapping as fused.hip (one launch = forward + analytic
// backward, every per-edge vector in the v_mfma_f32_16x16x4_f32 C/D layout, weights as one A-fragment stream in consumption
// order, per-centre reductions through LDS, saved rows in a per-wave scratch) with what the larger state forces:
//
//  * The edge tensor V[lm][u] is D*U = 576 floats per edge for model L: 144 registers per lane.  It stays in REGISTERS for the
//    whole tile -- streaming it through memory would cost ~50 KB per edge per step -- so a wave owns the whole 512-entry
//    register file of its SIMD: one 4-wave workgroup (64 edge slots) per CU, __launch_bounds__(256, 1).
//  * The tensor product is channel-wise, so it runs one 16-feature K-tile t at a time IN PLACE on V[.][t] (9 + 9 live rows), with
//    the Clebsch-Gordan table unrolled at compile time from cg_tables.h; the channel mixing V[lm] <- V[lm] @ M_l is done in place
//    per (l, m) row; backward the same way (mix^T in place, then the tensor-product gradient in place per K-tile).
//  * The per-centre environment sum goes through a double-buffered LDS stage holding ONE K-tile ([slots][D x 16]); one barrier
//    per K-tile.
//  * Two-body embedding from the per-type-pair spline table (fused_common.h), always.
// Reference graph: the TorchScript model executed at /root/reference/pair_nequip_allegro.cpp:409-430; the oracle is
// oracle/allegro_torch.py (autograd), so the hand-derived backward below is checked against an independent derivation.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/allegro_hip.h"
#ifndef AHIP_ROW_AUX
#define AHIP_ROW_AUX 2          // saved rows: non-temporal (fused_common.h)
#endif
#include "engine.h"
#include "fused_lx_common.h"
#include "prims.h"

namespace ahip {

// Shapes of one instantiation
template <int L, int UT, int NW> struct ShapeX {
  static constexpr int D = (L + 1) * (L + 1), NLP = L + 1, U = 16 * UT, EW = NLP * UT;   // EW: 16-feature tiles of an (l, u) weight vector
  static constexpr int SLOTS = 16 * NW;
  static constexpr int MAXA = 4;                        // centre atoms per tile (LDS budget of the environment rows)
  static constexpr int STG_LD = D * 16 + 4;             // one K-tile of a slot: [lm][16] + pad (16-byte rows, conflict-free b128 writes)
  static constexpr int ENVA = D * U + 4;                // environment row of one centre: [lm][u] + pad
  static constexpr int NP = CgX<L>::NP;
  // scratch rows (per wave, 1 KiB each): d x0/dd 4 | w0 EW | per layer: omega EW, silu'(z1) 4, silu'(z2) 4, u 4, V_in D*UT
  static constexpr int R_DX0 = 0, R_W0 = 4, LSZ = EW + 12 + D * UT;
  __host__ __device__ static constexpr int R_LAYER(int kk) { return 4 + EW + kk * LSZ; }
  __host__ __device__ static constexpr int R_TOTAL(int NL) { return 4 + EW + NL * LSZ; }
  static constexpr int O_OM = 0, O_Z1 = EW, O_Z2 = EW + 4, O_U = EW + 8, O_VIN = EW + 12;
};

template <int L, int UT, int NW> struct __attribute__((aligned(16))) LdsX {
  using S = ShapeX<L, UT, NW>;
  float stage[2][S::SLOTS * S::STG_LD];
  float env[LX_MAXNL][S::MAXA * S::ENVA];
  float denv[S::MAXA * S::ENVA];
  float tp[LX_MAXNL][S::NP * S::U];          // tensor-product path weights [layer][path][u]
  double eacc[S::MAXA];
  double virw[NW][6];
  int aoff[2][S::MAXA + 2];
  float rc[256];                             // model cutoff table [T*T], T <= 16 (the edge build packs a type in 4 bits)
  float scale[16], shift[16];
  float res[LX_MAXNL][2];
  int chunk[2];
  // Round 6: rows of the LAST layer that never travel to memory.  One workgroup per CU leaves ~58 KB of LDS unused: per wave NLROW register-image rows hold the last
  // layer's omega rows (l >= 1) and the first rows of its input tensor; its 8 latent-MLP rows go to the wave's own slots of stage[0], idle between that layer's
  // environment sum and its backward tensor product (as in k_fused: EpiSiluSaveDL / EpiSiluSaveZL).  22 of the 118 rows a wave-tile of the reference YAML's shape saves.
  static constexpr int NLROW = 14;
  float rowsl[NW][NLROW * ROW];
};
static_assert(sizeof(LdsX<2, 2, 4>) <= 160 * 1024, "LDS budget of one CU");


// Per-centre sum of one staged K-tile: env[a][lm][16 t + f] = scale * sum_{slots of a} stage[slot][lm][f].
// Work item = (centre, 4-feature column); its LPI adjacent lanes take every LPI-th slot with 16-byte LDS reads (all of a lane's reads
// are in flight together) and combine with log2(LPI) cross-lane adds -- a fixed order, so the sums are reproducible.
template <int L, int UT, int NW>
__device__ __forceinline__ void reduce_stage_x(const float *stg, const int *aoff, float *dst, int na, float scale, int t, int uwave) {
  using S = ShapeX<L, UT, NW>;
  constexpr int LPI = 4;                    // lanes per work item: 36 columns x 4 lanes = one round for a tile that holds one centre
  constexpr int NC = S::D * 4, PER_ROUND = NW * 64 / LPI, NRD = S::SLOTS / LPI;
  // lane -> (item, part): part = lane / 16, item = 16 * wave + lane % 16.  The 16 lanes of one 16-byte LDS read phase then hold 16
  // consecutive 4-feature columns of the same slot: distinct banks.  (Adjacent lanes = the parts of one item hit rows 148 floats
  // apart with 16-byte reads: SQ_LDS_BANK_CONFLICT was 76 % of the LDS-active cycles.)
  static_assert(LPI == 4, "four 16-lane groups");
  const int lane = fresh_lane();
  const int p = lane >> 4;
  for (int it = uwave * 16 + (lane & 15); it < ((na * NC + PER_ROUND - 1) / PER_ROUND) * PER_ROUND; it += PER_ROUND) {   // whole waves iterate together
    const bool live = it < na * NC;
    const int a = live ? it / NC : 0, c = live ? it - a * NC : 0;
    const int s0 = aoff[a] + p, s1 = live ? aoff[a + 1] : 0;
    // all reads are issued before the first is used: addresses past the centre's last slot are clamped to slot 0 of the stage
    // (always mapped) and their values discarded -- a conditional read per slot costs one LDS round trip each
    f32x4 v[NRD];
#pragma unroll
    for (int k = 0; k < NRD; ++k) {
      const int sl = s0 + LPI * k;
      v[k] = *(const f32x4 *)(stg + (sl < s1 ? sl : 0) * S::STG_LD + 4 * c);
    }
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < NRD; ++k)
      if ((s0 + LPI * k) < s1) acc += v[k];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v1 = acc[r];
      v1 += __shfl_xor(v1, 16, 64);
      v1 += __shfl_xor(v1, 32, 64);
      acc[r] = v1 * scale;
    }
    if (live && p == 0) *(f32x4 *)(dst + a * S::ENVA + (c >> 2) * S::U + 16 * t + 4 * (c & 3)) = acc;
  }
}

// Channel mixing, one (l, m) row at a time, in place on the parked edge tensor: V[lm] <- V[lm] @ M_l (forward, output rows
// saved as the next layer's V_in) or V[lm] <- V[lm] @ M_l^T (+ ds on the scalar row: backward).  With 32 tensor features a row
// is 2 x 2 tiles = 4 weight fragments = half a ring: rows alternate the ring phase, and the last (ninth) row is zero-padded on
// the host to 8 fragments so that the stream stays ring-aligned.
template <int LM, int D, int UT, bool FWD, int AR, bool LSPLIT = false>
__device__ __forceinline__ void mix_rows(float (&V)[D][UT][4], __amdgpu_buffer_rsrc_t WB, int &wp, int v16, LxRing<AR> &ring,
                                         __amdgpu_buffer_rsrc_t SB, int row0, const f32x4 (&ds)[UT], float *lbase = nullptr, int lrow0 = 0, int nlds = 0) {
  if constexpr (LM < D) {
    constexpr bool PADDED = (UT == 2) && (LM == D - 1) && (D % 2 == 1);
    constexpr int NTO = PADDED ? 4 : UT;
    constexpr int RP = (UT == 2 && !PADDED) ? 4 * (LM & 1) : 0;
    f32x4 vi[UT], o[NTO];
#pragma unroll
    for (int t = 0; t < UT; ++t) vi[t] = acc_get4(V[LM][t]);
    v16 = fresh_lane() << 4;          // formed here (as fused_lx2.hip does): the value computed at kernel entry was spilled and reloaded in front of every row store
    if constexpr (FWD) {
      // (lbase: the output rows are the LAST layer's input tensor -- its first nlds rows stay in LDS)
      if constexpr (LSPLIT) lx_lin<AR, UT, NTO, false, RP>(WB, wp, vi, o, v16, ring, EpiSaveNSplit<UT>{SB, row0, v16, lbase, LM * UT, lrow0, nlds, fresh_lane()});
      else lx_lin<AR, UT, NTO, false, RP>(WB, wp, vi, o, v16, ring, EpiSaveN<UT>{SB, row0 + LM * UT, v16});
    } else lx_lin<AR, UT, NTO, false, RP>(WB, wp, vi, o, v16, ring, EpiNone{});
#pragma unroll
    for (int t = 0; t < UT; ++t) acc_put4(V[LM][t], (!FWD && LM == 0) ? o[t] + ds[t] : o[t]);
    __builtin_amdgcn_sched_barrier(0);
    mix_rows<LM + 1, D, UT, FWD, AR, LSPLIT>(V, WB, wp, v16, ring, SB, row0, ds, lbase, lrow0, nlds);
  }
}

enum { PX_GEOM = 0, PX_EMB, PX_ENV, PX_TP, PX_LAT, PX_MIX, PX_OUT, PX_BLAT, PX_BMIX, PX_BTP, PX_BENV, PX_BEMB, PX_FIN, PX_N };
#define PHASEX(id) do { if (PROF) { long long _t = clock64(); pacc[id] += _t - tprev; tprev = _t; } } while (0)

// ---------------------------------------------------------------------------- the kernel
// NLT = number of layers: the layer loops are unrolled so that `last layer` / `first layer` are compile-time facts -- with
// run-time branches inside them the register allocator shuffles dozens of spill slots at every join (load, wait, store).
template <int L, int UT, int NW, int NLT, bool PROF, int AR>
__global__ void __launch_bounds__(NW * 64, 1) k_fused_lx(FusedLxArgs A) {
  using S = ShapeX<L, UT, NW>;
  constexpr int NTHREADS = NW * 64, D = S::D, U = S::U, EW = S::EW, MAXA = S::MAXA, STG_LD = S::STG_LD, ENVA = S::ENVA, NP = S::NP;
  constexpr bool SAVEZ = AR == 3;          // f16x2: raw pre-activation rows of the last hidden layer instead of silu' rows, no u rows (see fused.hip)
  // last-layer rows in LDS (LdsX::rowsl, stage[0] images): LDS rows [0, NOML) = omega l >= 1, [NOML, NLROW) = the first NVL rows of the input tensor
  constexpr int NLROW = LdsX<L, UT, NW>::NLROW, NOML = L * UT, NVL = (NLROW - NOML) < D * UT ? (NLROW - NOML) : D * UT;
  static_assert(NOML <= NLROW, "omega rows of the last layer fit the LDS rows");
  __shared__ LdsX<L, UT, NW> lds;
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 15, g = lane >> 4, wave = tid >> 6;
  const int uwave = __builtin_amdgcn_readfirstlane(wave);
  const int v16 = lane * 16;
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
  long long pacc[PX_N];
  long long tprev = 0;
  if (PROF) {
#pragma unroll
    for (int k = 0; k < PX_N; ++k) pacc[k] = 0;
    tprev = clock64();
  }
  LxRing<AR> ring;
  int wp = A.o_stream;
  lx_prime<AR>(WB, wp, v16, ring);
  if (tid < MAXA) lds.eacc[tid] = 0.0;
  if (lane < 6) lds.virw[wave][lane] = 0.0;
  if (tid < A.T * A.T) lds.rc[tid] = (float)A.rcut[tid];
  if (tid < A.T) { lds.scale[tid] = Wb[A.o_scale + tid]; lds.shift[tid] = Wb[A.o_shift + tid]; }
  if (tid < 2 * NL) lds.res[tid >> 1][tid & 1] = Wb[A.o_res[tid >> 1] + (tid & 1)];

  const int s = wave * 16 + j;                 // this lane's edge slot
  const int ca = tid >> 4;                     // centre slot served by this thread in the per-centre output step
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
      // a single centre with more edges than the tile has slots (the packing gives such a centre a tile of its own): it is
      // evaluated by the layer-at-a-time kernels afterwards (heavy_generic, allegro_hip.hip)
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
    // real spherical harmonics, component normalisation, m = -l..l (pair_allegro_amd/cg.py)
    constexpr float C3 = 1.7320508075688772f, C15 = 3.872983346207417f, C5H = 1.118033988749895f;
    float Y[D];
    Y[0] = 1.f;
    Y[1] = C3 * ny; Y[2] = C3 * nz; Y[3] = C3 * nx;
    if constexpr (L >= 2) {
      Y[4] = C15 * nx * ny; Y[5] = C15 * ny * nz; Y[6] = C5H * (2.f * nz * nz - nx * nx - ny * ny);
      Y[7] = C15 * nx * nz; Y[8] = 0.5f * C15 * (nx * nx - ny * ny);
    }
    const int envoff = aloc * ENVA + 4 * g;      // + kk * MAXA*ENVA (layer) + lm * U + 16 t
    // this lane's staging row in buffer b, formed where it is used from the hardware lane counter (round 6): as a value computed at the top of the tile it
    // lived in scratch once the last layer's rows moved into the stage, and its reload in front of every staging write drained the in-order load queue
    auto STQ = [&](int b) { const int ln = fresh_lane(); return lds.stage[b] + (uwave * 16 + (ln & 15)) * STG_LD + 4 * (ln >> 4); };
    PHASEX(PX_GEOM);

    // ---------------- two-body embedding x0(d; type pair) from the spline table ----------------
    f32x4 x[4];
    {
      lx_prime<AR>(WB, wp, v16, ring);        // RING_DROP: not carried through the finish / geometry phases of the tile boundary either
      const float tb_invh = (float)A.tb_nk / rc;
      const float sft = d * tb_invh;
      const int kq = min((int)sft, A.tb_nk - 1);
      const float tb_t = sft - (float)kq;
      const int tb_off = (A.o_tbtab + ((ti * A.T + tj) * A.tb_nk + kq) * 256 + 4 * g) * 4;      // byte offset inside the weight buffer: 32 bits per lane, the 16 gathers differ in the immediate offset
      const float vm = (valid && xx < 1.f) ? 1.f : 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const f32x4 c0 = bload_w(WB, tb_off + (t * 4 + 0) * 64, 0), c1 = bload_w(WB, tb_off + (t * 4 + 1) * 64, 0);
        const f32x4 c2 = bload_w(WB, tb_off + (t * 4 + 2) * 64, 0), c3 = bload_w(WB, tb_off + (t * 4 + 3) * 64, 0);
        x[t] = (c0 + tb_t * (c1 + tb_t * (c2 + tb_t * c3))) * vm;
        bstore(SB, v16, (S::R_DX0 + t) * ROW * 4, (c1 + tb_t * (2.f * c2 + (3.f * tb_t) * c3)) * (vm * tb_invh));
      }
    }
    // ---------------- tensor embedding V^0[lm][u] = w0[l][u] Y[lm] ----------------
    float V[D][UT][4];       // the edge tensor, forward; its gradient, backward: parked in AGPRs (acc_park)
    {
      f32x4 w0[EW];
      lx_lin<AR, 4, EW, false>(WB, wp, x, w0, v16, ring, EpiSave{SB, S::R_W0, v16});
#pragma unroll
      for (int lm = 0; lm < D; ++lm)
#pragma unroll
        for (int t = 0; t < UT; ++t) acc_put4(V[lm][t], lm == 0 ? w0[t] : w0[l_of_lm(lm) * UT + t] * Y[lm]);
    }
    if (ck == 0 && tid == 0) lds.chunk[cpar ^ 1] = claimed;
    __syncthreads();          // aoff visible; previous tile's LDS users done
    PHASEX(PX_EMB);

    // ---------------- layers, forward ----------------
#pragma unroll
    for (int kk = 0; kk < NL; ++kk) {
      const bool last = (kk == NL - 1);
      const int RL = S::R_LAYER(kk);
      float *const envk = lds.env[kk];
      {
        f32x4 om[EW];
        // the backward pass reads the l >= 1 rows only; the last layer's stay in LDS
        if (last) lx_lin<AR, 4, EW, false>(WB, wp, x, om, v16, ring, EpiSaveFromL<UT>{lds.rowsl[uwave], fresh_lane()});
        else lx_lin<AR, 4, EW, false>(WB, wp, x, om, v16, ring, EpiSaveFrom<UT>{{SB, RL + S::O_OM, v16}});
        // environment sum over the centre's edges, one K-tile at a time through the double-buffered stage
#ifndef ABL_NO_ENVSTAGE
#pragma unroll
        for (int t = 0; t < UT; ++t) {
          float *const sp = STQ(t & 1);
#pragma unroll
          for (int lm = 0; lm < D; ++lm) *(f32x4 *)(sp + lm * 16) = lm == 0 ? om[t] : om[l_of_lm(lm) * UT + t] * Y[lm];
          __syncthreads();
          reduce_stage_x<L, UT, NW>(lds.stage[t & 1], aoffp, envk, na, A.cenv, t, uwave);
          __builtin_amdgcn_sched_barrier(0);
        }
#endif
        __syncthreads();
      }
      PHASEX(PX_ENV);
      // tensor product, in place per K-tile.  RING_DROP: the weight-fragment ring is NOT carried through the tensor-product phases (no
      // linear runs in them): the fragments prefetched by the previous linear's tail are dropped and requested again under the last
      // half pass, which frees 32 registers where the pressure peaks (one extra L2 round trip per phase, hidden by that half pass)
      f32x4 sc[UT];            // scalar outputs (l3 = 0) of the tensor product: the latent MLP's second input
      {
        const float *en = envk + envoff;
        const float *tp = lds.tp[kk] + 4 * (fresh_lane() >> 4);      // (lane-derived offsets are formed where they are used: as tile-long values they end up in scratch)
#ifndef ABL_NO_FTP
        if (!last) {
#pragma unroll
          for (int t = 0; t < UT; ++t) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              f32x2 vin[D], out[D];
#pragma unroll
              for (int lm = 0; lm < D; ++lm) vin[lm] = acc_get2(V[lm][t], h);
              if (t == UT - 1 && h == 1) lx_prime<AR>(WB, wp, v16, ring);      // see RING_DROP below
              tp_fwd_x<L, false, U>(vin, en + 16 * t + 2 * h, tp + 16 * t + 2 * h, out);
#pragma unroll
              for (int lm = 0; lm < D; ++lm) acc_put2(V[lm][t], h, out[lm]);
              __builtin_amdgcn_sched_barrier(0);
            }
            sc[t] = acc_get4(V[0][t]);
          }
        } else {
#pragma unroll
          for (int t = 0; t < UT; ++t) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              f32x2 vin[D], out[1];
#pragma unroll
              for (int lm = 0; lm < D; ++lm) vin[lm] = acc_get2(V[lm][t], h);
              if (t == UT - 1 && h == 1) lx_prime<AR>(WB, wp, v16, ring);
              tp_fwd_x<L, true, U>(vin, en + 16 * t + 2 * h, tp + 16 * t + 2 * h, out);
              set_half(sc[t], h, out[0]);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        }
#else
#pragma unroll
        for (int t = 0; t < UT; ++t) sc[t] = acc_get4(V[0][t]) * *(const f32x4 *)(en + 16 * t) * *(const f32x4 *)(tp + 16 * t);
#endif
      }
      PHASEX(PX_TP);
      // latent MLP on [x, scalars]
#ifndef ABL_NO_LAT
      {
        f32x4 cat[4 + UT], z[4], z2[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) cat[t] = x[t];
#pragma unroll
        for (int t = 0; t < UT; ++t) cat[4 + t] = sc[t];
        if (last) {          // the last layer's rows: images 0..3 / 4..7 of the wave's own slots of stage[0] (idle until this layer's backward tensor product)
          lx_lin<AR, 4 + UT, 4, false>(WB, wp, cat, z, v16, ring, EpiSiluSaveDL{STQ(0), 0});
          if constexpr (SAVEZ) lx_lin<AR, 4, 4, false>(WB, wp, z, z2, v16, ring, EpiSiluSaveZL{STQ(0), 4});
          else lx_lin<AR, 4, 4, false>(WB, wp, z, z2, v16, ring, EpiSiluSaveDL{STQ(0), 4});
        } else {
          lx_lin<AR, 4 + UT, 4, false>(WB, wp, cat, z, v16, ring, EpiSiluSaveD{SB, RL + S::O_Z1, v16});
          if constexpr (SAVEZ) lx_lin<AR, 4, 4, false>(WB, wp, z, z2, v16, ring, EpiSiluSaveZ{SB, RL + S::O_Z2, v16});
          else lx_lin<AR, 4, 4, false>(WB, wp, z, z2, v16, ring, EpiSiluSaveD{SB, RL + S::O_Z2, v16});
        }
        const float ra = lds.res[kk][0], rbf = lds.res[kk][1] * fc;
        f32x4 xn[4];
        if constexpr (SAVEZ) lx_lin<AR, 4, 4, false>(WB, wp, z2, xn, v16, ring, EpiResidualNS<4>{x, ra, rbf});
        else lx_lin<AR, 4, 4, false>(WB, wp, z2, xn, v16, ring, EpiResidual<4>{{SB, RL + S::O_U, v16}, x, ra, rbf});
#pragma unroll
        for (int t = 0; t < 4; ++t) x[t] = xn[t];
      }
#else
      x[0] += sc[0]; x[1] += sc[1]; x[2] += sc[2]; x[3] += sc[3];
#endif
      PHASEX(PX_LAT);
      // channel mixing, in place per (l, m) row -> V^{kk+1}, saved as the next layer's V_in rows
#ifndef ABL_NO_MIX
      if (!last) {
        if (kk + 1 == NL - 1) mix_rows<0, D, UT, true, AR, true>(V, WB, wp, v16, ring, SB, S::R_LAYER(kk + 1) + S::O_VIN, sc, lds.rowsl[uwave], NOML, NVL);
        else mix_rows<0, D, UT, true, AR>(V, WB, wp, v16, ring, SB, S::R_LAYER(kk + 1) + S::O_VIN, sc);
      }
#endif
      PHASEX(PX_MIX);
    }

    // ---------------- read-out ----------------
    // saved rows are requested well ahead of their first use all through the backward pass (their round trip is an L2 miss:
    // 1-2 us, and with one wave per SIMD nothing else covers it): u and silu'(z2) of the last layer under the read-out MFMAs
    f32x4 upre[4], zt[4], w0pre[L * UT];
    if constexpr (!SAVEZ) load_rows<4>(SB, S::R_LAYER(NL - 1) + S::O_U, upre, v16);
    __builtin_amdgcn_sched_barrier(0);
    f32x4 zr[2];
    lx_lin<AR, 4, 2, false>(WB, wp, x, zr, v16, ring, EpiNone{});
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
    f32x4 dx[4];
    {
      f32x4 dzr[2];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) dzr[t][r] = deps * wo1[t][r] * dsilu1(zr[t][r]);
      lx_lin<AR, 2, 4, false>(WB, wp, dzr, dx, v16, ring, EpiNone{});
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) zt[t] = stg_load(STQ(0), 4 + t);            // the last layer's second hidden layer rows, from the staging tile
    float dfc_part = 0.f;
    float dY[D];
#pragma unroll
    for (int lm = 0; lm < D; ++lm) dY[lm] = 0.f;
    PHASEX(PX_OUT);

#pragma unroll
    for (int kk = NL - 1; kk >= 0; --kk) {
      const bool last = (kk == NL - 1);
      const int RL = S::R_LAYER(kk);
      f32x4 ds[UT];
      {
        f32x4 du[4], dh[4];
        f32x4 rows1[4];
        if (last) {
#pragma unroll
          for (int t = 0; t < 4; ++t) rows1[t] = stg_load(STQ(0), t);
        } else load_rows<4>(SB, RL + S::O_Z1, rows1, v16);     // silu'(z1): first used one linear from here
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (SAVEZ) {
          // the u rows are not saved (fused.hip: SAVEZ): <u, g> falls out of the epilogue of the first backward linear, fed with the unscaled gradient
          const float ra = lds.res[kk][0], rb = lds.res[kk][1];
          float ug = 0.f;
          lx_lin<AR, 4, 4, false>(WB, wp, dx, dh, v16, ring, EpiMulSiluZ<4>{zt, rb * fc, ug});
#pragma unroll
          for (int t = 0; t < 4; ++t) dx[t] = ra * dx[t];
          dfc_part += rb * ug;
          pin(dfc_part);
        } else {
          const float ra = lds.res[kk][0], rb = lds.res[kk][1];
          f32x4 accv = upre[0] * dx[0];
#pragma unroll
          for (int t = 1; t < 4; ++t) accv += upre[t] * dx[t];
          const float rbfc = rb * fc;
#pragma unroll
          for (int t = 0; t < 4; ++t) { du[t] = rbfc * dx[t]; dx[t] = ra * dx[t]; }
          dfc_part += rb * hsum4(accv);
          pin(dfc_part);
          lx_lin<AR, 4, 4, false>(WB, wp, du, dh, v16, ring, EpiMulRows<4>{zt});
        }
        lx_lin<AR, 4, 4, false>(WB, wp, dh, du, v16, ring, EpiMulRows<4>{rows1});
        f32x4 dcat[4 + UT];
        lx_lin<AR, 4, 4 + UT, false>(WB, wp, du, dcat, v16, ring, EpiNone{});
#pragma unroll
        for (int t = 0; t < 4; ++t) dx[t] += dcat[t];
#pragma unroll
        for (int t = 0; t < UT; ++t) ds[t] = dcat[4 + t];
      }
      PHASEX(PX_BLAT);
      // mix^T in place per (l, m) row: V holds dE/dV^{kk+1}, becomes dE/dV' (tensor-product output gradient)
      if (!last) mix_rows<0, D, UT, false, AR>(V, WB, wp, v16, ring, SB, 0, ds);
      PHASEX(PX_BMIX);
      // tensor-product gradient in place per K-tile; the per-edge environment gradient goes through the stage
      {
        const float *en = lds.env[kk] + envoff;
        const float *tp = lds.tp[kk] + 4 * (fresh_lane() >> 4);      // (lane-derived offsets are formed where they are used: as tile-long values they end up in scratch)
        // the saved input rows V^{kk}[.][t] (w0 rows for the first layer) of half pass (t, h) are requested one half pass
        // ahead: their round trip (L2 miss: 1-2 us) runs under the previous half pass
        // saved input rows of half pass i = 2 t + h live in vpre2[i & 1] and are requested TWO half passes ahead (an HBM round trip
        // under load is ~2 us, one half pass ~1 us)
        f32x2 vpre2[2][D];
        f32x4 omall[L * UT];      // omega rows of l >= 1 (all K-tiles) for the step after this one: requested before the last reduction
        auto request_vin = [&](int i) {
          const int t = i >> 1, h = i & 1;
          if (kk > 0) {
#pragma unroll
            for (int lm = 0; lm < D; ++lm) {
              if (last && lm * UT + t < NVL) vpre2[i & 1][lm] = *(const f32x2 *)(lds.rowsl[uwave] + (NOML + lm * UT + t) * ROW + fresh_lane() * 4 + 2 * h);
              else vpre2[i & 1][lm] = bload_half(SB, v16 + 8 * h, (RL + S::O_VIN + lm * UT + t) * ROW * 4);
            }
          } else {
#pragma unroll
            for (int l = 0; l <= L; ++l) vpre2[i & 1][l] = bload_half(SB, v16 + 8 * h, (S::R_W0 + l * UT + t) * ROW * 4);
          }
        };
        request_vin(0);
        request_vin(1);
#pragma unroll
        for (int t = 0; t < UT; ++t) {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            f32x2 vin[D], a[D], b[D], ee[D];
#pragma unroll
            for (int lm = 0; lm < D; ++lm) vin[lm] = vpre2[h][lm];
#pragma unroll
            for (int lm = 0; lm < D; ++lm) ee[lm] = *(const f32x2 *)(en + 16 * t + 2 * h + lm * U);
            f32x2 pwh[NP];          // path weights of this half pass: one batch of LDS reads, one wait
#pragma unroll
            for (int pth = 0; pth < NP; ++pth) pwh[pth] = *(const f32x2 *)(tp + 16 * t + 2 * h + pth * U);
            __builtin_amdgcn_sched_barrier(0);
            // next half pass's rows: requested AFTER this half pass's LDS reads -- a spilled LDS address reloaded between the
            // request and those reads would wait on vmcnt(0), i.e. on the rows just requested (loads return in order)
            // the stage pointer is needed after the tensor-product arithmetic: if it sits in a scratch slot, reload it NOW -- after
            // the request below a reload waits (vmcnt(0), in-order return) for the rows coming from HBM
            float *const sp = STQ(t & 1);
            if (2 * t + h + 2 < 2 * UT) request_vin(2 * t + h + 2);
            __builtin_amdgcn_sched_barrier(0);
            if (!last) {
              f32x2 gg[D];
#pragma unroll
              for (int lm = 0; lm < D; ++lm) gg[lm] = acc_get2(V[lm][t], h);
              tp_bwd_half<L, false, U, 0>(ee, pwh, gg, a);
              __builtin_amdgcn_sched_barrier(0);
              if (kk == 0) {
#pragma unroll
                for (int lm = D - 1; lm >= 1; --lm) vin[lm] = vin[l_of_lm(lm)] * Y[lm];
              }
              tp_bwd_half<L, false, U, 1>(vin, pwh, gg, b);
            } else {
              f32x2 gg[1];
              gg[0] = half_of(ds[t], h);
              tp_bwd_half<L, true, U, 0>(ee, pwh, gg, a);
              __builtin_amdgcn_sched_barrier(0);
              if (kk == 0) {
#pragma unroll
                for (int lm = D - 1; lm >= 1; --lm) vin[lm] = vin[l_of_lm(lm)] * Y[lm];
              }
              tp_bwd_half<L, true, U, 1>(vin, pwh, gg, b);
            }
#pragma unroll
            for (int lm = 0; lm < D; ++lm) { acc_put2(V[lm][t], h, a[lm]); *(f32x2 *)(sp + lm * 16 + 2 * h) = b[lm]; }
            __builtin_amdgcn_sched_barrier(0);
          }
          if (t == UT - 1) {
            if (last) {
#pragma unroll
              for (int r = 0; r < L * UT; ++r) omall[r] = lrow_load(lds.rowsl[uwave], r, fresh_lane());
            } else load_rows<L * UT>(SB, RL + S::O_OM + UT, omall, v16);
          }
          __syncthreads();
          reduce_stage_x<L, UT, NW>(lds.stage[t & 1], aoffp, lds.denv, na, A.cenv, t, uwave);
          __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
      PHASEX(PX_BTP);
      // environment weights backward: d omega[l][u] = sum_m denv[lm][u] Y[lm];  dY[lm] += sum_u denv[lm][u] omega[l][u]
        f32x4 dom[EW];
        const float *dn = lds.denv + envoff;
#pragma unroll
        for (int t = 0; t < UT; ++t) {
          f32x4 omr[L + 1];
#pragma unroll
          for (int l = 1; l <= L; ++l) omr[l] = omall[(l - 1) * UT + t];
#pragma unroll
          for (int l = 0; l <= L; ++l) dom[l * UT + t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int lm = 0; lm < D; ++lm) {
            const f32x4 dv = *(const f32x4 *)(dn + lm * U + 16 * t);
            if (lm == 0) dom[t] = dv;
            else {
              dom[l_of_lm(lm) * UT + t] += dv * Y[lm];
              dY[lm] += hsum4(dv * omr[l_of_lm(lm)]);
            }
          }
#pragma unroll
          for (int lm = 1; lm < D; ++lm) pin(dY[lm]);        // see pin(): keeps the sums where they are computed
          __builtin_amdgcn_sched_barrier(0);
        }
        lx_prime<AR>(WB, wp, v16, ring);          // RING_DROP: requested again only now -- the d omega / dY arithmetic above needs the registers
        // next iteration's u / silu'(z2) rows (or the l >= 1 embedding weights for the last step) under this linear
        if (kk > 0) {
          if constexpr (!SAVEZ) load_rows<4>(SB, S::R_LAYER(kk - 1) + S::O_U, upre, v16);
          load_rows<4>(SB, S::R_LAYER(kk - 1) + S::O_Z2, zt, v16);
        } else load_rows<L * UT>(SB, S::R_W0 + UT, w0pre, v16);
        __builtin_amdgcn_sched_barrier(0);
        lx_lin<AR, EW, 4, true>(WB, wp, dom, dx, v16, ring, EpiNone{});
      }
      PHASEX(PX_BENV);
    }
    // ---------------- embedding backward: V holds dE/dV^0 ----------------
    {
      f32x4 dw0[EW];
#pragma unroll
      for (int t = 0; t < UT; ++t) {
        f32x4 w0r[L + 1];
#pragma unroll
        for (int l = 1; l <= L; ++l) w0r[l] = w0pre[(l - 1) * UT + t];
#pragma unroll
        for (int l = 0; l <= L; ++l) dw0[l * UT + t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int lm = 0; lm < D; ++lm) {
          const f32x4 dv = acc_get4(V[lm][t]);
          if (lm == 0) dw0[t] = dv;
          else {
            dw0[l_of_lm(lm) * UT + t] += dv * Y[lm];
            dY[lm] += hsum4(dv * w0r[l_of_lm(lm)]);
          }
        }
#pragma unroll
        for (int lm = 1; lm < D; ++lm) pin(dY[lm]);
        __builtin_amdgcn_sched_barrier(0);
      }
      lx_lin<AR, EW, 4, true>(WB, wp, dw0, dx, v16, ring, EpiNone{});
      wp = A.o_stream;                                                     // last linear of the tile (wrap-around copy follows it)
    }
    PHASEX(PX_BEMB);
    // ---------------- two-body embedding backward ----------------
    float dd_part;
    {
      f32x4 rows[4];
      load_rows<4>(SB, S::R_DX0, rows, v16);
      f32x4 accv = dx[0] * rows[0];
#pragma unroll
      for (int t = 1; t < 4; ++t) accv += dx[t] * rows[t];
      dd_part = hsum4(accv);
      pin(dd_part);
    }
    // ---------------- geometry backward, outputs ----------------
    {
      const float ibs = AR == 3 ? lds.stage[0][s * STG_LD + D * 16] : 1.f;      // this slot's inverse backward scale (per centre type, written at the start of the backward pass)
      const float dfc_tot = gsum(dfc_part) * ibs;
      const float dd = dfc_tot * (dfc_dx / rc) + gsum(dd_part) * ibs;
      float yv[D];
#pragma unroll
      for (int lm = 1; lm < D; ++lm) yv[lm] = gsum(dY[lm]) * ibs;
      // G = sum_lm dE/dY_lm * dY_lm/dn (n treated as a free vector), then projected onto the sphere
      float Gx = C3 * yv[3], Gy = C3 * yv[1], Gz = C3 * yv[2];
      if constexpr (L >= 2) {
        Gx += C15 * (yv[4] * ny + yv[7] * nz + yv[8] * nx) - 2.f * C5H * yv[6] * nx;
        Gy += C15 * (yv[4] * nx + yv[5] * nz - yv[8] * ny) - 2.f * C5H * yv[6] * ny;
        Gz += C15 * (yv[5] * ny + yv[7] * nx) + 4.f * C5H * yv[6] * nz;
      }
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
        lds.virw[wave][lane] += (double)mine;
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
    PHASEX(PX_FIN);
    if (++ck == A.tchunk) { ck = 0; cpar ^= 1; cbase = __builtin_amdgcn_readfirstlane(lds.chunk[cpar]); }
  }
  __syncthreads();
  if (tid == 0) {
    for (int a = 0; a < MAXA; ++a) acc_part += lds.eacc[a];
  } else if (tid >= 64 && tid < 70) {
    for (int w = 0; w < NW; ++w) acc_part += lds.virw[w][tid - 64];
  }
  if (PROF && lane == 0) {
#pragma unroll
    for (int k = 0; k < PX_N; ++k) atomicAdd((unsigned long long *)&A.prof[k], (unsigned long long)pacc[k]);
  }
  if (tid == 0) A.partial[7 * (size_t)blockIdx.x] = acc_part;
  if (tid >= 64 && tid < 70) A.partial[7 * (size_t)blockIdx.x + 1 + (tid - 64)] = acc_part;
}

// ---------------------------------------------------------------------------- host side

bool fusedlx_model_supported(const Model &m, std::string *why) {
  const HostModel &h = m.hm;
  auto no = [&](const char *msg) { if (why) *why = msg; return false; };
  if (h.l_max != 2) return no("wide fused kernels are built for l_max = 2");
  if (!fused_widths_fit(h)) return no("wide fused kernels hold at most 64 tensor features, S=64, MLP width 64, read-out width 32 (narrower models run zero-padded)");
  if (h.mlp_depth != 2 || h.readout_depth != 1) return no("fused kernels need MLP depth 2 and read-out depth 1");
  if (h.num_bessels < 1) return no("no radial basis");      // any number of Bessel functions: the two-body embedding is always tabulated here
  if (h.num_layers < 1 || h.num_layers > LX_MAXNL) return no("fused kernels need 1..3 layers");
  if (h.num_types > 16) return no("fused kernels support at most 16 model types (4-bit packed edge types)");
  return true;
}

template <int L, int UT> static void fusedlx_prepare_t(Model &m, FusedLxState &st) {
  using S = ShapeX<L, UT, 4>;
  const HostModel &h = fused_host_model(m);          // at the kernel's fixed widths (zero-padded when the model is narrower)
  const int T = h.num_types, NL = h.num_layers, U = S::U, D = S::D;
  std::vector<float> w;
  FusedLxArgs &A = st.args;
  std::memset(&A, 0, sizeof(A));
  auto mark = [&]() { while (w.size() % 64) w.push_back(0.f); return (int)w.size(); };
  auto T_ = [&](const std::string &name) -> const double * { return h.get(name).data.data(); };
  st.arith = lx_arith_of(m);
  int h_flags = 0;        // float16 range findings over the weight stream (engine.h: H_RANGE_*)
  auto frag = [&](const double *W, int K, int N, int ldw) {
    if (st.arith == 3) h_flags |= append_frag_h(w, W, K, N, ldw);
    else append_frag(w, W, K, N, ldw);
  };
  auto fwd = [&](const double *W, int K, int N) { frag(W, K, N, N); };
  auto bwd = [&](const double *W, int K, int N) { auto t = transpose(W, K, N); frag(t.data(), N, K, K); };
  // one channel-mixing row; with 32 features the last row is padded to 64 output columns (8 fragments, see mix_rows)
  auto mixfrag = [&](const double *Wl, int lm, bool transposed) {
    std::vector<double> m2((size_t)U * U);
    for (int a = 0; a < U; ++a)
      for (int b = 0; b < U; ++b) m2[(size_t)a * U + b] = transposed ? Wl[(size_t)b * U + a] : Wl[(size_t)a * U + b];
    if (UT == 2 && lm == D - 1 && (D % 2) == 1) {
      std::vector<double> pad((size_t)U * 2 * U, 0.0);
      for (int a = 0; a < U; ++a)
        for (int b = 0; b < U; ++b) pad[(size_t)a * 2 * U + b] = m2[(size_t)a * U + b];
      frag(pad.data(), U, 2 * U, 2 * U);
    } else frag(m2.data(), U, U, U);
  };
  // ---- the weight stream, in the order one tile consumes it (see k_fused_lx) ----
  A.o_stream = mark();
  const size_t stream0 = w.size();
  fwd(T_("emb.w"), 64, U * (L + 1));
  for (int k = 0; k < NL; ++k) {
    const std::string lk = "l" + std::to_string(k + 1);
    fwd(T_(lk + ".env"), 64, U * (L + 1));
    fwd(T_(lk + ".lat.w0"), 64 + U, 64);
    fwd(T_(lk + ".lat.w1"), 64, 64);
    fwd(T_(lk + ".lat.w2"), 64, 64);
    if (k < NL - 1) {
      const double *mx = T_(lk + ".mix");            // [L+1][U][U]; block l serves its 2l+1 components
      for (int lm = 0; lm < D; ++lm) mixfrag(mx + (size_t)l_of_lm(lm) * U * U, lm, false);
    }
  }
  fwd(T_("out.w0"), 64, 32);
  bwd(T_("out.w0"), 64, 32);
  for (int k = NL - 1; k >= 0; --k) {
    const std::string lk = "l" + std::to_string(k + 1);
    bwd(T_(lk + ".lat.w2"), 64, 64);
    bwd(T_(lk + ".lat.w1"), 64, 64);
    bwd(T_(lk + ".lat.w0"), 64 + U, 64);
    if (k < NL - 1) {
      const double *mx = T_(lk + ".mix");
      for (int lm = 0; lm < D; ++lm) mixfrag(mx + (size_t)l_of_lm(lm) * U * U, lm, true);
    }
    bwd(T_(lk + ".env"), 64, U * (L + 1));
  }
  bwd(T_("emb.w"), 64, U * (L + 1));
  for (size_t i = 0; i < (size_t)RING * 256; ++i) w.push_back(w[stream0 + i]);      // wrap-around copy
  // two-body table
  A.tb_nk = 512;
  A.o_tbtab = mark();
  append_two_body_table(w, h, m.rcut_model_host, A.tb_nk);
  // small tables: path weights (last layer: only the scalar paths, the rest zero)
  A.o_tpl = mark();
  for (int k = 0; k < NL; ++k) {
    const HostTensor &tp = h.get("l" + std::to_string(k + 1) + ".tp");
    for (int p = 0; p < S::NP; ++p)
      for (int u = 0; u < U; ++u) w.push_back(p < tp.shape[0] ? (float)tp.data[(size_t)p * U + u] : 0.f);
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
    arith_range_verdict(m, h_flags);                 // auto: ArithDegraded (run_model falls back to the f32 instance); explicit f16x2: an overflow is an error
    A.err = alarm_word(m);
  }
  A.wave_scratch = (long long)S::R_TOTAL(NL) * ROW;
}

static void fusedlx_prepare(Model &m) {
  if (!m.fusedlx_state) m.fusedlx_state = new FusedLxState();
  FusedLxState &st = *(FusedLxState *)m.fusedlx_state;
  if (st.ready) return;
  st.L = m.hm.l_max; st.UT = fused_UF(m.hm) / 16;
  fusedlx_prepare_t<2, 2>(m, st);
  hipDeviceProp_t prop;
  AHIP_CHECK(hipGetDeviceProperties(&prop, m.device));
  st.ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  st.scratch.reserve((size_t)st.ncu * 4 * st.args.wave_scratch * sizeof(float));
  st.args.scratch = st.scratch.as<float>();
  st.partial.reserve((size_t)st.ncu * 7 * sizeof(double));
  st.ntiles.reserve(64);
  st.prof.reserve(64 * sizeof(long long));
  const char *pe = std::getenv("AHIP_FUSED_PROF");
  st.prof_on = pe && pe[0] == '1';
  st.ready = true;
}

bool fusedlx_run(Model &m, const ComputeArgs &a, std::string *why) {
  constexpr int NW = 4, SLOTS = 16 * NW;
  // (counts still in flight = single-pass edge build with heavy_thresh = SLOTS: every centre with more edges is listed, gets a tile of its own that the
  // kernel skips, and is evaluated by heavy_generic after the kernel has been enqueued -- the host reads the counts only then)
  if (!m.counts_pending && m.last_max_deg > SLOTS && (m.heavy_thresh != SLOTS || (long long)m.nheavy * 8 > m.inum)) {
    // centres with more than 64 edges are listed by the edge build and evaluated by the layer-at-a-time kernels; when the edge
    // build did not list them (two-pass fallback) or they are not a small minority, the whole system goes that way
    if (why) *why = "an atom has " + std::to_string(m.last_max_deg) + " edges (> " + std::to_string(SLOTS) + " per tile of the wide fused kernel)";
    return false;
  }
  if (m.edges_T_size != 4) { if (why) *why = "edge vectors are not float32"; return false; }
  if (fused_UF(m.hm) == 64) {
    return fusedlx2_run(m, a, why);          // 64 tensor features: the wave-pair kernel (fused_lx2.hip)
  }
  fusedlx_prepare(m);
  FusedLxState &st = *(FusedLxState *)m.fusedlx_state;
  m.last_fused_arith = st.arith;
  hipStream_t s = a.stream;
  const int inum = m.inum;
  const int maxa = ShapeX<2, 4, NW>::MAXA;
  const int grid = std::max(1, st.ncu - (m.reserve_wgs + 1) / 2);      // see fused.hip: slots left free for the exchange kernels
  lx_pack_tiles(m, st, a, SLOTS, maxa);
  FusedLxArgs A = st.args;
  A.wg_scratch = NW * A.wave_scratch;
  A.eoff = m.b_eoff.as<int>(); A.e_ii = m.b_eii.as<int>(); A.e_j = m.b_ej.as<int>();
  A.e_tt = m.b_ett.as<unsigned char>(); A.rvec = m.b_rvec.as<float>(); A.rcut = m.rcut_model_dev;
  lx_tile_args(m, st, A, SLOTS, maxa);
  // claims of TCHUNK tiles amortise the counter's round trip; with few tiles per workgroup the last claim decides the makespan
  // (10 648 Si atoms: 4 659 tiles on 512 workgroups = 12 instead of 10 tile times with claims of 4)
  A.tchunk = (lx_nedges_estimate(m) / 64 > (long long)grid * 256) ? TCHUNK : 1;
  A.f = a.f; A.eatom = a.eatom; A.partial = st.partial.as<double>();
  {
    StageTimer tm(m, "model_fused", s);
#define LX_LAUNCH(UTV, NLV, PROFV) do { if (st.arith == 3) hipLaunchKernelGGL((k_fused_lx<2, UTV, NW, NLV, PROFV, 3>), dim3(grid), dim3(NW * 64), 0, s, A); \
                                        else hipLaunchKernelGGL((k_fused_lx<2, UTV, NW, NLV, PROFV, 0>), dim3(grid), dim3(NW * 64), 0, s, A); } while (0)
#define LX_LAUNCH_NL(UTV) do { if (A.NL == 3) LX_LAUNCH(UTV, 3, false); else if (A.NL == 2) LX_LAUNCH(UTV, 2, false); else LX_LAUNCH(UTV, 1, false); } while (0)
    if (st.prof_on && A.NL == 3) {
      AHIP_CHECK(hipMemsetAsync(st.prof.p, 0, 64 * sizeof(long long), s));
      A.prof = st.prof.as<long long>();
      LX_LAUNCH(2, 3, true);
    } else LX_LAUNCH_NL(2);
#undef LX_LAUNCH_NL
#undef LX_LAUNCH
  }
  AHIP_CHECK(hipGetLastError());
  AHIP_CHECK(prim_sum_columns_f64(m.prim, st.partial.as<double>(), grid, 7, a.engvir, s));
  if (st.prof_on && A.NL == 3) {
    std::vector<long long> hp(PX_N);
    AHIP_CHECK(hipMemcpyAsync(hp.data(), st.prof.p, hp.size() * sizeof(long long), hipMemcpyDeviceToHost, s));
    AHIP_CHECK(hipStreamSynchronize(s));
    static const char *names[PX_N] = {"geom+tb", "embed", "env+reduce", "tp", "latent_mlp", "mix", "readout", "b_latent", "b_mix", "b_tp+reduce", "b_env", "b_embed", "finish"};
    double tot = 0;
    for (int k = 0; k < PX_N; ++k) tot += (double)hp[k];
    std::fprintf(stderr, "[ahip fused_lx prof] wave-cycles by phase (sum over %d waves):", grid * NW);
    for (int k = 0; k < PX_N; ++k) std::fprintf(stderr, " %s=%.1f%%", names[k], 100.0 * hp[k] / tot);
    std::fprintf(stderr, " | total=%.3g cycles\n", tot);
  }
  return true;
}

void fusedlx_free(Model &m) {
  if (!m.fusedlx_state) return;
  FusedLxState *st = (FusedLxState *)m.fusedlx_state;
  for (DevBuf *b : {&st->wbuf, &st->scratch, &st->seg_count, &st->seg_base, &st->tile_a0, &st->tile_e0, &st->centre, &st->ntiles, &st->partial, &st->prof}) b->release();
  delete st;
  m.fusedlx_state = nullptr;
}

}  // namespace ahip
