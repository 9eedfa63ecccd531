// Fused MFMA path: the whole Allegro model (forward + hand-derived backward) for a tile of centre
// atoms in ONE kernel launch, float32 compute on the gfx950 matrix cores.
//
// Mapping:
//  * tile  = consecutive centre atoms whose edges fit the edge slots of one workgroup: 4 waves / 64 slots (two
//            independent workgroups per CU, the default) or 8 waves / 128 slots; either way 2 waves per SIMD at
//            <= 256 registers.  Wave w owns slots 16w..16w+15; lane = (slot j = lane & 15, group g = lane >> 4).
//            Persistent workgroups claim chunks of consecutive tiles from a global counter (dynamic schedule).
//  * every per-edge feature vector lives in registers in the v_mfma_f32_16x16x4_f32 C/D layout:
//            tile t, register r of lane (j, g)  <->  feature 16 t + 4 g + r   (4 lanes share one edge).
//            With D = W^T-tile (rows = output features) x activations (cols = edges), register r of an
//            output tile is exactly the B operand of MFMA step r of the next layer: the MLP chains run
//            register-to-register with no LDS traffic and no shuffles.
//  * weights are pre-swizzled on the host into A-operand fragments (one coalesced 1 KiB dwordx4 load
//            feeds 4 MFMAs) and laid out as ONE stream in the exact order a tile consumes them (forward
//            weights, then the transposed copies in backward order), so the prefetch ring is a running pointer.
//  * the two-body embedding x0(d; type pair) is read from a per-pair cubic spline table built from the float64
//            MLP and its exact derivative (fused_prepare); option fused_tb=mlp evaluates it in the kernel.
//  * the only cross-edge coupling -- the per-centre environment sum and its gradient -- goes through
//            an LDS staging tile [slots][128 features] and a deterministic per-atom reduction.
//  * activations needed by the backward pass are stored as raw register images in a per-wave
//            private scratch (written and re-read by the same wave within the same tile: Infinity-Cache traffic).
//  * arithmetic of the linears (option fused_arith): f32-input MFMA; f16x2 (two float16 terms per operand, three f16-MFMA
//            products, f32 accumulate: fused_h.h); bf16x3 (exact 3-way bf16 split, six bf16-MFMA terms); tf32eq (two bf16 terms,
//            only for model files that set allow_tf32).
//
// Supported model shape (others run the generic path): l_max = 1, up to 32 tensor features, up to 64 scalars, MLP width up to 64, read-out width up to 32 (narrower than
// the kernel's fixed 32 / 64 / 64 / 32: zero-padded by the host, model_io.cpp: pad_host_model),
// MLPs of 2 hidden layers x 64 (1 or 3 hidden layers on the f16x2 instances with the tabulated two-body embedding: template parameter MD, round 5),
// read-out 1 x 32, <= 3 layers, <= 16 types; any number of Bessel functions and any cutoff-polynomial order (the radial basis only
// enters through the tabulated two-body embedding; fused_tb=mlp needs 8).  Reference graph:
// the TorchScript model executed at /root/reference/pair_nequip_allegro.cpp:409-430.
#include <hip/hip_runtime.h>

#include <cmath>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "../../include/allegro_hip.h"
#include "engine.h"
// weight fragments in flight per wave: 4 (two steps ahead) instead of the wide kernels' 8 -- the 16 registers are worth more to this kernel than the deeper prefetch
// (1 M-atom Si: 58.3 -> 56.8 ms; 2 fragments: slower than 4)
#define AHIP_RING 4
#include "fused_common.h"
#include "fused_h.h"
#include "prims.h"

// This file is compiled in two parts (Makefile; same options, halves the build time): AHIP_FUSED_PART 0 = the host side + the f32-input MFMA
// instances of k_fused, AHIP_FUSED_PART 1 = the bf16-split instances.  (Rounds 2-3 gave part 1 other compiler options because it computed wrong
// forces with part 0's: that was the store-data hazard described at fused_common.h: bstore, not the options.)
#ifndef AHIP_FUSED_PART
#define AHIP_FUSED_PART 0
#endif
namespace ahip {


// Two workgroup shapes (template parameter NW = waves per workgroup, 16 edge slots per wave):
//   NW = 4: 64-slot tiles, <= 6 centres, ~79 KB LDS -> TWO independent workgroups per CU.  Their barriers are
//           independent, so the two waves of a SIMD drift into different phases and one wave's MFMA chains run
//           under the other's tensor-product / reduction / memory phases.  Default.
//   NW = 8: 128-slot tiles, <= 12 centres, ~159 KB LDS, one workgroup per CU: for lists with 65..128 edges per centre.
[[maybe_unused]] static constexpr int MAX_TILE_SLOTS = 128;
static constexpr int MAXNL = 3;
static constexpr int STG_LD = 132;       // staging leading dimension: all 128 features of a slot; 16-byte aligned rows
static constexpr int ENV_LD = 132;       // per-atom environment row (the tensor product reads it as float4)


struct FusedArgs {
  // edge list
  const int *eoff, *e_ii, *e_j;
  const unsigned char *e_tt;     // per edge: (model type of centre) << 4 | (model type of neighbour)
  const int2 *centre;            // per centre ii: {atom index ilist[ii], model type}
  const float *rvec;
  const double *rcut;            // [T*T]
  int T, NL, p;
  float cenv;
  // tiles
  unsigned int *tile_counter;
  int tchunk;                    // tiles per claim of the dynamic schedule (1 for small systems: a workgroup's last claim sets the makespan)    // dynamic tile schedule (zeroed by k_pack_finish)
  const int *tile_a0, *tile_e0, *ntiles;   // tile t = centres [tile_a0[t], tile_a0[t+1]), edges [tile_e0[t], tile_e0[t+1])
  const int *maxdeg_sel;         // null, or the device word with the list's largest degree: the 4-wave shape runs iff it is <= 64, the 8-wave shape iff not (fused_run)
  // weights (offsets in floats into wbase)
  const float *wbase;
  int wbytes;
  int o_stream;                  // the per-tile weight-fragment stream (consumption order)
  int o_tbtab, tb_nk;            // tabulated two-body embedding: [pair][tb_nk intervals][tile 4][coef 4][16] cubic coefficients
  int o_tpl, o_pair, o_out1, o_scale, o_shift;
  int o_res[MAXNL];
  // scratch
  float *scratch;
  long long wg_scratch, wave_scratch;     // floats
  // outputs
  double *f, *eatom, *partial;            // partial [gridDim.x][7]
  long long *prof;                        // [PH_N] or unused
  float *dbg;                             // [E][8] per-edge diagnostics or null
  float cp[6];                            // cutoff polynomial: a, b, c of f = 1 - a x^p + b x^(p+1) - c x^(p+2) and a p, b (p + 1), c (p + 2) of its derivative (scalar registers, not per-lane values held over a tile)
  int *err;                               // host-mapped word: set when an edge gradient comes out non-finite (float16 range exceeded)
};

template <int NW, int NL = MAXNL> struct __attribute__((aligned(16))) Lds {
  static constexpr int SLOTS = 16 * NW;
  static constexpr int MAXA = NW == 4 ? 6 : 12;   // centre atoms per tile (LDS budget)
  float stage[SLOTS * STG_LD];
  float env[NL][MAXA * ENV_LD];
  float denv[MAXA * ENV_LD];
  float tp[NL][5 * 32];                   // tensor-product path weights [layer][path][u]
  float park[NW][8 * ROW];                // per-wave private park: V^{k+1} forward, dE/dV backward ([lm][t] images)
  double eacc[MAXA];                      // energy accumulated over this workgroup's tiles, per centre slot (one owner thread each)
  double virw[NW][6];                     // virial accumulated per wave (owner: lanes 0..5 of the wave)
  int aoff[2][MAXA + 2];                  // slot offsets of the tile's centres, double-buffered by tile parity
  float rc[16];                           // model cutoff table [T*T] for T <= 4 (more types: read from A.rcut, the LDS budget of two workgroups per CU is spent)
  float scale[16], shift[16];             // per-type energy scale / shift
  float res[NL][2];                       // residual update coefficients per layer
  int chunk[2];                           // first tile of the current / next claimed chunk
};

// ---------------------------------------------------------------------------- device helpers

// scratch row map (per wave, rows of 1 KiB)
__device__ __host__ constexpr int R_Z1TB() { return 0; }
__device__ __host__ constexpr int R_Z2TB() { return 4; }
__device__ __host__ constexpr int R_U0() { return 8; }
__device__ __host__ constexpr int R_W0() { return 12; }
// per layer: omega 4 | silu' of the MD hidden layers of the latent MLP, 4 each | u 4 | V_in 8   (MD = latent MLP depth, 1..3; 2 = the reference YAML's, 24 rows)
__device__ __host__ constexpr int R_LSZ(int MD) { return 16 + 4 * MD; }
__device__ __host__ constexpr int R_LAYER(int kk, int MD = 2) { return 16 + R_LSZ(MD) * kk; }
__device__ __host__ constexpr int R_TOTAL(int NL, int MD = 2) { return 16 + R_LSZ(MD) * NL; }

static constexpr float C_S3 = 1.7320508075688772f;
static constexpr float C_P1 = 0.5773502691896258f;     // (1,1,0): sqrt(1) * w3j = 1/sqrt(3)
static constexpr float C_P4 = 0.7071067811865476f;     // (1,1,1): sqrt(3) * w3j = eps_ijk / sqrt(2)

// ---- bf16x3 arithmetic: every f32 value is split EXACTLY into three bf16 terms (truncation: 8+8+8 significant
// bits), x = hi + mid + lo, and a product w*x is evaluated on the bf16 matrix cores as the six terms of weight
// 2^-16 and above:  w_hi x_hi + w_hi x_mid + w_mid x_hi + w_hi x_lo + w_mid x_mid + w_lo x_hi  (each bf16 product is
// exact in f32, accumulation is f32; the dropped terms are <= 3 * 2^-24 |w x|, the size of one f32 rounding).  One
// v_mfma_f32_16x16x32_bf16 (16 cycles on the matrix core, which -- unlike the f32-input MFMA -- runs beside the
// VALU) contracts 32 input features, so a K-step is a PAIR of 16-feature tiles: lane (j, g) supplies its 4
// registers of tile 2s and of tile 2s+1 as the 8 k-slots of its lane group.  The C/D layout is the same as the
// f32 form, so the register chaining of the f32 kernel carries over unchanged.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
struct Bop { u32x4 hi, mid, lo; };           // B operand of one K-step: 8 bf16 per lane and term

__device__ __forceinline__ f32x4 mfma_b(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// two f32 -> packed bf16 terms (low half = first value)
__device__ __forceinline__ void split2(float v0, float v1, unsigned &hi, unsigned &mid, unsigned &lo) {
  const unsigned a = __builtin_bit_cast(unsigned, v0), b = __builtin_bit_cast(unsigned, v1);
  hi = __builtin_amdgcn_perm(b, a, 0x07060302u);
  const float r0 = v0 - __builtin_bit_cast(float, a & 0xffff0000u), r1 = v1 - __builtin_bit_cast(float, b & 0xffff0000u);
  const unsigned ra = __builtin_bit_cast(unsigned, r0), rb = __builtin_bit_cast(unsigned, r1);
  mid = __builtin_amdgcn_perm(rb, ra, 0x07060302u);
  const float q0 = r0 - __builtin_bit_cast(float, ra & 0xffff0000u), q1 = r1 - __builtin_bit_cast(float, rb & 0xffff0000u);
  lo = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, q1), __builtin_bit_cast(unsigned, q0), 0x07060302u);
}
__device__ __forceinline__ Bop split_pair(const f32x4 &t0, const f32x4 &t1) {
  Bop b;
  unsigned h, m, l;
  split2(t0[0], t0[1], h, m, l); b.hi[0] = h; b.mid[0] = m; b.lo[0] = l;
  split2(t0[2], t0[3], h, m, l); b.hi[1] = h; b.mid[1] = m; b.lo[1] = l;
  split2(t1[0], t1[1], h, m, l); b.hi[2] = h; b.mid[2] = m; b.lo[2] = l;
  split2(t1[2], t1[3], h, m, l); b.hi[3] = h; b.mid[3] = m; b.lo[3] = l;
  return b;
}

// Streamed linear on the bf16x3 arithmetic.  in: KS K-steps (pairs of 16-feature tiles, already split); out: NT f32
// tiles through the same epilogue functors as linear_s; SPLIT additionally emits the outputs as the next linear's
// B operands (outb[p] = tiles 2p, 2p+1).  Fragments per (tile pair p, K-step): hi0 hi1 mid0 mid1 lo0 lo1, 1 KiB each,
// in consumption order; RINGB of them are in flight (a multiple of 6: every linear consumes a multiple of 6).
static constexpr int RINGB = 12;              // fragments in flight, three-term split (2 steps of 6)
static constexpr int RINGB2 = 8;              // two-term split (2 steps of 4): every linear of the tile sequence consumes a multiple of 8 fragments except the four mixing blocks (4 each, alternating phase) -- a deeper ring would need the per-layer fragment count to be a multiple of it
template <int NTERM> struct RingB { static constexpr int N = NTERM == 3 ? RINGB : RINGB2; };
// NTERM = 3: the float32-equivalent split above (six products).  NTERM = 2 ("tf32eq", used only when the model file sets allow_tf32,
// the reference's own licence for TF32-class arithmetic, pair_nequip_allegro.cpp:267-270): hi + mid of both operands, the three products
// w_hi x_hi + w_hi x_mid + w_mid x_hi; the dropped terms are <= 3 * 2^-17 |w x| (TF32 itself rounds both operands to 2^-11), the weight
// stream is 4 fragments per step = the bytes of the f32 stream, and all of it runs on the bf16 matrix cores beside the VALU.
template <int KS, int NT, bool ACC, bool SPLIT, int RP, class Epi, int NTERM = 3>
__device__ __forceinline__ void linear_b(__amdgpu_buffer_rsrc_t W, int &wp, const Bop (&in)[KS], f32x4 (&out)[NT], Bop (&outb)[NT / 2],
                                         int v16, u32x4 (&ring)[RingB<NTERM>::N], Epi epi) {
  static_assert(NT % 2 == 0, "output tiles are processed in pairs");
  static_assert(NTERM == 2 || NTERM == 3, "two or three bf16 terms per operand");
  constexpr int RB = RingB<NTERM>::N;
  constexpr int NF = 2 * NTERM;                       // fragments per step: term-major, two output tiles each
  constexpr int NPROD = NTERM == 3 ? 6 : 3, MF = 2 * NPROD;      // products per accumulator, MFMAs per step
  constexpr int NP = NT / 2, NSTEP = NP * KS, NS = NF * NSTEP;
  f32x4 acc0, acc1, prev0, prev1;
  // byte offset of the next fragments to request: ONE running scalar advanced step by step, as in linear_s (written as wp + constant the optimiser
  // forms every offset of the tile at the top of the tile loop and spills them to VGPR lanes: 508 scalar spills in the two-layer instance)
  int wo = (wp + RB * 256) * 4;
  pin_s(wo);
#pragma unroll
  for (int s = 0; s < NSTEP; ++s) {
    const int p = s / KS, ks = s % KS;
    if (ks == 0) {
      if (ACC) { acc0 = out[2 * p]; acc1 = out[2 * p + 1]; }
      else { acc0 = f32x4{0.f, 0.f, 0.f, 0.f}; acc1 = f32x4{0.f, 0.f, 0.f, 0.f}; }
    }
    u32x4 a[NF];
#pragma unroll
    for (int i = 0; i < NF; ++i) {
      a[i] = ring[(RP + NF * s + i) % RB];
      ring[(RP + NF * s + i) % RB] = __builtin_bit_cast(u32x4, bload_w(W, v16 + (i & 3) * 1024, wo + (i >> 2) * 4096));      // + k KiB: the instruction's immediate offset
    }
    wo += NF * 1024;
    pin_s(wo);
    // the two accumulators alternate; smallest terms first
#pragma unroll
    for (int m = 0; m < NPROD; ++m) {
      // NTERM 3, term m: (weight term, activation term) = (lo,hi) (mid,mid) (hi,lo) (mid,hi) (hi,mid) (hi,hi);  NTERM 2: (mid,hi) (hi,mid) (hi,hi)
      const int wt = NTERM == 3 ? (m == 0 ? 2 : (m == 1 || m == 3) ? 1 : 0) : (m == 0 ? 1 : 0);
      const int xt = NTERM == 3 ? ((m == 0 || m == 3 || m == 5) ? 0 : (m == 1 || m == 4) ? 1 : 2) : (m == 1 ? 1 : 0);
      const u32x4 bx = xt == 0 ? in[ks].hi : xt == 1 ? in[ks].mid : in[ks].lo;
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        if (hh == 0) acc0 = mfma_b(a[2 * wt], bx, acc0);
        else acc1 = mfma_b(a[2 * wt + 1], bx, acc1);
        if (p > 0) {
          // the 8 epilogue elements of the previous pair are spread over the KS * MF MFMAs of this pair
          const int idx = ks * MF + 2 * m + hh, tot = KS * MF;
          const int e0 = (idx * 8 + tot - 1) / tot, e1 = ((idx + 1) * 8 + tot - 1) / tot;
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (e >= e0 && e < e1) {
              if (e < 4) out[2 * (p - 1)][e] = epi.apply(2 * (p - 1), e, prev0[e]);
              else out[2 * (p - 1) + 1][e - 4] = epi.apply(2 * (p - 1) + 1, e - 4, prev1[e - 4]);
              if (e == 7) {
                epi.flush(2 * (p - 1));
                if (SPLIT) outb[p - 1] = split_pair(out[2 * (p - 1)], out[2 * (p - 1) + 1]);
              }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (ks == KS - 1) {
      epi.tile_done(2 * p, acc0);
      epi.tile_done(2 * p + 1, acc1);
      if (p == NP - 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { out[2 * p][r] = epi.apply(2 * p, r, acc0[r]); out[2 * p + 1][r] = epi.apply(2 * p + 1, r, acc1[r]); }
        epi.flush(2 * p);
        if (SPLIT) outb[p] = split_pair(out[2 * p], out[2 * p + 1]);
      } else { prev0 = acc0; prev1 = acc1; }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  wp += NS * 256;
  pin_s(wp);
}
template <int RB> __device__ __forceinline__ void ring_prime_b(__amdgpu_buffer_rsrc_t W, int wp, int v16, u32x4 (&ring)[RB]) {
  int wo = wp * 4;
  pin_s(wo);
#pragma unroll
  for (int j = 0; j < RB; ++j) ring[j] = __builtin_bit_cast(u32x4, bload_w(W, v16 + (j & 3) * 1024, wo + (j >> 2) * 4096));
}

// Per-centre sum of the staged tile: dst[a][f] = scale * sum_{slots of a} stage[slot][f], f < 128.
// 128 features x (NW/2) atoms per pass; 4 independent accumulators keep 4 LDS reads in flight.
template <int NW, int NL> __device__ __forceinline__ void reduce_stage(const Lds<NW, NL> &lds, const int *aoff, float *dst, int na, float scale, int tid) {
  const int fidx = tid & 127;
  for (int a = tid >> 7; a < na; a += NW / 2) {
    const int s0 = aoff[a], s1 = aoff[a + 1];
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

// One weight-fragment ring per arithmetic (see linear_s / linear_b); lin<> dispatches a linear of the tile sequence to
// the f32-input MFMA form or to the bf16x3 form (inputs are split into their three bf16 terms right here).
// AR: arithmetic of the tile's linears: 0 = f32-input MFMA, 1 = bf16x3 (three-term split, float32-equivalent), 2 = tf32eq (two-term bf16 split),
// 3 = f16x2 (two float16 terms, float32-equivalent inside float16's exponent range: fused_h.h)
template <int AR> struct RingT {
  int act = 1;                 // wave-uniform: 0 while the wave's 16 slots of the current tile hold no edge (its linears are skipped, see the tile loop)
  f32x4 f[AR != 0 ? 1 : RING];
  u32x4 b[AR == 0 ? 1 : (AR == 1 ? RINGB : AR == 2 ? RINGB2 : RINGH)];
};
template <int AR, int KT, int NT, bool ACC, int RPI, class Epi>
__device__ __forceinline__ void lin(__amdgpu_buffer_rsrc_t W, int &wp, const f32x4 (&in)[KT], f32x4 (&out)[NT], int v16, RingT<AR> &ring, Epi epi) {
  if (!ring.act) {             // (uniform) a wave without edges in this tile: no fragments, no MFMAs, no epilogue; finite outputs for the arithmetic around the linears
    if constexpr (!ACC) {
#pragma unroll
      for (int t = 0; t < NT; ++t) out[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    return;
  }
  if constexpr (AR == 3) {
    static_assert(KT % 2 == 0, "K-steps are pairs of 16-feature tiles");
    Hop b[1][KT / 2], unused[1][NT / 2];
#pragma unroll
    for (int ks = 0; ks < KT / 2; ++ks) b[0][ks] = split_pair_h(in[2 * ks], in[2 * ks + 1]);
    Epi ep[1] = {epi};
    linear_h<1, KT / 2, NT, ACC, false, (4 * RPI) % RINGH, Epi>(W, wp, b, reinterpret_cast<f32x4 (&)[1][NT]>(out), unused, v16, ring.b, ep);
  } else if constexpr (AR != 0) {
    static_assert(KT % 2 == 0, "K-steps are pairs of 16-feature tiles");
    constexpr int NTERM = AR == 1 ? 3 : 2;
    Bop b[KT / 2], unused[NT / 2];
#pragma unroll
    for (int ks = 0; ks < KT / 2; ++ks) b[ks] = split_pair(in[2 * ks], in[2 * ks + 1]);
    // ring phase: RPI counts the four 32x32 channel-mixing blocks, which consume less than a ring each (2 * NTERM fragments)
    linear_b<KT / 2, NT, ACC, false, (2 * NTERM * RPI) % RingB<NTERM>::N, Epi, NTERM>(W, wp, b, out, unused, v16, ring.b, epi);
  } else {
    linear_s<KT, NT, ACC, (4 * RPI) % RING, Epi>(W, wp, in, out, v16, ring.f, epi);
  }
}

// The three m components of an l = 1 row share ONE channel-mixing matrix: on the f16x2 arithmetic they go through the linear as three GROUPS that share every weight
// fragment (linear_h<3, ...>: one fragment load, three B operands, three accumulator sets) instead of three passes over three copies of the matrix in the stream:
// 8 KiB of fragments fewer per mixing block and direction and wave-tile.
template <int RPI, class Epi>
__device__ __forceinline__ void lin_m3(__amdgpu_buffer_rsrc_t W, int &wp, const f32x4 (&in0)[2], const f32x4 (&in1)[2], const f32x4 (&in2)[2], f32x4 (&out)[3][2], int v16,
                                       RingT<3> &ring, Epi (&epi)[3]) {
  if (!ring.act) {
#pragma unroll
    for (int q = 0; q < 3; ++q) { out[q][0] = f32x4{0.f, 0.f, 0.f, 0.f}; out[q][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    return;
  }
  Hop b[3][1], unused[3][1];
  b[0][0] = split_pair_h(in0[0], in0[1]); b[1][0] = split_pair_h(in1[0], in1[1]); b[2][0] = split_pair_h(in2[0], in2[1]);
  linear_h<3, 1, 2, false, false, (4 * RPI) % RINGH, Epi>(W, wp, b, out, unused, v16, ring.b, epi);
}

// MD: hidden layers of the latent MLP (allegro_mlp_hidden_layers_depth of /root/reference/tests/test_data/test_repro_allegro.yaml:94; 2 there).  1 and 3 exist for the
// f16x2 instances with the tabulated two-body embedding (round 5: the fused family widened by one axis); every MD-dependent piece below is `if constexpr`.
// What a tile reads from the edge lists: bounds (scalar), this lane's edge slot, this thread's centre, one slot offset.
struct TileIn {
  int a0, a1, e0, e1;
  float rx, ry, rz;
  int eii, jat, tt;
  int eoff;
};
__device__ __forceinline__ void tile_fetch(const FusedArgs &A, int tile, int ntiles, int s, int tid, TileIn &n) {
  n.a0 = n.a1 = n.e0 = n.e1 = 0;
  n.rx = 1.f; n.ry = 0.f; n.rz = 0.f;
  n.eii = 0; n.jat = 0; n.tt = 0; n.eoff = 0;
  if (tile >= ntiles) return;                      // (uniform)
  n.a0 = A.tile_a0[tile]; n.a1 = A.tile_a0[tile + 1]; n.e0 = A.tile_e0[tile]; n.e1 = A.tile_e0[tile + 1];
  const int e = n.e0 + s, na = n.a1 - n.a0;
  if (e < n.e1) {
    n.rx = A.rvec[3 * (size_t)e]; n.ry = A.rvec[3 * (size_t)e + 1]; n.rz = A.rvec[3 * (size_t)e + 2];
    n.eii = A.e_ii[e];
    n.jat = A.e_j[e];
    n.tt = A.e_tt[e];
  }
  if (tid <= na) n.eoff = A.eoff[n.a0 + tid];
}

template <int NW, bool PROF, int AR, bool TBT, int NLT, int MD = 2>
__global__ void __launch_bounds__(NW * 64, 2) k_fused(FusedArgs A) {
  constexpr int NTHREADS = NW * 64, MAXA = Lds<NW>::MAXA;
  // f16x2 instances (round 5): the last hidden layer of the latent MLP saves its RAW pre-activation rows and the layer's output rows u are not saved at all: the backward
  // pass needs u only for <u, g> (the cutoff gradient), and <u, g> = <silu(z), g W^T> falls out of its first linear's epilogue (EpiMulSiluZ), which rebuilds silu and silu'
  // from z -- 6 row stores and 6 row loads fewer per wave-tile of a two-layer model for ~10 VALU operations per value there: 45.8 -> 44.4 ms at 1 M Si atoms.
  constexpr bool SAVEZ = AR == 3;
  constexpr bool PARKV = true;             // (round 6; its A/B: profiles/r06_a_ab_k_fused_rows_and_spills.txt, 39.7 -> 37.8 ms)
  // The LAST layer's activation rows (silu' / pre-activations of its last two hidden layers) live in the wave's own slots of the staging tile instead of scratch rows:
  // the tile is idle from that layer's environment sum to its backward tensor product (EpiSiluSaveDL / EpiSiluSaveZL): 8 row stores and 8 row loads fewer per wave-tile.
  constexpr bool STGROWS = true;           // (36.0 -> 34.3 ms)
  constexpr int OZL = 4 + 4 * (MD - 1), OU = 4 + 4 * MD, OVIN = 8 + 4 * MD;      // row offsets inside a layer: silu' of the LAST hidden layer, u, V_in (MD = 2: 8, 12, 16)
  static_assert(MD >= 1 && MD <= 3 && (MD == 2 || (AR == 3 && TBT)), "latent MLP depth 1 / 3: f16x2 instances with the two-body table only");
  __shared__ Lds<NW, NLT> lds;
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 15, g = lane >> 4, wave = tid >> 6;
  const int v16 = lane * 16;
  // wave-uniform buffer descriptors (made provably uniform with readfirstlane)
  __amdgpu_buffer_rsrc_t SB, WB;
  {
    unsigned long long b = (unsigned long long)(A.scratch + (size_t)blockIdx.x * A.wg_scratch + (size_t)wave * A.wave_scratch);
    unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    SB = __builtin_amdgcn_make_buffer_rsrc((void *)(((unsigned long long)hi << 32) | lo), 0, (int)(A.wave_scratch * 4), 0x00020000);
    WB = __builtin_amdgcn_make_buffer_rsrc((void *)A.wbase, 0, A.wbytes, 0x00020000);
  }
  if (A.maxdeg_sel && ((*A.maxdeg_sel <= 64) != (NW == 4))) return;     // both shapes were launched: the tiles were packed for the other one
  const float *__restrict__ Wb = A.wbase;
  for (int k = tid; k < A.NL * 160; k += NTHREADS) {      // path weights with their CG constants folded in
    const int pth = (k % 160) / 32;
    lds.tp[k / 160][k % 160] = Wb[A.o_tpl + k] * (pth == 1 ? C_P1 : pth == 4 ? C_P4 : 1.f);
  }
  const int ntiles = *A.ntiles;
  // The layer count is a template parameter and both layer loops are unrolled: "last layer" is then a compile-time fact (the folded
  // read-out below has a different shape there), nothing is carried around a loop edge, and the register allocation of each layer
  // is its own.  With a run-time loop the same fold cost more in spills than its 64 MFMAs saved (64.3 -> 67.7 ms; unrolled: 61.0).
  constexpr int NL = NLT;
  double acc_part = 0.0;       // thread 0: energy; threads 64..69: virial components
  long long pacc[PH_N];
  long long tprev = 0;
  const long long t_clk0 = clock64(), t_wall0 = wall_clock64();
  if (PROF) {
#pragma unroll
    for (int k = 0; k < PH_N; ++k) pacc[k] = 0;
    tprev = clock64();
  }
  float *const pk = lds.park[wave];
  RingT<AR> ring;                              // the weight-fragment stream (see linear_s / linear_b)
  int wp = A.o_stream;
  if constexpr (AR != 0) ring_prime_b(WB, wp, v16, ring.b);
  else ring_prime(WB, wp, v16, ring.f);
  if (tid < MAXA) lds.eacc[tid] = 0.0;
  if (lane < 6) lds.virw[wave][lane] = 0.0;
  if (A.T <= 4 && tid < A.T * A.T) lds.rc[tid] = (float)A.rcut[tid];
  if (tid < A.T) { lds.scale[tid] = Wb[A.o_scale + tid]; lds.shift[tid] = Wb[A.o_shift + tid]; }
  if (tid < 2 * A.NL) lds.res[tid >> 1][tid & 1] = Wb[A.o_res[tid >> 1] + (tid & 1)];

  const int s = wave * 16 + j;                 // this lane's edge slot
  float *const st = lds.stage + s * STG_LD;
  // Dynamic tile schedule: workgroups claim chunks of TCHUNK consecutive tiles from a global counter (workgroup
  // speeds differ by +-12 % across the chip, a static round-robin leaves the slowest one 13 % behind the average).
  // The next chunk is claimed while the first tile of the current one runs.
  if (tid == 0) lds.chunk[0] = (int)atomicAdd(A.tile_counter, (unsigned)A.tchunk);
  __syncthreads();
  int par = 0, cpar = 0, ck = 0;
  int cbase = __builtin_amdgcn_readfirstlane(lds.chunk[0]);

  TileIn nx_;
  tile_fetch(A, cbase, ntiles, s, tid, nx_);

  for (;;) {
    const int tile = cbase + ck;
    if (tile >= ntiles) break;
    int claimed = 0;
    if (ck == 0 && tid == 0) claimed = (int)atomicAdd(A.tile_counter, (unsigned)A.tchunk);
    // Everything a tile reads from the lists is ONE level of loads behind the (scalar) tile bounds: the edge
    // build packs the type pair per edge and k_tile_info the centre index/type, so no dependent chain
    // (edge -> centre -> type) is exposed here.  The whole set was requested a phase before the previous tile ended (tile_fetch below).
    const int a0 = nx_.a0, a1 = nx_.a1, e0 = nx_.e0, e1 = nx_.e1;
    // A wave whose 16 slots lie behind the tile's last edge (one 33..48-edge centre in a 64-slot tile: Li3PO4, 45 % of the tiles) skips every linear of the tile
    // (lin: ring.act) and addresses its saved rows out of range (loads return 0, stores are dropped); it still takes part in the barriers and in the per-centre
    // reductions, which never read its slots.  Its stream position needs no repair: a tile ends with wp = stream start and the ring holding the stream's first
    // fragments, which is what the wave's ring has held since its last active tile.
    // The branch pays twice.  Li3PO4 (102 400 atoms): 9.30 -> 7.77 ms.  And 1 M Si, where no wave is ever empty (56 of 64 slots): 44.7 -> 39.7 ms -- the same with
    // a condition that is always true (profiles/r05_w_last_experiments.md §6): a wave-uniform branch around each linear makes it a scheduling / allocation region of
    // its own (spills of the headline instance 304 -> 120 B per lane).  The wide kernels lose with the same branch (fused_lx2: 20.3 -> 21.9 ms): their linears are
    // ordered by hand across each other.
    const int wact = __builtin_amdgcn_readfirstlane(e0 + wave * 16 < e1 ? 1 : 0);
    ring.act = wact;
    const int v16t = wact ? v16 : 0x7ffffff0;
    const int na = a1 - a0;
    par ^= 1;
    int *const aoffp = lds.aoff[par];
    const int e = e0 + s;
    const bool valid = e < e1;
    const float rx = nx_.rx, ry = nx_.ry, rz = nx_.rz;
    const int aloc = valid ? nx_.eii - a0 : 0, jat = nx_.jat, ti = nx_.tt >> 4, tj = nx_.tt & 15;
    if (tid <= na) aoffp[tid] = nx_.eoff - e0;
    // ---------------- geometry ----------------
    const float d = sqrtf(rx * rx + ry * ry + rz * rz);
    const float inv = 1.f / d;
    const float nx = rx * inv, ny = ry * inv, nz = rz * inv;
    const float rc = A.T <= 4 ? lds.rc[ti * A.T + tj] : (float)A.rcut[ti * A.T + tj];
    const float xx = d / rc;
    float fc, dfc_dx;
    cutoff_poly_c(A.p, A.cp, xx, fc, dfc_dx);
    if (!valid) { fc = 0.f; dfc_dx = 0.f; }
    const float Y1 = C_S3 * ny, Y2 = C_S3 * nz, Y3 = C_S3 * nx;
    const float pref = 2.f / rc;
    const float PI = 3.14159265358979323846f;
    const float *const envrow = lds.env[0] + aloc * ENV_LD;        // + kk * MAXA*ENV_LD
    const float *const denvrow = lds.denv + aloc * ENV_LD;
    PHASE(PH_GEOM);

    // ---------------- two-body embedding x0(d; type pair) ----------------
    f32x4 x[4];
    float tb_t = 0.f, tb_invh = 0.f;
    if constexpr (TBT) {
      // x0 depends on the edge only through (d, t_i, t_j): the MLP [one-hots, Bessel * cutoff] -> 64 -> 64 -> 64, times the
      // cutoff, is tabulated per type pair as piecewise cubics in d (Hermite data from the float64 MLP and its exact
      // derivative, host side: fused_prepare).  Interpolation error (h^4 |4th derivative| / 384 at 512 intervals) is below
      // float32 rounding; it replaces 3 + 3 linears (18 % of the MFMAs), their SiLU epilogues and 12 saved rows per wave-tile.
      tb_invh = (float)A.tb_nk / rc;
      const float sft = d * tb_invh;
      const int kq = min((int)sft, A.tb_nk - 1);
      tb_t = sft - (float)kq;
      // the entry's byte offset inside the weight buffer (32 bits per lane; the 16 gathers differ in the instruction's immediate offset)
#ifdef ABL_NOTBGATHER   // timing experiment only (results are wrong): every edge reads the same table entry
      const int tb_off = (A.o_tbtab + 4 * g) * 4;
#else
      const int tb_off = (A.o_tbtab + ((ti * A.T + tj) * A.tb_nk + kq) * 256 + 4 * g) * 4;
#endif
      const float vm = (valid && xx < 1.f) ? 1.f : 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const f32x4 c0 = bload_w(WB, tb_off + (t * 4 + 0) * 64, 0), c1 = bload_w(WB, tb_off + (t * 4 + 1) * 64, 0);
        const f32x4 c2 = bload_w(WB, tb_off + (t * 4 + 2) * 64, 0), c3 = bload_w(WB, tb_off + (t * 4 + 3) * 64, 0);
        x[t] = (c0 + tb_t * (c1 + tb_t * (c2 + tb_t * c3))) * vm;
        // d x0 / dd for the backward pass: one coalesced row now instead of three per-edge gathers then
        // (gathering the entry again in the backward pass instead of saving these four rows -- 12 gathers for 4 stores + 4 loads -- was measured in round 6: 33.5 vs 33.2 ms)
        bstore(SB, v16t, (R_Z1TB() + t) * ROW * 4, (c1 + tb_t * (2.f * c2 + (3.f * tb_t) * c3)) * (vm * tb_invh));
      }
    } else {
      f32x4 z[4], z2[4];
      {
        const float *pt = Wb + A.o_pair + (size_t)(ti * A.T + tj) * 64;
#pragma unroll
        for (int t = 0; t < 4; ++t) z[t] = *(const f32x4 *)(pt + 16 * t + 4 * g);
      }
      f32x4 bfin[2];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float n = (float)(4 * g + r + 1);
        // Bessel numbers 9..16 (groups 2,3) meet zero weight rows; revolutions: sin(pi n x)
        bfin[0][r] = g < 2 ? pref * __builtin_amdgcn_sinf(0.5f * n * xx) * inv * fc : 0.f;
        bfin[1][r] = 0.f;
      }
      lin<AR, 2, 4, true, 0>(WB, wp, bfin, z, v16t, ring, EpiSiluSaveD{SB, R_Z1TB(), v16t});
      lin<AR, 4, 4, false, 0>(WB, wp, z, z2, v16t, ring, EpiSiluSaveD{SB, R_Z2TB(), v16t});
      lin<AR, 4, 4, false, 0>(WB, wp, z2, x, v16t, ring, EpiSaveScale{{SB, R_U0(), v16t}, fc});
    }
    PHASE(PH_TB);
    // ---------------- tensor embedding weights (V^0 = w0 (x) Y is rebuilt where needed) -------------
    {
      f32x4 w0[4];
      // w0 goes to scratch (backward) and to the LDS park rows 0..3, where layer 0 picks it up
      lin<AR, 4, 4, false, 0>(WB, wp, x, w0, v16t, ring, EpiSavePark{{SB, R_W0(), v16t}, pk, 0, lane});
    }
    if (ck == 0 && tid == 0) lds.chunk[cpar ^ 1] = claimed;
    __syncthreads();          // aoff visible; previous tile's LDS users done
    PHASE(PH_EMB);

    // ---------------- layers, forward ----------------
    f32x4 zr[2];               // read-out pre-activations: produced by the last layer (see its latent MLP)
#pragma unroll
    for (int kk = 0; kk < NL; ++kk) {
      const bool last = (kk == NL - 1);
      const int RL = R_LAYER(kk, MD);
      float *const envk = lds.env[0] + kk * (MAXA * ENV_LD);
      f32x4 V[4][2];
      {
        f32x4 om[4];
        lin<AR, 4, 4, false, 0>(WB, wp, x, om, v16t, ring, EpiSaveFrom2{{SB, RL + 0, v16t}});      // rows RL + 2, RL + 3: the l = 1 weights, all the backward pass reads
        // environment sum over the centre's edges
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          float *sp = st + 16 * t + 4 * g;                   // features 16 t + 4 g + (0..3): one 16-byte store per component
          *(f32x4 *)(sp) = om[t];
          *(f32x4 *)(sp + 32) = om[2 + t] * Y1;
          *(f32x4 *)(sp + 64) = om[2 + t] * Y2;
          *(f32x4 *)(sp + 96) = om[2 + t] * Y3;
        }
        __syncthreads();
        reduce_stage(lds, aoffp, envk, na, A.cenv, tid);
        __syncthreads();
      }
      if (kk == 0) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const f32x4 w1 = park_load(pk, 2 + t, lane);
          V[0][t] = park_load(pk, t, lane);
          V[1][t] = w1 * Y1; V[2][t] = w1 * Y2; V[3][t] = w1 * Y3;
        }
      } else {
#pragma unroll
        for (int lm = 0; lm < 4; ++lm)
#pragma unroll
          for (int t = 0; t < 2; ++t) V[lm][t] = park_load(pk, 2 * lm + t, lane);   // V^{kk} parked by the previous layer's mix
      }
      __builtin_amdgcn_sched_barrier(0);
      PHASE(PH_ENV);
      // tensor product
      f32x4 Vp[4][2];
      {
        const float *en = envrow + kk * (MAXA * ENV_LD);
        const float *tp = lds.tp[kk];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          // float4 arithmetic over the lane's 4 features of this tile: register pairs go to v_pk_* as they are
          const int b = 16 * t + 4 * g;
          const f32x4 e0v = *(const f32x4 *)(en + b), e1v = *(const f32x4 *)(en + 32 + b), e2v = *(const f32x4 *)(en + 64 + b),
                      e3v = *(const f32x4 *)(en + 96 + b);
          const f32x4 v0 = V[0][t], v1 = V[1][t], v2 = V[2][t], v3 = V[3][t];
          Vp[0][t] = *(const f32x4 *)(tp + b) * v0 * e0v + *(const f32x4 *)(tp + 32 + b) * (v1 * e1v + v2 * e2v + v3 * e3v);
          if (!last) {
            const f32x4 p2 = *(const f32x4 *)(tp + 64 + b), p3 = *(const f32x4 *)(tp + 96 + b), c4 = *(const f32x4 *)(tp + 128 + b);
            const f32x4 p2v0 = p2 * v0, p3e0 = p3 * e0v;
            Vp[1][t] = p2v0 * e1v + p3e0 * v1 + c4 * (v2 * e3v - v3 * e2v);
            Vp[2][t] = p2v0 * e2v + p3e0 * v2 + c4 * (v3 * e1v - v1 * e3v);
            Vp[3][t] = p2v0 * e3v + p3e0 * v3 + c4 * (v1 * e2v - v2 * e1v);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      PHASE(PH_TP);
      // channel mixing -> V^{kk+1}: saved in the next layer's VIN rows and parked in LDS
      if (!last) {
        f32x4 o2[2];
        if constexpr (AR == 3) {
          // f16x2: the l = 0 block, then the l = 1 block once for its three components (lin_m3)
          f32x4 o3[3][2];
          if (PARKV && kk + 1 == NL - 1) {                 // (the LAST layer's input tensor is not saved: see below)
            lin<AR, 2, 2, false, 0>(WB, wp, Vp[0], o2, v16t, ring, EpiPark{pk, 0, lane});
            EpiPark ep[3] = {EpiPark{pk, 2, lane}, EpiPark{pk, 4, lane}, EpiPark{pk, 6, lane}};
            lin_m3<1>(WB, wp, Vp[1], Vp[2], Vp[3], o3, v16t, ring, ep);
          } else {
            lin<AR, 2, 2, false, 0>(WB, wp, Vp[0], o2, v16t, ring, EpiSavePark{{SB, R_LAYER(kk + 1, MD) + OVIN + 0, v16t}, pk, 0, lane});
            EpiSavePark ep[3] = {EpiSavePark{{SB, R_LAYER(kk + 1, MD) + OVIN + 2, v16t}, pk, 2, lane}, EpiSavePark{{SB, R_LAYER(kk + 1, MD) + OVIN + 4, v16t}, pk, 4, lane},
                                 EpiSavePark{{SB, R_LAYER(kk + 1, MD) + OVIN + 6, v16t}, pk, 6, lane}};
            lin_m3<1>(WB, wp, Vp[1], Vp[2], Vp[3], o3, v16t, ring, ep);
          }
        } else if (PARKV && kk + 1 == NL - 1) {
          // the LAST layer's input tensor is not saved: nothing writes the park between its forward tensor product and its backward one (the last layer has
          // no channel mixing), so the backward pass reads it from there -- 8 row stores and 8 row loads fewer per wave-tile
          lin<AR, 2, 2, false, 0>(WB, wp, Vp[0], o2, v16t, ring, EpiPark{pk, 0, lane});
          lin<AR, 2, 2, false, 1>(WB, wp, Vp[1], o2, v16t, ring, EpiPark{pk, 2, lane});
          lin<AR, 2, 2, false, 2>(WB, wp, Vp[2], o2, v16t, ring, EpiPark{pk, 4, lane});
          lin<AR, 2, 2, false, 3>(WB, wp, Vp[3], o2, v16t, ring, EpiPark{pk, 6, lane});
        } else {
          lin<AR, 2, 2, false, 0>(WB, wp, Vp[0], o2, v16t, ring, EpiSavePark{{SB, R_LAYER(kk + 1, MD) + OVIN + 0, v16t}, pk, 0, lane});
          lin<AR, 2, 2, false, 1>(WB, wp, Vp[1], o2, v16t, ring, EpiSavePark{{SB, R_LAYER(kk + 1, MD) + OVIN + 2, v16t}, pk, 2, lane});
          lin<AR, 2, 2, false, 2>(WB, wp, Vp[2], o2, v16t, ring, EpiSavePark{{SB, R_LAYER(kk + 1, MD) + OVIN + 4, v16t}, pk, 4, lane});
          lin<AR, 2, 2, false, 3>(WB, wp, Vp[3], o2, v16t, ring, EpiSavePark{{SB, R_LAYER(kk + 1, MD) + OVIN + 6, v16t}, pk, 6, lane});
        }
      }
      PHASE(PH_MIX);
      // latent MLP
      {
        f32x4 cat[6], z[4], z2[4];
        cat[0] = x[0]; cat[1] = x[1]; cat[2] = x[2]; cat[3] = x[3]; cat[4] = Vp[0][0]; cat[5] = Vp[0][1];
        // hidden layer h (1..MD) saves silu'(z), the last one its raw z on the f16x2 instances (SAVEZ); in the last layer of the model the rows of hidden layers
        // MD - 1 and MD go to the staging tile (images 0..3 and 4..7) instead of scratch
        const bool stg = STGROWS && last;
        float *const sq = st + 4 * g;
#define AHIP_HIDDEN(KT, IN, OUT, H) do { \
          if (stg && (H) >= MD - 1) { \
            if (SAVEZ && (H) == MD) lin<AR, KT, 4, false, 0>(WB, wp, IN, OUT, v16t, ring, EpiSiluSaveZL{sq, 4}); \
            else lin<AR, KT, 4, false, 0>(WB, wp, IN, OUT, v16t, ring, EpiSiluSaveDL{sq, (H) == MD ? 4 : 0}); \
          } else if (SAVEZ && (H) == MD) lin<AR, KT, 4, false, 0>(WB, wp, IN, OUT, v16t, ring, EpiSiluSaveZ{SB, RL + 4 * (H), v16t}); \
          else lin<AR, KT, 4, false, 0>(WB, wp, IN, OUT, v16t, ring, EpiSiluSaveD{SB, RL + 4 * (H), v16t}); } while (0)
        AHIP_HIDDEN(6, cat, z, 1);
        if constexpr (MD >= 2) AHIP_HIDDEN(4, z, z2, 2);
        if constexpr (MD >= 3) AHIP_HIDDEN(4, z2, z, 3);
#undef AHIP_HIDDEN
        f32x4 (&zl)[4] = MD == 2 ? z2 : z;              // output of the last hidden layer
        const float ra = lds.res[kk][0], rbf = lds.res[kk][1] * fc;
        if (!last) {
          f32x4 xn[4];
          if constexpr (SAVEZ) lin<AR, 4, 4, false, 0>(WB, wp, zl, xn, v16t, ring, EpiResidualNS<4>{x, ra, rbf});
          else lin<AR, 4, 4, false, 0>(WB, wp, zl, xn, v16t, ring, EpiResidual<4>{{SB, RL + OU, v16t}, x, ra, rbf});
          x[0] = xn[0]; x[1] = xn[1]; x[2] = xn[2]; x[3] = xn[3];
        } else {
          // Last layer: its new latent x' = ra x + rb fc (z2 W3) feeds nothing but the read-out's first linear, and no non-linearity
          // sits between the two, so  x' Wr = ra (x Wr) + rb fc (z2 (W3 Wr))  with W3 Wr multiplied out by the host in float64: two
          // 64 -> 32 linears instead of a 64 -> 64 and a 64 -> 32 one, here and (transposed) in the backward pass: 64 of the tile's
          // 1472 MFMAs per wave and two saved rows less.  z2 first: it dies there.
          f32x4 za[2], up[2];
          if constexpr (SAVEZ) lin<AR, 4, 2, false, 0>(WB, wp, zl, up, v16t, ring, EpiNone{});
          else lin<AR, 4, 2, false, 0>(WB, wp, zl, up, v16t, ring, EpiSave{SB, RL + OU, v16t});
          lin<AR, 4, 2, false, 0>(WB, wp, x, za, v16t, ring, EpiNone{});
          zr[0] = ra * za[0] + rbf * up[0]; zr[1] = ra * za[1] + rbf * up[1];
        }
      }
      PHASE(PH_LAT);
    }

    // ---------------- read-out ----------------
    // saved rows are requested one linear ahead of their first use all through the backward pass (their
    // round trip is an L2 miss: ~2 us): u and z2 of the last layer now, under the read-out MFMAs
    f32x4 upre[4], zt[4], w0h[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) if (!SAVEZ) upre[t] = bload(SB, v16t, (R_LAYER(NL - 1, MD) + OU + t) * ROW * 4);       // the last layer saved two rows: z2 (W3 Wr)
    if (!STGROWS) load_rows<4>(SB, R_LAYER(NL - 1, MD) + OZL, zt, v16t);
    __builtin_amdgcn_sched_barrier(0);
    f32x4 wo1[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) wo1[t] = *(const f32x4 *)(Wb + A.o_out1 + 16 * t + 4 * g);
    float eps = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) eps += silu1(zr[t][r]) * wo1[t][r];
    eps = gsum(eps);
    asm volatile("" : "+v"(eps));            // computed here, not sunk to its use at the end of the tile (keeps the read-out rows alive until then)

    // =========================== backward ===========================
    // f16x2: the backward pass is linear in this upstream gradient and runs scaled by a power of two that brings it to O(1) (float16 has no exponent range to spare)
    // ... PER CENTRE TYPE (round 6): the exponent of this centre type's own upstream gradient, so that species whose energy scales differ by orders of
    // magnitude each run at O(1); every edge of a centre shares it, which is all the per-centre sums need.  Its inverse waits in the slot's pad floats of
    // the staging tile (st[128]: never staged, never reduced) instead of a register held to the end of the tile.
    float bsc = 1.f;
    if constexpr (AR == 3) {
      int bex;
      (void)frexpf(lds.scale[ti] * A.cenv, &bex);
      bsc = ldexpf(1.f, -bex);
      if (g == 0) st[128] = ldexpf(1.f, bex);
    }
    const float deps = valid ? lds.scale[ti] * A.cenv * bsc : 0.f;
    f32x4 dx[4];
    f32x4 dzr[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) dzr[t][r] = deps * wo1[t][r] * dsilu1(zr[t][r]);
    lin<AR, 2, 4, false, 0>(WB, wp, dzr, dx, v16t, ring, EpiNone{});               // dzr Wr^T = the gradient w.r.t. x' of the last layer
    if (STGROWS) {                                                                 // the last layer's rows of its last hidden layer, from the staging tile
#pragma unroll
      for (int t = 0; t < 4; ++t) zt[t] = stg_load(st + 4 * g, 4 + t);
    }
    float dfc_part = 0.f, dY1 = 0.f, dY2 = 0.f, dY3 = 0.f;
    PHASE(PH_OUT);

#pragma unroll
    for (int kk = NL - 1; kk >= 0; --kk) {
      const bool last = (kk == NL - 1);
      const int RL = R_LAYER(kk, MD);
      f32x4 dVp[4][2], Vk[4][2], W0b[4];
      {
        f32x4 du[4], dh[4];
        f32x4 zt1[4];
        if (!last && SAVEZ) {
          // the u rows are not saved: <u, g> = <silu(z), g W^T> comes out of the first backward linear's epilogue (EpiMulSiluZ), whose input is the unscaled gradient
          const float ra = lds.res[kk][0], rb = lds.res[kk][1];
          if constexpr (MD >= 2) load_rows<4>(SB, RL + OZL - 4, zt1, v16t);
          __builtin_amdgcn_sched_barrier(0);
          float ug = 0.f;
          lin<AR, 4, 4, false, 0>(WB, wp, dx, dh, v16t, ring, EpiMulSiluZ<4>{zt, rb * fc, ug});
          dfc_part += rb * ug;
#pragma unroll
          for (int t = 0; t < 4; ++t) dx[t] = ra * dx[t];
        } else if (last && SAVEZ) {
          const float ra = lds.res[kk][0], rb = lds.res[kk][1];
          if constexpr (MD >= 2) { if (!STGROWS) load_rows<4>(SB, RL + OZL - 4, zt1, v16t); }
          __builtin_amdgcn_sched_barrier(0);
          float ug = 0.f;
          lin<AR, 2, 4, false, 0>(WB, wp, dzr, dh, v16t, ring, EpiMulSiluZ<4>{zt, rb * fc, ug});
          dfc_part += rb * ug;
#pragma unroll
          for (int t = 0; t < 4; ++t) dx[t] = ra * dx[t];
        } else if (!last) {
          const float ra = lds.res[kk][0], rb = lds.res[kk][1];
          f32x4 accv = upre[0] * dx[0];
#pragma unroll
          for (int t = 1; t < 4; ++t) accv += upre[t] * dx[t];
          const float rbfc = rb * fc;
#pragma unroll
          for (int t = 0; t < 4; ++t) { du[t] = rbfc * dx[t]; dx[t] = ra * dx[t]; }
          dfc_part += rb * ((accv[0] + accv[1]) + (accv[2] + accv[3]));
          if constexpr (MD >= 2) load_rows<4>(SB, RL + OZL - 4, zt1, v16t);                  // silu' of the hidden layer below the last: first used 96 MFMAs from here
          __builtin_amdgcn_sched_barrier(0);
          lin<AR, 4, 4, false, 0>(WB, wp, du, dh, v16t, ring, EpiMulRows<4>{zt});
        } else {
          // last layer (read-out folded in, see the forward pass): upre = the two rows z2 (W3 Wr); the z2 gradient comes straight
          // from dzr through (W3 Wr)^T
          const float ra = lds.res[kk][0], rb = lds.res[kk][1];
          const f32x4 accv = upre[0] * dzr[0] + upre[1] * dzr[1];
          const float rbfc = rb * fc;
          f32x4 du2[2];
#pragma unroll
          for (int t = 0; t < 2; ++t) du2[t] = rbfc * dzr[t];
#pragma unroll
          for (int t = 0; t < 4; ++t) dx[t] = ra * dx[t];
          dfc_part += rb * ((accv[0] + accv[1]) + (accv[2] + accv[3]));
          if constexpr (MD >= 2) { if (!STGROWS) load_rows<4>(SB, RL + OZL - 4, zt1, v16t); }
          __builtin_amdgcn_sched_barrier(0);
          lin<AR, 2, 4, false, 0>(WB, wp, du2, dh, v16t, ring, EpiMulRows<4>{zt});
        }
        if constexpr (MD >= 2) {
          if (STGROWS && last) {                                 // silu' of the hidden layer below the last, from the staging tile
#pragma unroll
            for (int t = 0; t < 4; ++t) zt1[t] = stg_load(st + 4 * g, t);
          }
        }
        // prefetch V^{kk} (input of this layer's tensor product) under the MFMAs that follow
        if (kk > 0) {
          if (!(PARKV && last)) {
#pragma unroll
            for (int lm = 0; lm < 4; ++lm) load_rows<2>(SB, RL + OVIN + 2 * lm, Vk[lm], v16t);
          }
        } else load_rows<4>(SB, R_W0(), W0b, v16t);
        __builtin_amdgcn_sched_barrier(0);
        // down the hidden layers: g_{k-1} = (g_k W_k^T) * silu'(z_{k-1}), then dcat = g_0 W_0^T
        f32x4 dcat[6];
        if constexpr (MD == 1) lin<AR, 4, 6, false, 0>(WB, wp, dh, dcat, v16t, ring, EpiNone{});
        else {
          f32x4 zt0[4];
          if constexpr (MD == 3) load_rows<4>(SB, RL + 4, zt0, v16t);
          lin<AR, 4, 4, false, 0>(WB, wp, dh, du, v16t, ring, EpiMulRows<4>{zt1});
          if constexpr (MD == 3) {
            lin<AR, 4, 4, false, 0>(WB, wp, du, dh, v16t, ring, EpiMulRows<4>{zt0});
            lin<AR, 4, 6, false, 0>(WB, wp, dh, dcat, v16t, ring, EpiNone{});
          } else lin<AR, 4, 6, false, 0>(WB, wp, du, dcat, v16t, ring, EpiNone{});
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) dx[t] += dcat[t];
        dVp[0][0] = dcat[4]; dVp[0][1] = dcat[5];             // ds
      }
      if (kk == 0) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          Vk[0][t] = W0b[t];
          Vk[1][t] = W0b[2 + t] * Y1; Vk[2][t] = W0b[2 + t] * Y2; Vk[3][t] = W0b[2 + t] * Y3;
        }
      }
      if (PARKV && last && kk > 0) {                          // the last layer's input tensor, still in the park since its forward pass
#pragma unroll
        for (int lm = 0; lm < 4; ++lm)
#pragma unroll
          for (int t = 0; t < 2; ++t) Vk[lm][t] = park_load(pk, 2 * lm + t, lane);
      }
      f32x4 om1[2];
      load_rows<2>(SB, RL + 2, om1, v16t);                    // omega of this layer, l = 1 part: used after the gradient reduction
      __builtin_amdgcn_sched_barrier(0);
      PHASE(PH_BLAT);
      if (!last) {
        f32x4 in2[2], o2[2];
        in2[0] = park_load(pk, 0, lane); in2[1] = park_load(pk, 1, lane);
        lin<AR, 2, 2, false, 0>(WB, wp, in2, o2, v16t, ring, EpiNone{});
        dVp[0][0] += o2[0]; dVp[0][1] += o2[1];
        if constexpr (AR == 3) {
          f32x4 i1[2] = {park_load(pk, 2, lane), park_load(pk, 3, lane)}, i2[2] = {park_load(pk, 4, lane), park_load(pk, 5, lane)},
                i3[2] = {park_load(pk, 6, lane), park_load(pk, 7, lane)};
          f32x4 o3[3][2];
          EpiNone ep[3];
          lin_m3<1>(WB, wp, i1, i2, i3, o3, v16t, ring, ep);
#pragma unroll
          for (int q = 0; q < 3; ++q) { dVp[1 + q][0] = o3[q][0]; dVp[1 + q][1] = o3[q][1]; }
        } else {
          in2[0] = park_load(pk, 2, lane); in2[1] = park_load(pk, 3, lane);
          lin<AR, 2, 2, false, 1>(WB, wp, in2, dVp[1], v16t, ring, EpiNone{});
          in2[0] = park_load(pk, 4, lane); in2[1] = park_load(pk, 5, lane);
          lin<AR, 2, 2, false, 2>(WB, wp, in2, dVp[2], v16t, ring, EpiNone{});
          in2[0] = park_load(pk, 6, lane); in2[1] = park_load(pk, 7, lane);
          lin<AR, 2, 2, false, 3>(WB, wp, in2, dVp[3], v16t, ring, EpiNone{});
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      PHASE(PH_BMIX);
      // tensor-product backward: dV (w.r.t. V^{kk}, parked) and the per-edge environment gradient
      {
        const float *en = envrow + kk * (MAXA * ENV_LD);
        const float *tp = lds.tp[kk];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int b = 16 * t + 4 * g;
          const f32x4 e0v = *(const f32x4 *)(en + b), e1v = *(const f32x4 *)(en + 32 + b), e2v = *(const f32x4 *)(en + 64 + b),
                      e3v = *(const f32x4 *)(en + 96 + b);
          const f32x4 v0 = Vk[0][t], v1 = Vk[1][t], v2 = Vk[2][t], v3 = Vk[3][t];
          const f32x4 g0 = dVp[0][t];
          const f32x4 q0 = *(const f32x4 *)(tp + b) * g0, q1 = *(const f32x4 *)(tp + 32 + b) * g0;
          f32x4 a0v = q0 * e0v, a1v = q1 * e1v, a2v = q1 * e2v, a3v = q1 * e3v;       // dV
          f32x4 b0v = q0 * v0, b1v = q1 * v1, b2v = q1 * v2, b3v = q1 * v3;           // denv_e
          if (!last) {
            const f32x4 g1 = dVp[1][t], g2 = dVp[2][t], g3 = dVp[3][t];
            const f32x4 q2 = *(const f32x4 *)(tp + 64 + b), q3 = *(const f32x4 *)(tp + 96 + b), c4 = *(const f32x4 *)(tp + 128 + b);
            const f32x4 q3e0 = q3 * e0v, q2v0 = q2 * v0;
            a0v += q2 * (e1v * g1 + e2v * g2 + e3v * g3);
            b0v += q3 * (v1 * g1 + v2 * g2 + v3 * g3);
            a1v += q3e0 * g1 + c4 * (e2v * g3 - e3v * g2);      // (e x g)_1
            a2v += q3e0 * g2 + c4 * (e3v * g1 - e1v * g3);
            a3v += q3e0 * g3 + c4 * (e1v * g2 - e2v * g1);
            b1v += q2v0 * g1 + c4 * (g2 * v3 - g3 * v2);         // (g x v)_1
            b2v += q2v0 * g2 + c4 * (g3 * v1 - g1 * v3);
            b3v += q2v0 * g3 + c4 * (g1 * v2 - g2 * v1);
          }
          float *sp = st + b;
          *(f32x4 *)(sp) = b0v; *(f32x4 *)(sp + 32) = b1v; *(f32x4 *)(sp + 64) = b2v; *(f32x4 *)(sp + 96) = b3v;
          park_store(pk, 0 + t, a0v, lane);
          park_store(pk, 2 + t, a1v, lane);
          park_store(pk, 4 + t, a2v, lane);
          park_store(pk, 6 + t, a3v, lane);
          __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        reduce_stage(lds, aoffp, lds.denv, na, A.cenv, tid);
        __syncthreads();
      }
      __builtin_amdgcn_sched_barrier(0);
      PHASE(PH_BTP);
      {
        f32x4 dom[4];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int b = 16 * t + 4 * g;
          const f32x4 d0 = *(const f32x4 *)(denvrow + b), d1 = *(const f32x4 *)(denvrow + 32 + b), d2 = *(const f32x4 *)(denvrow + 64 + b),
                      d3 = *(const f32x4 *)(denvrow + 96 + b);
          dom[t] = d0;
          dom[2 + t] = d1 * Y1 + d2 * Y2 + d3 * Y3;
          const f32x4 p1 = d1 * om1[t], p2 = d2 * om1[t], p3 = d3 * om1[t];
          dY1 += (p1[0] + p1[1]) + (p1[2] + p1[3]); dY2 += (p2[0] + p2[1]) + (p2[2] + p2[3]); dY3 += (p3[0] + p3[1]) + (p3[2] + p3[3]);
          __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("" : "+v"(dY1), "+v"(dY2), "+v"(dY3));      // summed HERE: sunk to their use at the end of the tile, the sums keep a gradient row alive (in scratch) across the linear below
        if (kk > 0) {                                                      // next iteration's u and z2 rows
          if (!SAVEZ) load_rows<4>(SB, R_LAYER(kk - 1, MD) + OU, upre, v16t);
          load_rows<4>(SB, R_LAYER(kk - 1, MD) + OZL, zt, v16t);
        } else {                                                           // two-body u and z2 rows, l=1 embedding weights
          if constexpr (!TBT) {
            load_rows<4>(SB, R_U0(), upre, v16t);
            load_rows<4>(SB, R_Z2TB(), zt, v16t);
          } else load_rows<4>(SB, R_Z1TB(), zt, v16t);                      // d x0 / dd rows of the two-body table
          load_rows<2>(SB, R_W0() + 2, w0h, v16t);
        }
        __builtin_amdgcn_sched_barrier(0);
        lin<AR, 4, 4, true, 0>(WB, wp, dom, dx, v16t, ring, EpiNone{});
      }
      PHASE(PH_BENV);
    }
    // ---------------- embedding backward ----------------
    // The NEXT tile's list entries are requested here, two phases before they are used: at the top of a tile they were two dependent
    // round trips (bounds, then entries) behind the wrap-around weight fragments in the in-order return queue -- 8 % of the wave-cycles.
    // (The next chunk's first tile was published in lds.chunk by thread 0 during the chunk's first tile, several barriers ago.)
    {
      const int ntile = (ck + 1 == A.tchunk) ? __builtin_amdgcn_readfirstlane(lds.chunk[cpar ^ 1]) : tile + 1;
      tile_fetch(A, ntile, ntiles, s, tid, nx_);
    }
    f32x4 zt1b[4];
    if constexpr (!TBT) load_rows<4>(SB, R_Z1TB(), zt1b, v16t);
    __builtin_amdgcn_sched_barrier(0);
    {
      f32x4 dw0[4];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const f32x4 d0 = park_load(pk, 0 + t, lane), d1 = park_load(pk, 2 + t, lane), d2 = park_load(pk, 4 + t, lane), d3 = park_load(pk, 6 + t, lane);
        dw0[t] = d0;
        dw0[2 + t] = d1 * Y1 + d2 * Y2 + d3 * Y3;
#pragma unroll
        for (int r = 0; r < 4; ++r) { dY1 += d1[r] * w0h[t][r]; dY2 += d2[r] * w0h[t][r]; dY3 += d3[r] * w0h[t][r]; }
      }
      asm volatile("" : "+v"(dY1), "+v"(dY2), "+v"(dY3));
      lin<AR, 4, 4, true, 0>(WB, wp, dw0, dx, v16t, ring, EpiNone{});
      if constexpr (TBT) wp = A.o_stream;                                 // last linear of the tile
    }
    PHASE(PH_BEMB);
    // ---------------- two-body embedding backward ----------------
    float dd_part = 0.f;
    if constexpr (TBT) {
      // dE/dd through x0 = sum_f dE/dx0_f * d x0_f / dd (the derivative rows saved by the forward pass, cutoff included)
      f32x4 accv = dx[0] * zt[0];
#pragma unroll
      for (int t = 1; t < 4; ++t) accv += dx[t] * zt[t];
      dd_part = (accv[0] + accv[1]) + (accv[2] + accv[3]);
    } else {
      f32x4 du[4], dh[4];
      float acc = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) { acc += upre[t][r] * dx[t][r]; du[t][r] = fc * dx[t][r]; }
      dfc_part += acc;
      __builtin_amdgcn_sched_barrier(0);
      lin<AR, 4, 4, false, 0>(WB, wp, du, dh, v16t, ring, EpiMulRows<4>{zt});
      lin<AR, 4, 4, false, 0>(WB, wp, dh, du, v16t, ring, EpiMulRows<4>{zt1b});
      f32x4 dbf[2];
      lin<AR, 4, 2, false, 0>(WB, wp, du, dbf, v16t, ring, EpiNone{});   // its prefetches already fetch the next tile's first fragments
      wp = A.o_stream;                                                    // (the stream ends with a copy of its first RING entries)
      const float dfdd = dfc_dx / rc;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float n = (float)(4 * g + r + 1);
        // argument in revolutions for v_sin/v_cos (|arg| <= 4 for the 8 real Bessels): abs error ~1e-6, far
        // inside the force budget; groups 2,3 multiply an exact zero (zero-padded weight columns)
        const float sn = __builtin_amdgcn_sinf(0.5f * n * xx), cs = __builtin_amdgcn_cosf(0.5f * n * xx);
        const float b = pref * sn * inv;
        const float db = pref * (cs * PI * n / rc * inv - sn * inv * inv);
        dd_part += dbf[0][r] * (db * fc + b * dfdd);
      }
    }
    PHASE(PH_BTB);
    // ---------------- geometry backward, outputs ----------------
    {
      const float ibs = AR == 3 ? st[128] : 1.f;
      const float dfc_tot = gsum(dfc_part) * ibs;
      const float dd = dfc_tot * (dfc_dx / rc) + gsum(dd_part) * ibs;
      const float y1 = gsum(dY1) * ibs, y2 = gsum(dY2) * ibs, y3 = gsum(dY3) * ibs;
      const float Gx = C_S3 * y3, Gy = C_S3 * y1, Gz = C_S3 * y2;
      // unit vector and 1 / d again from the edge vector (the same expressions as at the top of the tile): four values less to hold across the whole tile
      float rxe = rx, rye = ry, rze = rz;
      asm volatile("" : "+v"(rxe), "+v"(rye), "+v"(rze));
      const float inve = 1.f / sqrtf(rxe * rxe + rye * rye + rze * rze);
      const float nxe = rxe * inve, nye = rye * inve, nze = rze * inve;
      const float gn = Gx * nxe + Gy * nye + Gz * nze;
      const float gx = dd * nxe + (Gx - gn * nxe) * inve;
      const float gy = dd * nye + (Gy - gn * nye) * inve;
      const float gz = dd * nze + (Gz - gn * nze) * inve;
      if (A.dbg && valid && g == 0) {
        float *dp = A.dbg + 8 * (size_t)e;
        dp[0] = gx; dp[1] = gy; dp[2] = gz; dp[3] = dd; dp[4] = dfc_tot; dp[5] = y1; dp[6] = y2; dp[7] = y3;
      }
#if !defined(ABL_NOW) && !defined(ABL_NOROWS) && !defined(ABL_NOROWLD) && !defined(ABL_NOROWST)
      if (AR == 3 && valid && !(fabsf(gx) + fabsf(gy) + fabsf(gz) + fabsf(eps) < 3.0e38f)) *A.err = 1;      // inf / NaN: an operand left float16's range
#endif
      if (g == 0) {
        st[0] = valid ? gx : 0.f; st[1] = valid ? gy : 0.f; st[2] = valid ? gz : 0.f; st[3] = valid ? eps : 0.f;      // selects, not products: a skipped wave's values are arbitrary
        if (valid) {
#ifndef ABL_NOATOM      // timing experiment only (results are wrong): no force scatter to the neighbour atoms
          atomicAdd(&A.f[3 * (size_t)jat], -(double)gx);
          atomicAdd(&A.f[3 * (size_t)jat + 1], -(double)gy);
          atomicAdd(&A.f[3 * (size_t)jat + 2], -(double)gz);
#endif
        }
      }
      // virial of this wave's 16 edges: butterfly over the slot lanes, lane 0 publishes 6 partials
      float w6[6] = {-rx * gx, -ry * gy, -rz * gz, -0.5f * (rx * gy + ry * gx), -0.5f * (rx * gz + rz * gx), -0.5f * (ry * gz + rz * gy)};
#pragma unroll
      for (int c = 0; c < 6; ++c) w6[c] = valid ? w6[c] : 0.f;
#pragma unroll
      for (int c = 0; c < 6; ++c) {
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) w6[c] += __shfl_xor(w6[c], off, 64);
      }
      if (lane < 6) {         // after the butterfly every lane holds the 6 totals: lane c owns component c
        const float mine = lane == 0 ? w6[0] : lane == 1 ? w6[1] : lane == 2 ? w6[2] : lane == 3 ? w6[3] : lane == 4 ? w6[4] : w6[5];
        lds.virw[wave][lane] += (double)mine;
      }
    }
    // everything the finish derives from the thread index is formed here, from an opaque copy: computed once at kernel entry, those addresses lived
    // the whole tile in scratch and their reloads drained the in-order load queue
    int tidf = tid;
    asm volatile("" : "+v"(tidf));
    const int caf = tidf >> 4;
    int2 cil = make_int2(0, 0);
    if (caf < na) cil = A.centre[a0 + caf];          // this thread's centre {atom, type}: fetched here, under the barrier, instead of held since the tile's top
    __syncthreads();
    {
      // per-centre sums of (g, eps): 16 lanes per atom = 4 columns x 4 row-parts; the atom index, scale and
      // shift of this thread's centre were prefetched with the tile.  No trailing barrier: the next tile's first
      // staging write sits behind its own barrier, and its slot offsets go to the other parity buffer.
      const int col = tidf & 3, part = (tidf >> 2) & 3;
      float sum = 0.f;
      if (caf < na)
        for (int sl = aoffp[caf] + part; sl < aoffp[caf + 1]; sl += 4) sum += lds.stage[sl * STG_LD + col];
      sum += __shfl_xor(sum, 4, 64);
      sum += __shfl_xor(sum, 8, 64);
      if (caf < na && part == 0) {
        if (col < 3) atomicAdd(&A.f[3 * (size_t)cil.x + col], (double)sum);
        else {
          const float ei = lds.scale[cil.y] * (sum * A.cenv) + lds.shift[cil.y];
          if (A.eatom) A.eatom[cil.x] = (double)ei;
          lds.eacc[caf] += (double)ei;
        }
      }
    }
    PHASE(PH_FIN);
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
    for (int k = 0; k < PH_N; ++k) atomicAdd((unsigned long long *)&A.prof[k], (unsigned long long)pacc[k]);
  }
  if (A.prof && tid == 0) {
    // per workgroup: shader cycles, start and end on the constant 100 MHz counter, hardware id (placement)
    long long *o = A.prof + PH_N + 4 * (size_t)blockIdx.x;
    o[0] = clock64() - t_clk0;
    o[1] = t_wall0;
    o[2] = wall_clock64();
    o[3] = ((long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32) | (unsigned)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
  }
  if (tid == 0) A.partial[7 * (size_t)blockIdx.x] = acc_part;
  if (tid >= 64 && tid < 70) A.partial[7 * (size_t)blockIdx.x + 1 + (tid - 64)] = acc_part;
}


// ---------------------------------------------------------------------------- the bf16-split instances (fused_bf.o)
void fused_launch_bf16(int nw, bool prof, int arith, bool tbt, int grid, hipStream_t s, const FusedArgs &A);
void fused_launch_f16(int nw, bool prof, int md, int grid, hipStream_t s, const FusedArgs &A);
#if AHIP_FUSED_PART == 2
// the f16x2 instances (fused_h.o): two-body table only; latent MLP depth 1..3 (profiling build for depth 2 only)
void fused_launch_f16(int nw, bool prof, int md, int grid, hipStream_t s, const FusedArgs &A) {
#ifdef AHIP_ASM_ONLY      // tools/asm_k_fused.sh: the headline instance alone, for a quick look at its code (static census, spills)
#ifndef AHIP_ASM_NW
#define AHIP_ASM_NW 4
#endif
  hipLaunchKernelGGL((k_fused<AHIP_ASM_NW, false, 3, true, 2, 2>), dim3(grid), dim3(AHIP_ASM_NW * 64), 0, s, A);
#else
#define AHIP_LAUNCH_NL(NWV, PROFV, NLV, MDV) hipLaunchKernelGGL((k_fused<NWV, PROFV, 3, true, NLV, MDV>), dim3(grid), dim3(NWV * 64), 0, s, A)
#define AHIP_LAUNCH(NWV, PROFV, MDV) do { if (A.NL == 1) AHIP_LAUNCH_NL(NWV, PROFV, 1, MDV); else if (A.NL == 2) AHIP_LAUNCH_NL(NWV, PROFV, 2, MDV); else AHIP_LAUNCH_NL(NWV, PROFV, 3, MDV); } while (0)
#define AHIP_LAUNCH_NW(PROFV, MDV) do { if (nw == 4) AHIP_LAUNCH(4, PROFV, MDV); else AHIP_LAUNCH(8, PROFV, MDV); } while (0)
  if (md == 1) AHIP_LAUNCH_NW(false, 1);
  else if (md == 3) AHIP_LAUNCH_NW(false, 3);
  else if (prof) AHIP_LAUNCH_NW(true, 2);
  else AHIP_LAUNCH_NW(false, 2);
#undef AHIP_LAUNCH_NW
#undef AHIP_LAUNCH
#undef AHIP_LAUNCH_NL
#endif
}
#endif
#if AHIP_FUSED_PART == 1
void fused_launch_bf16(int nw, bool prof, int arith, bool tbt, int grid, hipStream_t s, const FusedArgs &A) {
#define AHIP_LAUNCH_NL(NWV, PROFV, B3V, TBV, NLV) hipLaunchKernelGGL((k_fused<NWV, PROFV, B3V, TBV, NLV>), dim3(grid), dim3(NWV * 64), 0, s, A)
#define AHIP_LAUNCH(NWV, PROFV, B3V, TBV) do { if (A.NL == 1) AHIP_LAUNCH_NL(NWV, PROFV, B3V, TBV, 1); else if (A.NL == 2) AHIP_LAUNCH_NL(NWV, PROFV, B3V, TBV, 2); else AHIP_LAUNCH_NL(NWV, PROFV, B3V, TBV, 3); } while (0)
#define AHIP_LAUNCH_TB(NWV, PROFV, B3V) do { if (tbt) AHIP_LAUNCH(NWV, PROFV, B3V, true); else AHIP_LAUNCH(NWV, PROFV, B3V, false); } while (0)
#define AHIP_LAUNCH_NW(PROFV, B3V) do { if (nw == 4) AHIP_LAUNCH_TB(4, PROFV, B3V); else AHIP_LAUNCH_TB(8, PROFV, B3V); } while (0)
  if (prof) { if (arith == 1) AHIP_LAUNCH_NW(true, 1); else AHIP_LAUNCH_NW(true, 2); }
  else { if (arith == 1) AHIP_LAUNCH_NW(false, 1); else AHIP_LAUNCH_NW(false, 2); }
#undef AHIP_LAUNCH_NW
#undef AHIP_LAUNCH_TB
#undef AHIP_LAUNCH
#undef AHIP_LAUNCH_NL
}
#endif

#if AHIP_FUSED_PART == 0
// ---------------------------------------------------------------------------- host side
struct FusedState {
  DevBuf wbuf, scratch, seg_count, seg_base, tile_a0, tile_e0, centre, ntiles, partial;
  FusedArgs args;
  bool ready = false, prof_on = false, dbg_on = false, clk_on = false;
  bool tbt = true;             // two-body embedding from the spline table (default) or evaluated as an MLP (option fused_tb=mlp)
  int md = 2;                  // hidden layers of the latent MLP (template parameter MD of k_fused)
  int arith = 0;               // 0: f32-input MFMA; 1: bf16x3 (option fused_arith=bf16x3 / AHIP_FUSED_ARITH=b3); 2: tf32eq (two-term bf16 split; fused_arith=auto picks it when the model file says allow_tf32 = 1); 3: f16x2 (fused_h.h)
  DevBuf prof, dbg;
  int ncu = 256;
  int force_nw = 0;            // AHIP_FUSED_NW=4|8 pins the workgroup shape (A/B measurements)
};

// bf16x3 fragments of W [K][N]: per (tile pair p, K-step ks) six 1 KiB entries hi0 hi1 mid0 mid1 lo0 lo1; lane (i, g) of an
// entry holds the 8 k-slots of its group: slots 0..3 = rows 16 (2 ks) + 4 g + s, slots 4..7 = rows 16 (2 ks + 1) + 4 g + s - 4;
// column 16 (2 p + half) + i.  Terms by truncation (exact three-way split of the f32 weight).
static inline unsigned bf16_trunc_bits(float f) { unsigned u; std::memcpy(&u, &f, 4); return u >> 16; }
static inline float bf16_bits_to_f(unsigned h) { unsigned u = h << 16; float f; std::memcpy(&f, &u, 4); return f; }
static int append_frag_b(std::vector<float> &out, const double *W, int K, int N, int ldw, int nterm = 3) {
  int KS, NT;
  frag_dims_b(K, N, KS, NT);
  for (int p = 0; p < NT / 2; ++p)
    for (int ks = 0; ks < KS; ++ks)
      for (int term = 0; term < nterm; ++term)
        for (int half = 0; half < 2; ++half)
          for (int lane = 0; lane < 64; ++lane)
            for (int w = 0; w < 4; ++w) {
              unsigned word = 0;
              for (int e = 0; e < 2; ++e) {
                const int sl = 2 * w + e, g = lane >> 4;
                const int k = sl < 4 ? 16 * (2 * ks) + 4 * g + sl : 16 * (2 * ks + 1) + 4 * g + (sl - 4);
                const int n = 16 * (2 * p + half) + (lane & 15);
                const float v = (k < K && n < N) ? (float)W[(size_t)k * ldw + n] : 0.f;
                const unsigned hi = bf16_trunc_bits(v);
                const float r1 = v - bf16_bits_to_f(hi);
                const unsigned mid = bf16_trunc_bits(r1);
                const float r2 = r1 - bf16_bits_to_f(mid);
                const unsigned lo = bf16_trunc_bits(r2);
                const unsigned t = term == 0 ? hi : term == 1 ? mid : lo;
                word |= t << (16 * e);
              }
              float f;
              std::memcpy(&f, &word, 4);
              out.push_back(f);
            }
  return 2 * nterm * KS * (NT / 2);
}
// arithmetic (FusedState::arith) and two-body mode that options, environment and model metadata resolve to
static int fused_resolve_arith(const Model &m, bool &tbt) {
  const HostModel &h = m.hm;
  const char *tb = std::getenv("AHIP_FUSED_TB");
  tbt = (tb ? std::string(tb) : m.opt_fused_tb) != "mlp";
  const std::string arith = arith_option(m);
  // auto: tf32eq iff the model file licenses it; else f16x2 -- unless the model has degraded to f32 (engine.h) or the self-check is running its f32 pass
  int a = (arith == "b3" || arith == "bf16x3") ? 1 : (arith == "tf32eq" || (arith == "auto" && h.allow_tf32 && m.arith_force != 0)) ? 2
          : (arith == "f16x2" || (arith == "auto" && arith_auto_is_f16x2(m))) ? 3 : 0;
  if (a == 3 && !tbt) a = 0;        // the f16x2 instances exist with the tabulated two-body embedding only
  return a;
}
bool fused_model_supported(const Model &m, std::string *why) {
  const HostModel &h = m.hm;
  auto no = [&](const char *msg) { if (why) *why = msg; return false; };
  if (h.l_max != 1) return no("fused kernels need l_max = 1");
  if (!fused_widths_fit(h)) return no("fused kernels hold at most U=32, S=64, MLP width 64, read-out width 32 (narrower models run zero-padded)");
  if (h.mlp_depth < 1 || h.mlp_depth > 3 || h.readout_depth != 1) return no("fused kernels need MLP depth 1..3 and read-out depth 1");
  if (h.mlp_depth != 2) {           // round 5: depth 1 and 3 on the f16x2 instances with the tabulated two-body embedding
    bool tbt;
    if (fused_resolve_arith(m, tbt) != 3) return no("MLP depth 1 / 3 runs on the f16x2 arithmetic with the tabulated two-body embedding only (fused_arith=auto|f16x2, fused_tb=table, allow_tf32 = 0)");
  }
  // the radial basis only enters through the two-body embedding: tabulated (default) any number of Bessel functions will do, evaluated in the kernel
  // (fused_tb=mlp) its first linear is laid out for 8
  {
    const char *tb = std::getenv("AHIP_FUSED_TB");
    const bool in_kernel = (tb ? std::string(tb) : m.opt_fused_tb) == "mlp";
    if (h.num_bessels < 1 || (in_kernel && h.num_bessels != 8)) return no("fused_tb=mlp needs 8 Bessel functions (the tabulated two-body embedding takes any number)");
  }
  if (h.num_layers < 1 || h.num_layers > MAXNL) return no("fused kernels need 1..3 layers");
  if (h.num_types > 16) return no("fused kernels support at most 16 model types (4-bit packed edge types)");
  return true;
}

static void fused_prepare(Model &m) {
  if (!m.fused_state) m.fused_state = new FusedState();
  FusedState &st = *(FusedState *)m.fused_state;
  if (st.ready) return;
  const HostModel &h = fused_host_model(m);          // the model at the kernel's fixed widths (zero-padded when it is narrower)
  const int T = h.num_types, NL = h.num_layers;
  std::vector<float> w;
  FusedArgs &A = st.args;
  std::memset(&A, 0, sizeof(A));
  auto mark = [&]() { while (w.size() % 64) w.push_back(0.f); return (int)w.size(); };
  // two-body: pair table (type-type rows of the first layer)
  const HostTensor &w0 = h.get("tb.w0");          // [2T+8][64]
  A.o_pair = mark();
  for (int ti = 0; ti < T; ++ti)
    for (int tj = 0; tj < T; ++tj)
      for (int n = 0; n < 64; ++n) w.push_back((float)(w0.data[(size_t)ti * 64 + n] + w0.data[(size_t)(T + tj) * 64 + n]));
  // ---- the weight stream, in the order one tile consumes it ----
  A.o_stream = mark();
  const size_t stream0 = w.size();
  st.arith = fused_resolve_arith(m, st.tbt);
  st.md = h.mlp_depth;
  const int MD = st.md;
  const bool b3 = st.arith == 1 || st.arith == 2, tbt = st.tbt;
  const int nterm = st.arith == 1 ? 3 : 2;
  int h_flags = 0;        // float16 range findings over the weight stream (engine.h: H_RANGE_*)
  auto fwd = [&](const double *W, int K, int N) {
    if (st.arith == 3) h_flags |= append_frag_h(w, W, K, N, N);
    else if (b3) append_frag_b(w, W, K, N, N, nterm); else append_frag(w, W, K, N, N);
  };
  auto bwd = [&](const double *W, int K, int N) {
    auto t = transpose(W, K, N);
    if (st.arith == 3) h_flags |= append_frag_h(w, t.data(), N, K, K);
    else if (b3) append_frag_b(w, t.data(), N, K, K, nterm); else append_frag(w, t.data(), N, K, K);
  };
  auto T_ = [&](const std::string &name) -> const double * { return h.get(name).data.data(); };
  const double *wc = w0.data.data() + (size_t)2 * T * 64;       // Bessel block [8][64]
  std::vector<double> w3r((size_t)64 * 32, 0.0);                // W3 (last layer) @ Wr (read-out), float64
  {
    const double *W3 = T_("l" + std::to_string(NL) + ".lat.w" + std::to_string(MD)), *Wr = T_("out.w0");
    for (int i = 0; i < 64; ++i)
      for (int q = 0; q < 64; ++q)
        for (int n = 0; n < 32; ++n) w3r[(size_t)i * 32 + n] += W3[(size_t)i * 64 + q] * Wr[(size_t)q * 32 + n];
  }
  if (!tbt && MD != 2) throw UnsupportedError("fused_tb=mlp needs MLP depth 2");
  if (!tbt) {
    fwd(wc, 8, 64);
    fwd(T_("tb.w1"), 64, 64);
    fwd(T_("tb.w2"), 64, 64);
  }
  fwd(T_("emb.w"), 64, 64);
  for (int k = 0; k < NL; ++k) {
    const std::string lk = "l" + std::to_string(k + 1);
    fwd(T_(lk + ".env"), 64, 64);
    if (k < NL - 1) {
      const double *mx = T_(lk + ".mix");            // [2][32][32]; the l=1 block serves m = -1, 0, 1
      fwd(mx, 32, 32);
      for (int c = 0; c < (st.arith == 3 ? 1 : 3); ++c) fwd(mx + 1024, 32, 32);      // f16x2: once, shared by the three components (k_fused: lin_m3)
    }
    fwd(T_(lk + ".lat.w0"), 96, 64);
    for (int hl = 1; hl < MD; ++hl) fwd(T_(lk + ".lat.w" + std::to_string(hl)), 64, 64);
    if (k < NL - 1) fwd(T_(lk + ".lat.w" + std::to_string(MD)), 64, 64);
    else {            // last layer: the read-out's first linear applied, multiplied into W3, to z2 and then to x (k_fused, latent MLP)
      fwd(w3r.data(), 64, 32);
      fwd(T_("out.w0"), 64, 32);
    }
  }
  bwd(T_("out.w0"), 64, 32);
  for (int k = NL - 1; k >= 0; --k) {
    const std::string lk = "l" + std::to_string(k + 1);
    if (k < NL - 1) bwd(T_(lk + ".lat.w" + std::to_string(MD)), 64, 64);
    else bwd(w3r.data(), 64, 32);
    for (int hl = MD - 1; hl >= 1; --hl) bwd(T_(lk + ".lat.w" + std::to_string(hl)), 64, 64);
    bwd(T_(lk + ".lat.w0"), 96, 64);
    if (k < NL - 1) {
      const double *mx = T_(lk + ".mix");
      bwd(mx, 32, 32);
      for (int c = 0; c < (st.arith == 3 ? 1 : 3); ++c) bwd(mx + 1024, 32, 32);
    }
    bwd(T_(lk + ".env"), 64, 64);
  }
  bwd(T_("emb.w"), 64, 64);
  if (!tbt) {
    bwd(T_("tb.w2"), 64, 64);
    bwd(T_("tb.w1"), 64, 64);
    bwd(wc, 8, 64);
  }
  {   // wrap-around copy: the last linear of a tile prefetches the first fragments of the next tile
    const size_t n = (size_t)(st.arith == 1 ? RINGB : st.arith == 2 ? RINGB2 : st.arith == 3 ? RINGH : RING) * 256;
    for (size_t i = 0; i < n; ++i) w.push_back(w[stream0 + i]);
  }
  // two-body embedding table (see k_fused): per type pair, cubic Hermite in d on [0, r_c(pair)] from the float64 MLP
  A.tb_nk = 512;
  A.o_tbtab = mark();
  if (tbt) append_two_body_table(w, h, m.rcut_model_host, A.tb_nk);
  // small tables
  A.o_tpl = mark();
  for (int k = 0; k < NL; ++k) {
    const HostTensor &tp = h.get("l" + std::to_string(k + 1) + ".tp");
    for (int p = 0; p < 5; ++p)
      for (int u = 0; u < 32; ++u) w.push_back(p < tp.shape[0] ? (float)tp.data[(size_t)p * 32 + u] : 0.f);
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
  hipDeviceProp_t prop;
  AHIP_CHECK(hipGetDeviceProperties(&prop, m.device));
  st.ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  // Persistent workgroups, 8 waves per CU either way (two per SIMD, 256 registers each).
  A.wave_scratch = (long long)R_TOTAL(NL, MD) * ROW;
  st.scratch.reserve((size_t)st.ncu * 8 * A.wave_scratch * sizeof(float));
  A.scratch = st.scratch.as<float>();
  st.partial.reserve((size_t)st.ncu * 2 * 7 * sizeof(double));
  if (const char *nwe = std::getenv("AHIP_FUSED_NW")) st.force_nw = std::atoi(nwe);
  st.ntiles.reserve(64);
  st.prof.reserve((64 + 4 * 2 * (size_t)st.ncu) * sizeof(long long));
  const char *pe = std::getenv("AHIP_FUSED_PROF");
  st.prof_on = pe && pe[0] == '1';
  st.clk_on = std::getenv("AHIP_FUSED_CLK") != nullptr;
  const char *de = std::getenv("AHIP_FUSED_DBG");
  st.dbg_on = de && de[0] == '1';
  st.ready = true;
}

bool fused_run(Model &m, const ComputeArgs &a, std::string *why) {
  // Tile shape (4 waves / 64 slots or 8 waves / 128 slots) follows the largest degree of THIS step's edge list.  The host does not read that
  // back: a degree cannot exceed the row length of the list it was filtered from, which is known since the list was handed
  // over -- rows <= 64: 4 waves, no question; rows <= 128: both shapes are launched and the device word decides (the other kernel returns
  // at once, ~3 us); only lists with longer rows (two-pass edge build, counts read back there) take the decision on the host.
  if (m.edges_T_size != 4) { if (why) *why = "edge vectors are not float32"; return false; }
  fused_prepare(m);
  FusedState &st = *(FusedState *)m.fused_state;
  if (st.prof_on || st.clk_on || st.dbg_on) edges_counts(m);      // instrumented runs size their buffers / reports from the counts
  int nw = 0;                                   // 0: decided on the device
  if (!m.counts_pending) {
    if (m.last_max_deg > MAX_TILE_SLOTS) {
      if (why) *why = "an atom has " + std::to_string(m.last_max_deg) + " edges (> 128 per tile)";
      return false;
    }
    nw = m.last_max_deg <= 64 ? 4 : 8;
    if (st.force_nw == 8 || (st.force_nw == 4 && m.last_max_deg <= 64)) nw = st.force_nw;
  } else if (st.force_nw == 8) nw = 8;
  else if (m.max_list_row >= 0 && m.max_list_row <= 64) nw = 4;
  const int *maxdeg_sel = nw == 0 ? m.d_maxdeg : nullptr;
  m.last_fused_arith = st.arith;
  hipStream_t s = a.stream;
  const int inum = m.inum;
  const int tile_slots = nw == 8 ? 128 : 64, maxa = nw == 8 ? Lds<8>::MAXA : Lds<4>::MAXA;      // nw == 0: the packing kernels widen them themselves
  const int nseg = (inum + SEG - 1) / SEG;
  st.seg_count.reserve((size_t)(nseg + 1) * sizeof(int));
  st.seg_base.reserve((size_t)(nseg + 2) * sizeof(int));
  st.tile_a0.reserve((size_t)(inum + nseg + 2) * sizeof(int));
  st.tile_e0.reserve((size_t)(inum + nseg + 2) * sizeof(int));
  static_assert(Lds<4>::MAXA == 6, "allegro_hip.hip requests the 4-wave tile shape as 64 slots / 6 centres");
  const bool prepacked = m.tiles_packed && nw == 4 && m.pack_slots == tile_slots && m.pack_maxa == maxa;      // the edge build packed the tiles (edges.hip)
  if (!prepacked) {
    StageTimer tm(m, "tile_pack", s);
    const unsigned B = 64;
    st.centre.reserve((size_t)std::max(inum, 1) * sizeof(int2));
    const bool small = inum <= PACK_SMALL_ATOMS;
    if (small)
      hipLaunchKernelGGL(k_pack_small, dim3(1), dim3(PACK_SMALL_SEGS), 0, s, inum, m.b_eoff.as<int>(), nseg, st.tile_a0.as<int>(), st.tile_e0.as<int>(), st.ntiles.as<int>(), tile_slots, maxa,
                         m.d_ilist, a.mtype, st.centre.as<int2>(), maxdeg_sel);
    else {
      hipLaunchKernelGGL(k_pack_tiles<false>, dim3((nseg + B - 1) / B), dim3(B), 0, s, inum, m.b_eoff.as<int>(), nseg, st.seg_count.as<int>(), (const int *)nullptr, (int *)nullptr, tile_slots, maxa, maxdeg_sel);
      AHIP_CHECK(prim_exclusive_scan_i32(m.prim, st.seg_count.as<int>(), st.seg_base.as<int>(), nseg, s));
      hipLaunchKernelGGL(k_pack_tiles<true>, dim3((nseg + B - 1) / B), dim3(B), 0, s, inum, m.b_eoff.as<int>(), nseg, (int *)nullptr, st.seg_base.as<int>(), st.tile_a0.as<int>(), tile_slots, maxa, maxdeg_sel);
      hipLaunchKernelGGL(k_pack_finish, dim3(1), dim3(1), 0, s, inum, nseg, st.seg_base.as<int>(), st.tile_a0.as<int>(), st.ntiles.as<int>());
      hipLaunchKernelGGL(k_centre_info, dim3((inum + 255) / 256), dim3(256), 0, s, inum, m.d_ilist, a.mtype, st.centre.as<int2>());
    }
    if (!m.have_ett) {            // two-pass edge build: its counts are on the host
      m.b_ett.reserve((size_t)std::max<long long>(m.nedges, 1));
      hipLaunchKernelGGL(k_edge_types, dim3((unsigned)((m.nedges + 255) / 256)), dim3(256), 0, s, m.nedges, m.b_eii.as<int>(), m.b_ej.as<int>(), m.d_ilist, a.mtype, m.b_ett.as<unsigned char>());
      m.have_ett = true;
    }
    if (!small) {
      const int tcap = inum + nseg + 1;              // upper bound on tiles + 1
      hipLaunchKernelGGL(k_tile_e0, dim3((tcap + 255) / 256), dim3(256), 0, s, st.ntiles.as<int>(), st.tile_a0.as<int>(), m.b_eoff.as<int>(), st.tile_e0.as<int>());
    }
  }
  FusedArgs A = st.args;
  A.eoff = m.b_eoff.as<int>(); A.e_ii = m.b_eii.as<int>(); A.e_j = m.b_ej.as<int>();
  A.e_tt = m.b_ett.as<unsigned char>(); A.rvec = m.b_rvec.as<float>(); A.rcut = m.rcut_model_dev;
  int *const ntl = prepacked ? m.b_ntiles.as<int>() : st.ntiles.as<int>();
  A.centre = prepacked ? m.b_centre.as<int2>() : st.centre.as<int2>();
  A.tile_a0 = prepacked ? m.b_tile_a0.as<int>() : st.tile_a0.as<int>(); A.tile_e0 = prepacked ? m.b_tile_e0.as<int>() : st.tile_e0.as<int>(); A.ntiles = ntl;
  A.tile_counter = (unsigned int *)(ntl + 1);
  A.maxdeg_sel = maxdeg_sel;
  m.d_ntiles_last = ntl; m.last_tile_slots = nw == 0 ? 0 : 16 * nw;
  A.f = a.f; A.eatom = a.eatom; A.partial = st.partial.as<double>();
  // edge total for the claim size below: the value itself when it is on the host, else the last one that was, else the list's size
  // scaled by the volume ratio of cutoff and list spheres at a skin of 1 A
  const long long nedges_est = !m.counts_pending ? m.nedges : m.nedges_hint > 0 ? m.nedges_hint : (long long)(0.58 * (double)m.nneigh);
  // persistent workgroups fill every CU; reserve_wgs leaves a few slots free so that the exchange kernels of another stream
  // (ghost pack / unpack, RCCL send / recv) can be scheduled while this kernel runs (md.py, overlapped schedule)
  const int grid4 = std::max(1, st.ncu * 2 - m.reserve_wgs), grid8 = std::max(1, st.ncu - m.reserve_wgs);
  const int grid = nw == 8 ? grid8 : grid4;     // rows of `partial` that are summed
  if (nw == 0) AHIP_CHECK(hipMemsetAsync(st.partial.p, 0, (size_t)grid * 7 * sizeof(double), s));     // the shape that returns at once writes nothing
  {
    StageTimer tm(m, "model_fused", s);
    if (st.dbg_on) {
      st.dbg.reserve((size_t)std::max<long long>(m.nedges, 1) * 8 * sizeof(float));
      AHIP_CHECK(hipMemsetAsync(st.dbg.p, 0, (size_t)m.nedges * 8 * sizeof(float), s));
      A.dbg = st.dbg.as<float>();
    }
    if (st.prof_on || st.clk_on) {
      AHIP_CHECK(hipMemsetAsync(st.prof.p, 0, (64 + 4 * (size_t)grid) * sizeof(long long), s));
      A.prof = st.prof.as<long long>();
    }
    for (int shape = 4; shape <= 8; shape += 4) {
      if (nw != 0 && nw != shape) continue;
      const int g = shape == 8 ? grid8 : grid4;
      A.wg_scratch = shape * A.wave_scratch;
      // claims of TCHUNK tiles amortise the counter's round trip; with few tiles per workgroup the last claim decides the makespan
      // (10 648 Si atoms: 4 659 tiles on 512 workgroups = 12 instead of 10 tile times with claims of 4)
      A.tchunk = (nedges_est / (16 * shape) > (long long)g * 256) ? TCHUNK : 1;
      if (const char *tc = std::getenv("AHIP_TCHUNK")) A.tchunk = std::max(1, std::atoi(tc));       // experiments
      if (st.arith == 3) { fused_launch_f16(shape, st.prof_on, st.md, g, s, A); continue; }                        // fused_h.o
      if (st.arith != 0) { fused_launch_bf16(shape, st.prof_on, st.arith, st.tbt, g, s, A); continue; }     // fused_bf.o
#define AHIP_LAUNCH_NL(NWV, PROFV, TBV, NLV) hipLaunchKernelGGL((k_fused<NWV, PROFV, 0, TBV, NLV>), dim3(g), dim3(NWV * 64), 0, s, A)
#define AHIP_LAUNCH(NWV, PROFV, TBV) do { if (A.NL == 1) AHIP_LAUNCH_NL(NWV, PROFV, TBV, 1); else if (A.NL == 2) AHIP_LAUNCH_NL(NWV, PROFV, TBV, 2); else AHIP_LAUNCH_NL(NWV, PROFV, TBV, 3); } while (0)
#define AHIP_LAUNCH_TB(NWV, PROFV) do { if (st.tbt) AHIP_LAUNCH(NWV, PROFV, true); else AHIP_LAUNCH(NWV, PROFV, false); } while (0)
#define AHIP_LAUNCH_NW(PROFV) do { if (shape == 4) AHIP_LAUNCH_TB(4, PROFV); else AHIP_LAUNCH_TB(8, PROFV); } while (0)
      if (st.prof_on) AHIP_LAUNCH_NW(true); else AHIP_LAUNCH_NW(false);
#undef AHIP_LAUNCH_NW
#undef AHIP_LAUNCH_TB
#undef AHIP_LAUNCH
#undef AHIP_LAUNCH_NL
    }
  }
  AHIP_CHECK(hipGetLastError());
  AHIP_CHECK(prim_sum_columns_f64(m.prim, st.partial.as<double>(), grid, 7, a.engvir, s));
  if (st.prof_on || st.clk_on) {
    std::vector<long long> hp(PH_N + 4 * (size_t)grid);
    AHIP_CHECK(hipMemcpyAsync(hp.data(), st.prof.p, hp.size() * sizeof(long long), hipMemcpyDeviceToHost, s));
    AHIP_CHECK(hipStreamSynchronize(s));
    if (st.prof_on) {
      static const char *names[PH_N] = {"geom", "tb_mlp", "embed", "env+reduce", "tp", "mix", "latent_mlp", "readout", "b_latent", "b_mix", "b_tp+reduce", "b_env", "b_embed", "b_tb", "finish"};
      double tot = 0;
      for (int k = 0; k < PH_N; ++k) tot += (double)hp[k];
      std::fprintf(stderr, "[ahip fused prof] wave-cycles by phase (sum over %d waves):", grid * nw);
      for (int k = 0; k < PH_N; ++k) std::fprintf(stderr, " %s=%.1f%%", names[k], 100.0 * hp[k] / tot);
      std::fprintf(stderr, " | total=%.3g cycles\n", tot);
    }
    {
      // workgroup timeline: spread of start/end times, effective clock, and (AHIP_FUSED_CLK=2) one line per workgroup
      long long t0 = hp[PH_N + 1], t1 = hp[PH_N + 2];
      for (int b = 0; b < grid; ++b) { t0 = std::min(t0, hp[PH_N + 4 * b + 1]); t1 = std::max(t1, hp[PH_N + 4 * b + 2]); }
      double dmin = 1e30, dmax = 0, dsum = 0, smax = 0, mhz = 0;
      for (int b = 0; b < grid; ++b) {
        const long long *o = &hp[PH_N + 4 * b];
        const double d = (double)(o[2] - o[1]) * 1e-5;
        dmin = std::min(dmin, d); dmax = std::max(dmax, d); dsum += d;
        smax = std::max(smax, (double)(o[1] - t0) * 1e-5);
        mhz += (double)o[0] / ((double)(o[2] - o[1]) * 1e-2);
      }
      std::fprintf(stderr, "[ahip fused clk] %d workgroups: kernel span %.3f ms, workgroup duration min/avg/max %.3f/%.3f/%.3f ms, latest start +%.3f ms, %.0f MHz\n",
                   grid, (double)(t1 - t0) * 1e-5, dmin, dsum / grid, dmax, smax, mhz / grid);
      const char *ce = std::getenv("AHIP_FUSED_CLK");
      if (ce && ce[0] == '2')
        for (int b = 0; b < grid; ++b) {
          const long long *o = &hp[PH_N + 4 * b];
          const unsigned hw = (unsigned)o[3], xcc = (unsigned)(o[3] >> 32);
          std::fprintf(stderr, "[ahip fused wg] %d start %.3f end %.3f xcc %u se %u sh %u cu %u\n", b, (double)(o[1] - t0) * 1e-5, (double)(o[2] - t0) * 1e-5,
                       xcc & 15, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15);
        }
    }
  }
  return true;
}

void fused_free(Model &m) {
  if (!m.fused_state) return;
  FusedState *st = (FusedState *)m.fused_state;
  for (DevBuf *b : {&st->wbuf, &st->scratch, &st->seg_count, &st->seg_base, &st->tile_a0, &st->tile_e0, &st->centre, &st->ntiles, &st->partial, &st->prof, &st->dbg}) b->release();
  delete st;
  m.fused_state = nullptr;
}

// ---------------------------------------------------------------------------- diagnostic
// Two-wave self-test of the streamed register-chain linear: out[32][N] = in[32][K] @ W[K][N].
template <int KT, int NT>
__global__ void __launch_bounds__(128) k_selftest_linear(const float *Wf, int wbytes, const float *in, int K, float *out, int N) {
  const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4, row = (threadIdx.x >> 6) * 16 + j;
  f32x4 a[KT], o[NT];
#pragma unroll
  for (int t = 0; t < KT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int k = feat16(t, r, g);
      a[t][r] = k < K ? in[row * K + k] : 0.f;
    }
  __amdgpu_buffer_rsrc_t WB = __builtin_amdgcn_make_buffer_rsrc((void *)Wf, 0, wbytes, 0x00020000);
  f32x4 ring[RING];
  int wp = 0;
  ring_prime(WB, wp, lane * 16, ring);
  linear_s<KT, NT, false, 0>(WB, wp, a, o, lane * 16, ring, EpiNone{});
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int n = feat16(t, r, g);
      if (n < N) out[row * N + n] = o[t][r];
    }
}

template <int KS, int NT, int NTERM>
__global__ void __launch_bounds__(128) k_selftest_linear_b(const float *Wf, int wbytes, const float *in, int K, float *out, int N) {
  const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4, row = (threadIdx.x >> 6) * 16 + j;
  f32x4 a[2 * KS], o[NT];
#pragma unroll
  for (int t = 0; t < 2 * KS; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int k = feat16(t, r, g);
      a[t][r] = k < K ? in[row * K + k] : 0.f;
    }
  Bop b[KS], ob[NT / 2];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) b[ks] = split_pair(a[2 * ks], a[2 * ks + 1]);
  __amdgpu_buffer_rsrc_t WB = __builtin_amdgcn_make_buffer_rsrc((void *)Wf, 0, wbytes, 0x00020000);
  u32x4 ring[RingB<NTERM>::N];
  int wp = 0;
  ring_prime_b(WB, wp, lane * 16, ring);
  linear_b<KS, NT, false, false, 0, EpiNone, NTERM>(WB, wp, b, o, ob, lane * 16, ring, EpiNone{});
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int n = feat16(t, r, g);
      if (n < N) out[row * N + n] = o[t][r];
    }
}

template <int KS, int NT>
__global__ void __launch_bounds__(128) k_selftest_linear_h(const float *Wf, int wbytes, const float *in, int K, float *out, int N) {
  const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4, row = (threadIdx.x >> 6) * 16 + j;
  f32x4 a[2 * KS], o[1][NT];
#pragma unroll
  for (int t = 0; t < 2 * KS; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int k = feat16(t, r, g);
      a[t][r] = k < K ? in[row * K + k] : 0.f;
    }
  Hop b[1][KS], ob[1][NT / 2];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) b[0][ks] = split_pair_h(a[2 * ks], a[2 * ks + 1]);
  __amdgpu_buffer_rsrc_t WB = __builtin_amdgcn_make_buffer_rsrc((void *)Wf, 0, wbytes, 0x00020000);
  u32x4 ring[RINGH];
  int wp = 0;
  ring_prime_b(WB, wp, lane * 16, ring);
  EpiNone ep[1];
  linear_h<1, KS, NT, false, false, 0, EpiNone>(WB, wp, b, o, ob, lane * 16, ring, ep);
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int n = feat16(t, r, g);
      if (n < N) out[row * N + n] = o[0][t][r];
    }
}

#endif   // AHIP_FUSED_PART == 0
}  // namespace ahip

#if AHIP_FUSED_PART == 0
using namespace ahip;

// Diagnostic (AHIP_FUSED_DBG=1): per-edge {g[3], dd, dfc, dY[3]} of the last fused compute.
extern "C" int ahip_debug_fused_edges(ahip_model *mh, float *out, long long nedges) {
  ahip::Model *m = (ahip::Model *)mh;
  if (!m || !m->fused_state) return AHIP_ERR_STATE;
  FusedState &st = *(FusedState *)m->fused_state;
  if (!st.dbg_on || !st.dbg.p || nedges != m->nedges) return AHIP_ERR_STATE;
  if (hipDeviceSynchronize() != hipSuccess) return AHIP_ERR_DEVICE;
  try { copy_d2h(out, st.dbg.p, (size_t)nedges * 8 * sizeof(float)); } catch (...) { return AHIP_ERR_DEVICE; }
  return 0;
}

extern "C" int ahip_debug_fused_linear(int K, int N, const double *W, const float *in, float *out) {
  try {
    const char *ar = std::getenv("AHIP_FUSED_ARITH");
    const int nterm = (ar && std::string(ar) == "tf32eq") ? 2 : 3;
    const bool h2 = ar && std::string(ar) == "f16x2";
    const bool b3 = h2 || (ar && (std::string(ar) == "b3" || std::string(ar) == "bf16x3" || std::string(ar) == "tf32eq"));
    std::vector<float> frag;
    int KT, NT;
    if (h2) { append_frag_h(frag, W, K, N, N); frag_dims_b(K, N, KT, NT); }
    else if (b3) { append_frag_b(frag, W, K, N, N, nterm); frag_dims_b(K, N, KT, NT); }
    else { append_frag(frag, W, K, N, N); frag_dims(K, N, KT, NT); }
    frag.resize(frag.size() + (size_t)(RINGB + 2) * 256, 0.f);      // the ring prefetches past the end
    float *dW = nullptr, *din = nullptr, *dout = nullptr;
    AHIP_CHECK(hipMalloc((void **)&dW, frag.size() * sizeof(float)));
    AHIP_CHECK(hipMalloc((void **)&din, (size_t)32 * K * sizeof(float)));
    AHIP_CHECK(hipMalloc((void **)&dout, (size_t)32 * N * sizeof(float)));
    AHIP_CHECK(hipMemcpy(dW, frag.data(), frag.size() * sizeof(float), hipMemcpyHostToDevice));
    AHIP_CHECK(hipMemcpy(din, in, (size_t)32 * K * sizeof(float), hipMemcpyHostToDevice));
    bool ok = true;
    const int wbytes = (int)(frag.size() * sizeof(float));
#define CASEB(ks, nt) do { if (h2) hipLaunchKernelGGL((k_selftest_linear_h<ks, nt>), dim3(1), dim3(128), 0, 0, dW, wbytes, din, K, dout, N); \
                           else if (nterm == 3) hipLaunchKernelGGL((k_selftest_linear_b<ks, nt, 3>), dim3(1), dim3(128), 0, 0, dW, wbytes, din, K, dout, N); \
                           else hipLaunchKernelGGL((k_selftest_linear_b<ks, nt, 2>), dim3(1), dim3(128), 0, 0, dW, wbytes, din, K, dout, N); } while (0)
    if (b3) {
      if (KT == 1 && NT == 2) CASEB(1, 2);
      else if (KT == 1 && NT == 4) CASEB(1, 4);
      else if (KT == 2 && NT == 2) CASEB(2, 2);
      else if (KT == 2 && NT == 4) CASEB(2, 4);
      else if (KT == 3 && NT == 4) CASEB(3, 4);
      else if (KT == 2 && NT == 6) CASEB(2, 6);
      else ok = false;
    } else
#undef CASEB_
#define CASE(kt, nt) hipLaunchKernelGGL((k_selftest_linear<kt, nt>), dim3(1), dim3(128), 0, 0, dW, wbytes, din, K, dout, N)
    if (KT == 2 && NT == 2) CASE(2, 2);
    else if (KT == 2 && NT == 4) CASE(2, 4);
    else if (KT == 4 && NT == 2) CASE(4, 2);
    else if (KT == 4 && NT == 4) CASE(4, 4);
    else if (KT == 6 && NT == 4) CASE(6, 4);
    else if (KT == 4 && NT == 6) CASE(4, 6);
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
#endif   // AHIP_FUSED_PART == 0
