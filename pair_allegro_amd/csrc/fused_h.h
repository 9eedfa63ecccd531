// f16x2 arithmetic of the fused kernels' streamed linears (round 5).
//
// Every float32 operand v is represented by TWO float16 terms, both rounded to nearest even:
//     hi = f16(v),   lo' = f16((v - f16(v)) * 2^11)                    (v - hi is exact in float32)
// so that v = hi + 2^-11 lo' up to 2^-22 |v| (11 + 11 significant bits and the sign of the remainder; float32 itself: 2^-24), for every v whose hi
// term is a normal float16, |v| in [2^-14, 65504]: the 2^11 keeps the remainder out of the float16 subnormals.  A product w x is evaluated on the
// f16 matrix cores as three exact products with float32 accumulation,
//     w x  ~  w_hi x_hi  +  2^-11 (w_hi x_lo' + w_lo' x_hi)                (dropped: w_lo x_lo <= 2^-22 |w x|)
// the cross terms in an accumulator of their own that is folded in once per output tile (one fma per output value).  Against the three-term bf16 split
// (fused.hip: linear_b, six products per K-step) this is half the MFMAs, two thirds of the weight-stream bytes (those of the f32 stream) and 3 instead of
// 5.5 VALU operations per split value -- v_cvt_pk_f16_f32 converts two values per instruction -- at the same measured force error (DESIGN 4.2, round 5:
// max|dF| vs the float64 oracle 2.9e-6 against 2.8e-6 for float32 fmaf chains; tests/test_arith_emulation.py emulates all three on the CPU).
// What float16 does NOT have is bf16's exponent range: operands must stay inside [2^-14, 65504] to keep their precision.  Forward activations of a
// sane model do (O(1e-3 .. 1e2)); the backward pass is linear in the upstream gradient, so k_fused runs it scaled by a power of two chosen from the
// centre type's energy scale (derived in the kernels with frexp since round 6, undone on the edge gradient); a non-finite edge gradient raises FusedArgs::err (host: under
// fused_arith=auto the model switches to the f32 instance, engine.h; with an explicit f16x2: StateError).
//
// linear_h<G, ...>: G = 1 or 2 edge groups (16 slots each) of the wave share every weight fragment (two B operands, two accumulator sets per
// fragment): the fragment stream through the 64 B/clk/CU register-return path per edge halves with G = 2.
#pragma once
#include <cstring>

#include "fused_common.h"

namespace ahip {

typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
struct Hop { u32x4 hi, lo; };                  // B operand of one K-step (two 16-feature tiles): 8 f16 per lane and term
static constexpr float H_LO_SCALE = 2048.f, H_LO_INV = 1.f / 2048.f;
static constexpr int RINGH = 8;                // fragments in flight: two steps of hi0 hi1 lo0 lo1

__device__ __forceinline__ f32x4 mfma_h(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// two f32 -> packed f16 terms (low half = first value)
__device__ __forceinline__ void split2_h(float v0, float v1, unsigned &hi, unsigned &lo) {
  const f32x2 v = {v0, v1};
  const f16x2 h = __builtin_convertvector(v, f16x2);
  const f32x2 r = (v - __builtin_convertvector(h, f32x2)) * H_LO_SCALE;
  const f16x2 l = __builtin_convertvector(r, f16x2);
  hi = __builtin_bit_cast(unsigned, h);
  lo = __builtin_bit_cast(unsigned, l);
}
__device__ __forceinline__ Hop split_pair_h(const f32x4 &t0, const f32x4 &t1) {
  Hop b;
  unsigned h, l;
  split2_h(t0[0], t0[1], h, l); b.hi[0] = h; b.lo[0] = l;
  split2_h(t0[2], t0[3], h, l); b.hi[1] = h; b.lo[1] = l;
  split2_h(t1[0], t1[1], h, l); b.hi[2] = h; b.lo[2] = l;
  split2_h(t1[2], t1[3], h, l); b.hi[3] = h; b.lo[3] = l;
  return b;
}

// first RINGH fragments of the stream at wp into the ring
__device__ __forceinline__ void ring_prime_h(__amdgpu_buffer_rsrc_t W, int wp, int v16, u32x4 (&ring)[RINGH]) {
  int wo = wp * 4;
  pin_s(wo);
#pragma unroll
  for (int j = 0; j < RINGH; ++j) ring[j] = __builtin_bit_cast(u32x4, bload_w(W, v16 + (j & 3) * 1024, wo + (j >> 2) * 4096));
}

// Epilogue functors may process the two values of a register pair at once (apply2: packed f32 VALU operations); the others go value by value.
template <class E> __device__ __forceinline__ auto epi_apply2(E &e, int ot, int r, f32x2 v, int) -> decltype(e.apply2(ot, r, v)) { return e.apply2(ot, r, v); }
template <class E> __device__ __forceinline__ f32x2 epi_apply2(E &e, int ot, int r, f32x2 v, long) { return f32x2{e.apply(ot, r, v[0]), e.apply(ot, r + 1, v[1])}; }
// (a packed-f32 SiLU epilogue, EpiSiluSaveD2, was measured equal to the scalar one in round 5 -- 47.6 vs 47.7 ms -- and is gone)

// Streamed linear, f16x2.  in[g][ks]: KS K-steps (pairs of 16-feature tiles, split) per group; out[g][t]: NT f32 tiles through the epilogue functors of
// fused_common.h (one functor per group); SPLIT additionally emits the outputs as the next linear's B operands.  Fragments per (tile pair p, K-step):
// hi0 hi1 lo0 lo1, 1 KiB each, in consumption order, RINGH of them in flight.  The 8 G epilogue elements of pair p-1 are spread over the MFMAs of pair p.
template <int G, int KS, int NT, bool ACC, bool SPLIT, int RP, class Epi>
__device__ __forceinline__ void linear_h(__amdgpu_buffer_rsrc_t W, int &wp, const Hop (&in)[G][KS], f32x4 (&out)[G][NT], Hop (&outb)[G][NT / 2],
                                         int v16, u32x4 (&ring)[RINGH], Epi (&epi)[G]) {
  static_assert(NT % 2 == 0, "output tiles are processed in pairs");
  constexpr int RB = RINGH, NF = 4, NP = NT / 2, NSTEP = NP * KS, NS = NF * NSTEP;
  constexpr int MF = 6 * G;                        // MFMAs per step
  f32x4 ah[G][2], ac[G][2], prev[G][2];
  int wo = (wp + RB * 256) * 4;                    // byte offset of the next fragments to request: one running scalar (see linear_s)
  pin_s(wo);
#pragma unroll
  for (int s = 0; s < NSTEP; ++s) {
    const int p = s / KS, ks = s % KS;
    if (ks == 0) {
#pragma unroll
      for (int g = 0; g < G; ++g)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          ah[g][hh] = ACC ? out[g][2 * p + hh] : f32x4{0.f, 0.f, 0.f, 0.f};
          ac[g][hh] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    u32x4 a[NF];
#pragma unroll
    for (int i = 0; i < NF; ++i) {
      a[i] = ring[(RP + NF * s + i) % RB];
      ring[(RP + NF * s + i) % RB] = __builtin_bit_cast(u32x4, bload_w(W, v16 + i * 1024, wo));      // + i KiB: the instruction's immediate offset
    }
    wo += NF * 1024;
    pin_s(wo);
    // products: (w_lo', x_hi) -> ac, (w_hi, x_lo') -> ac, (w_hi, x_hi) -> ah; the accumulators of the two tiles and of the groups alternate
#pragma unroll
    for (int m = 0; m < 3; ++m) {
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
          if (m == 0) ac[g][hh] = mfma_h(a[2 + hh], in[g][ks].hi, ac[g][hh]);
          else if (m == 1) ac[g][hh] = mfma_h(a[hh], in[g][ks].lo, ac[g][hh]);
          else ah[g][hh] = mfma_h(a[hh], in[g][ks].hi, ah[g][hh]);
          if (p > 0) {
            // the 4 G register pairs of the previous tile pair, spread over this pair's MFMAs
            const int idx = ks * MF + (m * 2 + hh) * G + g, tot = KS * MF, NE = 4 * G;
            const int e0 = (idx * NE + tot - 1) / tot, e1 = ((idx + 1) * NE + tot - 1) / tot;
#pragma unroll
            for (int e = 0; e < NE; ++e)
              if (e >= e0 && e < e1) {
                const int eg = e / 4, th = (e % 4) / 2, r = 2 * (e % 2), ot = 2 * (p - 1) + th;
                const f32x2 y = epi_apply2(epi[eg], ot, r, f32x2{prev[eg][th][r], prev[eg][th][r + 1]}, 0);
                out[eg][ot][r] = y[0]; out[eg][ot][r + 1] = y[1];
                if (e % 4 == 3) {
                  epi[eg].flush(2 * (p - 1));
                  if (SPLIT) outb[eg][p - 1] = split_pair_h(out[eg][2 * (p - 1)], out[eg][2 * (p - 1) + 1]);
                }
              }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    if (ks == KS - 1) {
#pragma unroll
      for (int g = 0; g < G; ++g) {
        // fold the cross terms in: one fma per output value
        const f32x4 r0 = ac[g][0] * H_LO_INV + ah[g][0], r1 = ac[g][1] * H_LO_INV + ah[g][1];
        epi[g].tile_done(2 * p, r0);
        epi[g].tile_done(2 * p + 1, r1);
        if (p == NP - 1) {
#pragma unroll
          for (int r = 0; r < 4; r += 2) {
            const f32x2 y0 = epi_apply2(epi[g], 2 * p, r, f32x2{r0[r], r0[r + 1]}, 0), y1 = epi_apply2(epi[g], 2 * p + 1, r, f32x2{r1[r], r1[r + 1]}, 0);
            out[g][2 * p][r] = y0[0]; out[g][2 * p][r + 1] = y0[1]; out[g][2 * p + 1][r] = y1[0]; out[g][2 * p + 1][r + 1] = y1[1];
          }
          epi[g].flush(2 * p);
          if (SPLIT) outb[g][p] = split_pair_h(out[g][2 * p], out[g][2 * p + 1]);
        } else { prev[g][0] = r0; prev[g][1] = r1; }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  wp += NS * 256;
  pin_s(wp);
}

// ---- host side ----
// K-steps (pairs of 16-feature input tiles) and output tiles (padded to even) of a [K][N] linear on the 16x16x32 matrix instructions
static void frag_dims_b(int K, int N, int &KS, int &NT) {
  KS = (K + 31) / 32;
  NT = (N + 15) / 16;
  NT += NT & 1;
}
// f16x2 fragments of W [K][N] (fused_h.h): per (tile pair p, K-step ks) four 1 KiB entries hi0 hi1 lo0 lo1 in append_frag_b's lane layout;
// hi = f16(w), lo' = f16((w - hi) * 2^11), both round-to-nearest-even.  Returns range findings (engine.h): H_RANGE_OVERFLOW when a weight exceeds float16's
// range, H_RANGE_TINY when the whole matrix sits below 2^-10 (hi terms near or inside float16's subnormals: the split keeps an absolute 2^-36, i.e. fewer
// than 26 relative bits of such a matrix).
static int append_frag_h(std::vector<float> &out, const double *W, int K, int N, int ldw) {
  int KS, NT;
  frag_dims_b(K, N, KS, NT);
  bool ok = true;
  double wmax = 0.0;
  for (int k = 0; k < K; ++k)
    for (int n = 0; n < N; ++n) wmax = std::max(wmax, std::fabs(W[(size_t)k * ldw + n]));
  for (int p = 0; p < NT / 2; ++p)
    for (int ks = 0; ks < KS; ++ks)
      for (int term = 0; term < 2; ++term)
        for (int half = 0; half < 2; ++half)
          for (int lane = 0; lane < 64; ++lane)
            for (int w = 0; w < 4; ++w) {
              unsigned word = 0;
              for (int e = 0; e < 2; ++e) {
                const int sl = 2 * w + e, g = lane >> 4;
                const int k = sl < 4 ? 16 * (2 * ks) + 4 * g + sl : 16 * (2 * ks + 1) + 4 * g + (sl - 4);
                const int n = 16 * (2 * p + half) + (lane & 15);
                const float v = (k < K && n < N) ? (float)W[(size_t)k * ldw + n] : 0.f;
                if (!(std::fabs(v) < 32768.f)) ok = false;
                const _Float16 hi = (_Float16)v;
                const _Float16 lo = (_Float16)((v - (float)hi) * H_LO_SCALE);
                const _Float16 t = term == 0 ? hi : lo;
                unsigned short bits;
                std::memcpy(&bits, &t, 2);
                word |= (unsigned)bits << (16 * e);
              }
              float f;
              std::memcpy(&f, &word, 4);
              out.push_back(f);
            }
  return (ok ? 0 : H_RANGE_OVERFLOW) | ((wmax > 0.0 && wmax < 0x1p-10) ? H_RANGE_TINY : 0);
}
}  // namespace ahip
