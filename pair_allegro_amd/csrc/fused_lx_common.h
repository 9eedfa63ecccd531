// Pieces shared by the wide fused kernels (fused_lx.hip: one wave per SIMD holding the whole edge tensor; fused_lx2.hip: wave pairs,
// each wave holding half of its channels): kernel arguments, Clebsch-Gordan table access, half-row tensor product and its
// gradient, the AGPR park of the edge tensor, small helpers.
#pragma once
#include "cg_tables.h"
#include "engine.h"
#include "fused_common.h"
#include "fused_h.h"
#include "prims.h"
#include <algorithm>

namespace ahip {

static constexpr int LX_MAXNL = 3;

template <int L> struct CgX;
template <> struct CgX<1> { static constexpr const AhipCgEntry *tab = ahip_cg_l1; static constexpr int N = AHIP_CG_L1_N, NS = AHIP_CG_L1_NSCALAR, NP = AHIP_CG_L1_NPATHS, NPS = AHIP_CG_L1_NPATHS_SCALAR; };
template <> struct CgX<2> { static constexpr const AhipCgEntry *tab = ahip_cg_l2; static constexpr int N = AHIP_CG_L2_N, NS = AHIP_CG_L2_NSCALAR, NP = AHIP_CG_L2_NPATHS, NPS = AHIP_CG_L2_NPATHS_SCALAR; };

__host__ __device__ constexpr int l_of_lm(int lm) { return lm == 0 ? 0 : (lm < 4 ? 1 : 2); }

struct FusedLxArgs {
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
  int tchunk;                    // tiles per claim of the dynamic schedule (1 for small systems: a workgroup's last claim sets the makespan)
  const int *tile_a0, *tile_e0, *ntiles;
  // weights (offsets in floats into wbase)
  const float *wbase;
  int wbytes;
  int o_stream, o_stream_hi;     // o_stream_hi: the second wave half's stream (fused_lx2.hip)
  int o_tbtab, tb_nk, o_tpl, o_out1, o_scale, o_shift;
  int o_res[LX_MAXNL];
  // scratch
  float *scratch;
  long long wg_scratch, wave_scratch;     // floats
  // outputs
  double *f, *eatom, *partial;            // partial [gridDim.x][7]
  long long *prof;
  float cp[6];                            // cutoff polynomial coefficients (fused_common.h: cutoff_poly_c): wave-uniform kernel arguments instead of per-lane values held over a tile
  int *err;                               // host-mapped word: set when an edge gradient comes out non-finite (float16 range exceeded)
};

// Arithmetic of the wide kernels' streamed linears: AR = 0 the f32-input MFMA (linear_s), AR = 3 f16x2 (fused_h.h: linear_h, one edge group).  Both
// consume the same number of 1 KiB fragments per linear (a multiple of the ring depth, except the 32-feature mixing rows of fused_lx.hip: half a ring, alternating phase RP).
template <int AR> struct LxRing {
  f32x4 f[AR == 0 ? RING : 1];
  u32x4 h[AR == 3 ? RINGH : 1];
};
template <int AR> __device__ __forceinline__ void lx_prime(__amdgpu_buffer_rsrc_t W, int wp, int v16, LxRing<AR> &ring) {
  if constexpr (AR == 3) ring_prime_h(W, wp, v16, ring.h);
  else ring_prime(W, wp, v16, ring.f);
}
template <int AR, int KT, int NT, bool ACC, int RP = 0, class Epi>
__device__ __forceinline__ void lx_lin(__amdgpu_buffer_rsrc_t W, int &wp, const f32x4 (&in)[KT], f32x4 (&out)[NT], int v16, LxRing<AR> &ring, Epi epi) {
  if constexpr (AR == 3) {
    static_assert(KT % 2 == 0 && RINGH == RING, "K-steps are pairs of 16-feature tiles; one wrap-around copy serves both arithmetics");
    Hop b[1][KT / 2], unused[1][NT / 2];
#pragma unroll
    for (int ks = 0; ks < KT / 2; ++ks) b[0][ks] = split_pair_h(in[2 * ks], in[2 * ks + 1]);
    Epi ep[1] = {epi};
    linear_h<1, KT / 2, NT, ACC, false, RP, Epi>(W, wp, b, reinterpret_cast<f32x4 (&)[1][NT]>(out), unused, v16, ring.h, ep);
  } else linear_s<KT, NT, ACC, RP, Epi>(W, wp, in, out, v16, ring.f, epi);
}

// host-side state of one wide fused kernel family
struct FusedLxState {
  DevBuf wbuf, scratch, seg_count, seg_base, tile_a0, tile_e0, centre, ntiles, partial, prof;
  FusedLxArgs args;
  bool ready = false, prof_on = false;
  int ncu = 256;
  int L = 0, UT = 0;
  int arith = 0;               // 0: f32-input MFMA, 3: f16x2 (lx_arith_of)
};

// edge total for the claim-size heuristic: the value itself when it is on the host, else the last one that was, else from the list's size
static long long lx_nedges_estimate(const Model &m) {
  return !m.counts_pending ? m.nedges : m.nedges_hint > 0 ? m.nedges_hint : (long long)(0.58 * (double)m.nneigh);
}
// Tile packing shared by the wide fused kernels: consecutive centres into tiles of <= slots edges and <= maxa centres
// (k_pack_tiles, fused_common.h), per-centre {atom, type}, per-edge packed types, first edge of every tile.
// the tile arrays the kernel reads: the edge build's (it packed them itself, edges.hip) or this state's
static bool lx_prepacked(const Model &m, int slots, int maxa) { return m.tiles_packed && m.pack_slots == slots && m.pack_maxa == maxa; }
static void lx_tile_args(Model &m, FusedLxState &st, FusedLxArgs &A, int slots, int maxa) {
  const bool pre = lx_prepacked(m, slots, maxa);
  int *const ntl = pre ? m.b_ntiles.as<int>() : st.ntiles.as<int>();
  A.centre = pre ? m.b_centre.as<int2>() : st.centre.as<int2>();
  A.tile_a0 = pre ? m.b_tile_a0.as<int>() : st.tile_a0.as<int>(); A.tile_e0 = pre ? m.b_tile_e0.as<int>() : st.tile_e0.as<int>(); A.ntiles = ntl;
  A.tile_counter = (unsigned int *)(ntl + 1);
  m.d_ntiles_last = ntl; m.last_tile_slots = slots;
}
static void lx_pack_tiles(Model &m, FusedLxState &st, const ComputeArgs &a, int slots, int maxa) {
  if (lx_prepacked(m, slots, maxa)) return;
  hipStream_t s = a.stream;
  const int inum = m.inum;
  const int nseg = (inum + SEG - 1) / SEG;
  st.seg_count.reserve((size_t)(nseg + 1) * sizeof(int));
  st.seg_base.reserve((size_t)(nseg + 2) * sizeof(int));
  st.tile_a0.reserve((size_t)(inum + nseg + 2) * sizeof(int));
  st.tile_e0.reserve((size_t)(inum + nseg + 2) * sizeof(int));
  StageTimer tm(m, "tile_pack", s);
  const unsigned B = 64;
  st.centre.reserve((size_t)std::max(inum, 1) * sizeof(int2));
  const bool small = inum <= PACK_SMALL_ATOMS;
  if (small)
    hipLaunchKernelGGL(k_pack_small, dim3(1), dim3(PACK_SMALL_SEGS), 0, s, inum, m.b_eoff.as<int>(), nseg, st.tile_a0.as<int>(), st.tile_e0.as<int>(), st.ntiles.as<int>(), slots, maxa,
                       m.d_ilist, a.mtype, st.centre.as<int2>());
  else {
    hipLaunchKernelGGL(k_pack_tiles<false>, dim3((nseg + B - 1) / B), dim3(B), 0, s, inum, m.b_eoff.as<int>(), nseg, st.seg_count.as<int>(), (const int *)nullptr, (int *)nullptr, slots, maxa);
    AHIP_CHECK(prim_exclusive_scan_i32(m.prim, st.seg_count.as<int>(), st.seg_base.as<int>(), nseg, s));
    hipLaunchKernelGGL(k_pack_tiles<true>, dim3((nseg + B - 1) / B), dim3(B), 0, s, inum, m.b_eoff.as<int>(), nseg, (int *)nullptr, st.seg_base.as<int>(), st.tile_a0.as<int>(), slots, maxa);
    hipLaunchKernelGGL(k_pack_finish, dim3(1), dim3(1), 0, s, inum, nseg, st.seg_base.as<int>(), st.tile_a0.as<int>(), st.ntiles.as<int>());
    hipLaunchKernelGGL(k_centre_info, dim3((inum + 255) / 256), dim3(256), 0, s, inum, m.d_ilist, a.mtype, st.centre.as<int2>());
  }
  if (!m.have_ett) {
    m.b_ett.reserve((size_t)std::max<long long>(m.nedges, 1));
    hipLaunchKernelGGL(k_edge_types, dim3((unsigned)((m.nedges + 255) / 256)), dim3(256), 0, s, m.nedges, m.b_eii.as<int>(), m.b_ej.as<int>(), m.d_ilist, a.mtype, m.b_ett.as<unsigned char>());
    m.have_ett = true;
  }
  if (!small) {
    const int tcap = inum + nseg + 1;
    hipLaunchKernelGGL(k_tile_e0, dim3((tcap + 255) / 256), dim3(256), 0, s, st.ntiles.as<int>(), st.tile_a0.as<int>(), m.b_eoff.as<int>(), st.tile_e0.as<int>());
  }
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
// scalar f32 helpers on 2-feature half rows: component indices written out (v_fmac_f32 with the CG constant as an inline
// literal; packed f32 would hold every constant in an SGPR pair and the kernel runs out of SGPRs)
#ifdef AHIP_LX_SCALAR_TP
__device__ __forceinline__ f32x2 fma_cab(float c, const f32x2 &a, const f32x2 &b, const f32x2 &acc) {      // acc + c * (a * b)
  f32x2 r;
  r[0] = fmaf(c, a[0] * b[0], acc[0]); r[1] = fmaf(c, a[1] * b[1], acc[1]);
  return r;
}
__device__ __forceinline__ f32x2 fma_cgo(float c, const f32x2 &gq, const f32x2 &o, const f32x2 &acc) {    // acc + (c * g) * o
  f32x2 r;
  r[0] = fmaf(c * gq[0], o[0], acc[0]); r[1] = fmaf(c * gq[1], o[1], acc[1]);
  return r;
}
__device__ __forceinline__ f32x2 fma_rows(const f32x2 &a, const f32x2 &b, const f32x2 &acc) {              // acc + a * b
  f32x2 r;
  r[0] = fmaf(a[0], b[0], acc[0]); r[1] = fmaf(a[1], b[1], acc[1]);
  return r;
}
__device__ __forceinline__ f32x2 mul_rows(const f32x2 &a, const f32x2 &b) {
  f32x2 r;
  r[0] = a[0] * b[0]; r[1] = a[1] * b[1];
  return r;
}
#else
// packed f32 (v_pk_mul_f32 / v_pk_fma_f32): a lone wave on a SIMD issues one instruction per 4 cycles whatever it is, so the
// two-feature packed forms halve the tensor product's issue time
__device__ __forceinline__ f32x2 fma_cab(float c, const f32x2 &a, const f32x2 &b, const f32x2 &acc) { return acc + c * (a * b); }
__device__ __forceinline__ f32x2 fma_cgo(float c, const f32x2 &gq, const f32x2 &o, const f32x2 &acc) { return acc + (c * gq) * o; }
__device__ __forceinline__ f32x2 fma_rows(const f32x2 &a, const f32x2 &b, const f32x2 &acc) { return acc + a * b; }
__device__ __forceinline__ f32x2 mul_rows(const f32x2 &a, const f32x2 &b) { return a * b; }
#endif

// ---------------------------------------------------------------------------- tensor product, CG table unrolled
// The tensor product is element-wise over features, so it is evaluated on HALF rows (2 of the lane's 4 features of a K-tile)
// at a time: the row sets it keeps live (inputs, environment, outputs) are half as large, which is what lets the whole edge
// tensor stay in registers next to them.
// out[i3] += pw[path] * c * v[i1] * e[i2] over the table entries; every index is a compile-time constant.  The table is sorted
// by path: the entries of one path accumulate (CG constants as literals) into at most 2 l3 + 1 partial rows, which are scaled
// by the path weight once: two VALU operations per entry and feature.
template <int L, bool SCALAR, int U>
__device__ __forceinline__ void tp_fwd_x(const f32x2 (&v)[(L + 1) * (L + 1)], const float *en, const float *tp,
                                         f32x2 (&out)[SCALAR ? 1 : (L + 1) * (L + 1)]) {
  constexpr int D = (L + 1) * (L + 1), DOUT = SCALAR ? 1 : D, N = SCALAR ? CgX<L>::NS : CgX<L>::N;
  f32x2 ee[D];
#pragma unroll
  for (int k = 0; k < D; ++k) ee[k] = *(const f32x2 *)(en + k * U);
#pragma unroll
  for (int k = 0; k < DOUT; ++k) out[k] = f32x2{0.f, 0.f};
  f32x2 acc[2 * L + 1];
#pragma unroll
  for (int q = 0; q < N; ++q) {
    constexpr const AhipCgEntry *tab = CgX<L>::tab;
    const int p = tab[q].path, l3 = l_of_lm(tab[q].i3), b3 = l3 * l3;
    if (q == 0 || tab[q - 1].path != p) {
#pragma unroll
      for (int k = 0; k < 2 * L + 1; ++k) acc[k] = f32x2{0.f, 0.f};
    }
    acc[tab[q].i3 - b3] = fma_cab((float)tab[q].c, v[tab[q].i1], ee[tab[q].i2], acc[tab[q].i3 - b3]);
    if (q == N - 1 || tab[q + 1].path != p) {
      const f32x2 pw = *(const f32x2 *)(tp + p * U);
#pragma unroll
      for (int k = 0; k < 2 * L + 1; ++k)
        if (k < 2 * l3 + 1) out[b3 + k] = fma_rows(pw, acc[k], out[b3 + k]);
      __builtin_amdgcn_sched_barrier(0);       // one path at a time: bounds the live products
    }
  }
}
// Gradient of the tensor product, as two passes over the table so that each pass keeps three row sets live instead of five:
//   which = 0:  r[i1] += pw c g[i3] o[i2]   with o = environment rows (from LDS)     -> gradient w.r.t. the edge tensor
//   which = 1:  r[i2] += pw c g[i3] o[i1]   with o = edge tensor rows (registers)   -> per-edge environment gradient
// per path the output-gradient rows are scaled by the path weight once; two VALU operations per entry and feature.
template <int L, bool SCALAR, int U, int WHICH>
__device__ __forceinline__ void tp_bwd_half(const f32x2 (&o)[(L + 1) * (L + 1)], const f32x2 (&pw)[CgX<L>::NP],
                                            const f32x2 (&g)[SCALAR ? 1 : (L + 1) * (L + 1)], f32x2 (&r)[(L + 1) * (L + 1)]) {
  constexpr int D = (L + 1) * (L + 1), N = SCALAR ? CgX<L>::NS : CgX<L>::N;
#pragma unroll
  for (int k = 0; k < D; ++k) r[k] = f32x2{0.f, 0.f};
  f32x2 gp[2 * L + 1];
#pragma unroll
  for (int q = 0; q < N; ++q) {
    constexpr const AhipCgEntry *tab = CgX<L>::tab;
    const int p = tab[q].path, l3 = l_of_lm(tab[q].i3), b3 = l3 * l3;
    if (q == 0 || tab[q - 1].path != p) {
#pragma unroll
      for (int k = 0; k < 2 * L + 1; ++k)
        if (k < 2 * l3 + 1) gp[k] = mul_rows(pw[p], g[SCALAR ? 0 : b3 + k]);
    }
    if (WHICH == 0) r[tab[q].i1] = fma_cgo((float)tab[q].c, gp[tab[q].i3 - b3], o[tab[q].i2], r[tab[q].i1]);
    else r[tab[q].i2] = fma_cgo((float)tab[q].c, gp[tab[q].i3 - b3], o[tab[q].i1], r[tab[q].i2]);
    if (q == N - 1 || tab[q + 1].path != p) __builtin_amdgcn_sched_barrier(0);
  }
}
// the same with the path weights read from LDS (tp + path * U) where a path starts, instead of handed in as a register array
template <int L, bool SCALAR, int U, int WHICH>
__device__ __forceinline__ void tp_bwd_half(const f32x2 (&o)[(L + 1) * (L + 1)], const float *tp,
                                            const f32x2 (&g)[SCALAR ? 1 : (L + 1) * (L + 1)], f32x2 (&r)[(L + 1) * (L + 1)]) {
  constexpr int D = (L + 1) * (L + 1), N = SCALAR ? CgX<L>::NS : CgX<L>::N;
#pragma unroll
  for (int k = 0; k < D; ++k) r[k] = f32x2{0.f, 0.f};
  f32x2 gp[2 * L + 1];
#pragma unroll
  for (int q = 0; q < N; ++q) {
    constexpr const AhipCgEntry *tab = CgX<L>::tab;
    const int p = tab[q].path, l3 = l_of_lm(tab[q].i3), b3 = l3 * l3;
    if (q == 0 || tab[q - 1].path != p) {
      const f32x2 pwp = *(const f32x2 *)(tp + p * U);
#pragma unroll
      for (int k = 0; k < 2 * L + 1; ++k)
        if (k < 2 * l3 + 1) gp[k] = mul_rows(pwp, g[SCALAR ? 0 : b3 + k]);
    }
    if (WHICH == 0) r[tab[q].i1] = fma_cgo((float)tab[q].c, gp[tab[q].i3 - b3], o[tab[q].i2], r[tab[q].i1]);
    else r[tab[q].i2] = fma_cgo((float)tab[q].c, gp[tab[q].i3 - b3], o[tab[q].i1], r[tab[q].i2]);
    if (q == N - 1 || tab[q + 1].path != p) __builtin_amdgcn_sched_barrier(0);
  }
}

// ---------------------------------------------------------------------------- tensor product, grouped tables (cg_tables.h: AhipCgG)
// O[out] = sum over the table of  pw'[path] * sign * scale * A[a] * B[b],  pw' = path weight x the path's most frequent |c| (folded by the
// host, ahip_cg_l?_cbase).  The A rows of a path are scaled by pw' once; an entry with that |c| is then ONE fma into O[out], the other
// entries are summed per (path, out, |c|) group and enter with one more fma: 47 + 137 + 27 packed operations per table pass for l_max = 2
// instead of 2 x 137 + 47.  VAR 0: forward (A = edge tensor rows i1, B = environment rows i2, O = output rows i3);
// VAR 1: gradient w.r.t. the edge tensor (A = output gradient i3, B = environment i2, O over i1); VAR 2: per-edge environment gradient
// (A = output gradient i3, B = edge tensor i1, O over i2).  SCALAR: only the paths with l3 = 0 (last layer).
template <int L, int VAR> struct CgG;
template <> struct CgG<1, 0> { static constexpr const AhipCgG *tab = ahip_cg_l1_f; };
template <> struct CgG<1, 1> { static constexpr const AhipCgG *tab = ahip_cg_l1_b0; };
template <> struct CgG<1, 2> { static constexpr const AhipCgG *tab = ahip_cg_l1_b1; };
template <> struct CgG<2, 0> { static constexpr const AhipCgG *tab = ahip_cg_l2_f; };
template <> struct CgG<2, 1> { static constexpr const AhipCgG *tab = ahip_cg_l2_b0; };
template <> struct CgG<2, 2> { static constexpr const AhipCgG *tab = ahip_cg_l2_b1; };
template <int L, int VAR, bool SCALAR, int U, int NA, int NB, int NO>
__device__ __forceinline__ void tp_g(const f32x2 (&A)[NA], const f32x2 (&B)[NB], const float *tp, f32x2 (&O)[NO]) {
  constexpr int N = CgX<L>::N;
#pragma unroll
  for (int k = 0; k < NO; ++k) O[k] = f32x2{0.f, 0.f};
  f32x2 sa[2 * L + 1];
  f32x2 s = f32x2{0.f, 0.f};
  // the path weights are requested ONE PATH AHEAD (the tables run through the paths 0, 1, 2, ... in order): read where a path starts, every
  // path began with an LDS round trip that nothing covered -- 15 per table pass, 12 passes per layer
  f32x2 pwn = f32x2{0.f, 0.f};
  if (!SCALAR) pwn = *(const f32x2 *)(tp + CgG<L, VAR>::tab[0].path * U);
#pragma unroll
  for (int q = 0; q < N; ++q) {
    constexpr const AhipCgG *tab = CgG<L, VAR>::tab;
    const int l3 = VAR == 0 ? l_of_lm(tab[q].out) : l_of_lm(tab[q].a);
    if (!(SCALAR && l3 != 0)) {
      const int la = l_of_lm(tab[q].a), ba = la * la;
      if (q == 0 || tab[q - 1].path != tab[q].path) {
        f32x2 pwp;
        if (SCALAR) pwp = *(const f32x2 *)(tp + tab[q].path * U);
        else {
          pwp = pwn;
          if (tab[q].path + 1 < CgX<L>::NP) pwn = *(const f32x2 *)(tp + (tab[q].path + 1) * U);
        }
#pragma unroll
        for (int k = 0; k < 2 * L + 1; ++k)
          if (k < 2 * la + 1) sa[k] = pwp * A[ba + k];
      }
      const f32x2 av = tab[q].sign > 0 ? sa[tab[q].a - ba] : -sa[tab[q].a - ba];
      if (tab[q].kind == 0) O[tab[q].out] = O[tab[q].out] + av * B[tab[q].b];
      else {
        if (tab[q].kind & 2) s = av * B[tab[q].b]; else s = s + av * B[tab[q].b];
        if (tab[q].kind & 4) O[tab[q].out] = O[tab[q].out] + (float)tab[q].scale * s;
      }
      if (q == N - 1 || tab[q + 1].path != tab[q].path) __builtin_amdgcn_sched_barrier(0);       // one path at a time: bounds the live products
    }
  }
}

// The edge tensor lives in the ACCUMULATOR half of the unified register file (AGPRs): its 144 registers then do not compete
// with the arithmetic VGPRs in the register allocator.  A value is parked with v_accvgpr_write (inline asm: there is no
// builtin); reads are plain uses, the compiler inserts v_accvgpr_read.  No MFMA reads a parked value within the next
// instructions (every park is followed by VALU / memory work), so the asm needs no wait states of its own.
#ifdef AHIP_NO_ACC_PARK
// two waves per SIMD (fused_lx2.hip): 256 registers per wave either way, and a kernel that names no AGPR gets all of them as
// ordinary VGPRs (with AGPRs in use the compiler splits the file 128 / 128, whatever the kernel needs)
__device__ __forceinline__ float acc_park(float x) { asm volatile("" : "+v"(x)); return x; }      // pinned: the value is computed HERE (see pin())
#else
__device__ __forceinline__ float acc_park(float x) {
  float a;
  asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(a) : "v"(x));
  return a;
}
#endif
__device__ __forceinline__ void acc_put4(float (&dst)[4], const f32x4 &v) {
  dst[0] = acc_park(v[0]); dst[1] = acc_park(v[1]); dst[2] = acc_park(v[2]); dst[3] = acc_park(v[3]);
}
__device__ __forceinline__ void acc_put2(float (&dst)[4], int h, const f32x2 &v) {
  if (h == 0) { dst[0] = acc_park(v[0]); dst[1] = acc_park(v[1]); } else { dst[2] = acc_park(v[0]); dst[3] = acc_park(v[1]); }
}
__device__ __forceinline__ f32x4 acc_get4(const float (&src)[4]) { return f32x4{src[0], src[1], src[2], src[3]}; }
__device__ __forceinline__ f32x2 acc_get2(const float (&src)[4], int h) { return h == 0 ? f32x2{src[0], src[1]} : f32x2{src[2], src[3]}; }

__device__ __forceinline__ f32x2 half_of(const f32x4 &v, int h) { return h == 0 ? f32x2{v[0], v[1]} : f32x2{v[2], v[3]}; }
__device__ __forceinline__ void set_half(f32x4 &v, int h, const f32x2 &x) { if (h == 0) { v[0] = x[0]; v[1] = x[1]; } else { v[2] = x[0]; v[3] = x[1]; } }
// 8-byte half of a saved row image (row = [lane][4 floats])
__device__ __forceinline__ f32x2 bload_half(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
#ifdef ABL_NOROWS
  const float q = __builtin_bit_cast(float, (voff + soff) | 0x3f000000);
  return f32x2{q, q};
#else
  typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, AHIP_ROW_AUX));
#endif
}

// Pins a running per-edge sum where it is computed.  The sums over channels (dE/dY, the cutoff and distance derivatives, the edge energy) are
// only consumed at the end of the tile; without the pin the optimiser SINKS their whole accumulation chains down to that use, which keeps
// every operand row (LDS reads of the environment gradient, saved omega / w0 rows) alive until then -- in scratch: they were most of the
// kernel's spill traffic (18 + 18 + 21 sixteen-byte reloads in the finish phase).
__device__ __forceinline__ void pin(float &v) { asm volatile("" : "+v"(v)); }

__device__ __forceinline__ float hsum4(const f32x4 &v) { return (v[0] + v[1]) + (v[2] + v[3]); }

// The thread index is recomputed here from the hardware lane counter (and the wave index the caller keeps in an SGPR): a value
// derived from threadIdx at kernel entry is live across the whole tile and ends up in a scratch slot, whose reload at this point
// waits (vmcnt(0), loads return in order) for the saved rows the caller has just requested from HBM.
__device__ __forceinline__ int fresh_lane() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}

// Last-layer rows of k_fused_lx that live in LDS instead of scratch (round 6; register images: 16 B per lane, like the park rows of fused.hip)
__device__ __forceinline__ void lrow_store(float *base, int row, f32x4 v, int lane) { *(f32x4 *)(base + row * ROW + lane * 4) = v; }
__device__ __forceinline__ f32x4 lrow_load(const float *base, int row, int lane) { return *(const f32x4 *)(base + row * ROW + lane * 4); }
// omega rows (output tiles FIRST ..) of the last layer -> LDS rows 0 ..
template <int FIRST> struct EpiSaveFromL {
  static constexpr bool STORES = false;
  float *base; int lane;
  __device__ __forceinline__ void tile_done(int ot, const f32x4 &acc) const { if (ot >= FIRST) lrow_store(base, ot - FIRST, acc, lane); }
  __device__ __forceinline__ float apply(int, int, float v) const { return v; }
  __device__ __forceinline__ void flush(int) const {}
};
// the last layer's input tensor rows (first NMAX output tiles of a mixing row): row index row0 + ot goes to LDS row lrow0 + index while index < NLDS, else to scratch
template <int NMAX> struct EpiSaveNSplit {
  static constexpr bool STORES = true;
  __amdgpu_buffer_rsrc_t S; int srow0, v16;
  float *base; int idx0, lrow0, nlds, lane;
  __device__ __forceinline__ void tile_done(int ot, const f32x4 &acc) const {
    if (ot >= NMAX) return;
    if (idx0 + ot < nlds) lrow_store(base, lrow0 + idx0 + ot, acc, lane);
    else bstore(S, v16, (srow0 + idx0 + ot) * ROW * 4, acc);
  }
  __device__ __forceinline__ float apply(int, int, float v) const { return v; }
  __device__ __forceinline__ void flush(int) const {}
};
// Saves only the first NMAX output tiles of a linear (the rest is zero padding of a ring-aligned fragment block)
template <int NMAX> struct EpiSaveN {
  static constexpr bool STORES = true;
  __amdgpu_buffer_rsrc_t S; int row0, v16;
  __device__ __forceinline__ void tile_done(int ot, const f32x4 &acc) const { if (ot < NMAX) bstore(S, v16, (row0 + ot) * ROW * 4, acc); }
  __device__ __forceinline__ float apply(int, int, float v) const { return v; }
  __device__ __forceinline__ void flush(int) const {}
};

}  // namespace ahip
