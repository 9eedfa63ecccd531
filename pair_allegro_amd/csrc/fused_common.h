// Pieces shared by the fused MFMA kernels (fused.hip: model S, l_max = 1, 32 tensor features; fused_lx.hip: l_max <= 2,
// 32 or 64 tensor features): register-image rows, the streamed register-chain linear with its epilogue functors, the
// A-operand fragment layout of the weight stream, tile packing.  See fused.hip / DESIGN.md 4.2 for the mapping.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <string>
#include <vector>

#include "engine.h"

namespace ahip {

typedef float f32x4 __attribute__((ext_vector_type(4)));
static constexpr int SEG = 128;          // atoms per sequential packing segment (one thread walks a segment: the chain of 512 cost 0.08 ms per pass whatever the system size; a segment ends its last tile early: < 1 % more tiles)
static constexpr int ROW = 256;          // floats per saved register image of one 16-feature tile (4 regs x 64 lanes)
#ifndef AHIP_RING
#define AHIP_RING 8
#endif
static constexpr int RING = AHIP_RING;   // weight fragments in flight per wave
static constexpr int TCHUNK = 4;         // tiles per claim of the dynamic tile schedule

// Switches that exist for timing experiments only and compute WRONG results (or drop a hazard pad) are refused outside an experiment build.
#if (defined(ABL_NOROWS) || defined(ABL_NOW) || defined(AHIP_NO_STORE_PAD) || defined(ABL_NO_ENVSTAGE) || defined(ABL_NO_FTP) || defined(ABL_NO_LAT) || defined(ABL_NO_MIX) || \
     defined(ABL_NOSYNC) || defined(ABL_NOREDUCE) || defined(ABL_NOROWST) || defined(ABL_NOROWLD) || defined(ABL_NOATOM) || defined(ABL_NOTBGATHER)) && !defined(AHIP_EXPERIMENT_SWITCHES)
#error "ABL_* / AHIP_NO_STORE_PAD are timing-experiment switches (wrong results): add -DAHIP_EXPERIMENT_SWITCHES, never in the product build"
#endif

__host__ __device__ inline int feat16(int t, int r, int g) { return 16 * t + 4 * g + r; }

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// 16-byte buffer accesses: wave-uniform descriptor + scalar byte offset + per-lane byte offset
// cache policy of the saved-row traffic (aux bits of the buffer instructions; 2 = nt: stream through L2 without displacing the
// weight stream, which every CU re-reads); the including kernel may define it before this header
#ifndef AHIP_ROW_AUX
#define AHIP_ROW_AUX 0
#endif
// keeps a wave-uniform value in a scalar register and opaque to the optimiser at this point (no code motion / constant folding through it)
__device__ __forceinline__ void pin_s(int &v) { asm volatile("" : "+s"(v)); }
#ifdef ABL_NOROWS   // timing experiment only (results are wrong): no saved-row traffic
__device__ __forceinline__ f32x4 bload(__amdgpu_buffer_rsrc_t, int voff, int soff) { const float q = __builtin_bit_cast(float, (voff + soff) | 0x3f000000); return f32x4{q, q, q, q}; }
__device__ __forceinline__ void bstore(__amdgpu_buffer_rsrc_t, int, int, f32x4 v) { asm volatile("" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3])); }
#else
#ifdef ABL_NOROWLD   // timing experiment only (results are wrong): saved rows are stored but never loaded
__device__ __forceinline__ f32x4 bload(__amdgpu_buffer_rsrc_t, int voff, int soff) { const float q = __builtin_bit_cast(float, (voff + soff) | 0x3f000000); return f32x4{q, q, q, q}; }
#else
__device__ __forceinline__ f32x4 bload(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, AHIP_ROW_AUX));
}
#endif
// HAZARD (gfx950, found in round 4 by editing the assembly of a failing build one instruction class / region at a time, tools/asm_variant.sh):
// a buffer_store_dwordx4 reads its 16 bytes of data AFTER it has issued; a VALU or MFMA instruction that overwrites one of the four data registers
// in the next issue slot gets there first and the store writes the NEW value.  LLVM knows this hazard ("VMEM store of more than 8 bytes
// followed by a write of its data registers", GCNHazardRecognizer::createsVALUHazard) and pads it -- except for MUBUF stores whose soffset is
// an SGPR, which the ISA manuals exempt.  Every store here has its row offset in an SGPR, the saved rows die with the store, so the register
// allocator hands the registers to the next instruction whenever the schedule allows it: wrong rows for the backward pass, only when no
// other wave's instruction happens to be issued in between (forces off by a few per cent on a few edges, different ones on every launch;
// rounds 2-3: "accumulator stores must have completed", "the bf16 instances break under two compiler options").  The s_nop below READS the
// data registers (so nothing may overwrite them before it) and supplies the two wait states the hazard needs on gfx940+.
// tools/store_hazard.hip reproduces it in 40 lines.
#ifdef ABL_NOROWST    // timing experiment only (results are wrong): saved rows are loaded but never stored
__device__ __forceinline__ void bstore(__amdgpu_buffer_rsrc_t, int, int, f32x4 v) { asm volatile("" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3])); }
#else
__device__ __forceinline__ void bstore(__amdgpu_buffer_rsrc_t r, int voff, int soff, f32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, soff, AHIP_ROW_AUX);
#ifndef AHIP_NO_STORE_PAD          // experiment switch (tools/store_hazard.hip, A/B timing): builds the unsafe form
  asm volatile("s_nop 1" ::"v"(v));
#endif
}
#endif
#endif
#ifdef ABL_NOW       // timing experiment only (results are wrong): no weight-fragment traffic
__device__ __forceinline__ f32x4 bload_w(__amdgpu_buffer_rsrc_t, int voff, int soff) { const float q = __builtin_bit_cast(float, ((voff + soff) & 0xffff) | 0x3c000000); return f32x4{q, q, q, q}; }
#else
__device__ __forceinline__ f32x4 bload_w(__amdgpu_buffer_rsrc_t r, int voff, int soff) {        // weight fragments: default policy
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
#endif

__device__ __forceinline__ float sigmoidf_fast(float z) { return __builtin_amdgcn_rcpf(1.f + __expf(-z)); }
__device__ __forceinline__ float silu1(float z) { return z * sigmoidf_fast(z); }
__device__ __forceinline__ float dsilu1(float z) {
  float s = sigmoidf_fast(z);
  return s * (1.f + z * (1.f - s));
}

// saved register images: one row = one 16-feature tile = f32x4 per lane
template <int NT> __device__ __forceinline__ void load_rows(__amdgpu_buffer_rsrc_t S, int row0, f32x4 (&v)[NT], int v16) {
#pragma unroll
  for (int t = 0; t < NT; ++t) v[t] = bload(S, v16, (row0 + t) * ROW * 4);
}
// wave-private LDS park rows, same image as the scratch rows (16 B per lane, conflict-free)
__device__ __forceinline__ void park_store(float *pk, int row, f32x4 v, int lane) { *(f32x4 *)(pk + row * ROW + lane * 4) = v; }
__device__ __forceinline__ f32x4 park_load(const float *pk, int row, int lane) { return *(const f32x4 *)(pk + row * ROW + lane * 4); }

__device__ __forceinline__ void cutoff_poly(int p, float x, float &f, float &df) {
  if (x >= 1.f) { f = 0.f; df = 0.f; return; }
  float xp1 = 1.f;
  for (int k = 0; k < p - 1; ++k) xp1 *= x;
  const float xp = xp1 * x;
  const float a = 0.5f * (p + 1) * (p + 2), b = (float)p * (p + 2), c = 0.5f * p * (p + 1);
  f = 1.f - a * xp + b * xp * x - c * xp * x * x;
  df = -a * p * xp1 + b * (p + 1) * xp - c * (p + 2) * xp * x;
}

// the same polynomial with its six coefficients precomputed by the host (FusedArgs::cp: wave-uniform kernel arguments instead of per-lane values kept over a tile)
__device__ __forceinline__ void cutoff_poly_c(int p, const float (&c)[6], float x, float &f, float &df) {
  if (x >= 1.f) { f = 0.f; df = 0.f; return; }
  float xp1 = 1.f;
  for (int k = 0; k < p - 1; ++k) xp1 *= x;
  const float xp = xp1 * x;
  f = 1.f - c[0] * xp + c[1] * xp * x - c[2] * xp * x * x;
  df = -c[3] * xp1 + c[4] * xp - c[5] * xp * x;
}

// ---- streamed linear: running weight-fragment ring + epilogue under the next tile pair's MFMAs ----
// The sequence of linears of a tile is static and the host lays the fragments out in consumption order, so
// fragment n of the stream always sits at wp + n*256 floats and the 8-deep ring never drains: each step
// consumes two fragments (output tiles 2p and 2p+1 share the B operand, so their two accumulation chains
// alternate and hide the 16x16x4 MFMA's 8-cycle dependent-issue gap) and requests the two fragments 8 ahead.
// Every streamed linear consumes a multiple of 8 fragments except the four 32x32 channel-mixing blocks
// (4 each), which alternate ring phase RP = 0, 4.  The element-wise epilogue of pair p-1 (SiLU, save,
// scaling, SiLU') is executed one register at a time between the MFMAs of pair p.
struct EpiNone {
  static constexpr bool STORES = false;      // tile_done() stores MFMA result registers (see linear_s)
  __device__ __forceinline__ void tile_done(int, const f32x4 &) const {}
  __device__ __forceinline__ float apply(int, int, float v) const { return v; }
  __device__ __forceinline__ void flush(int) const {}
};
struct EpiSave {             // raw rows to scratch, value unchanged
  static constexpr bool STORES = true;
  __amdgpu_buffer_rsrc_t S; int row0, v16;
  __device__ __forceinline__ void tile_done(int ot, const f32x4 &acc) const { bstore(S, v16, (row0 + ot) * ROW * 4, acc); }
  __device__ __forceinline__ float apply(int, int, float v) const { return v; }
  __device__ __forceinline__ void flush(int) const {}
};
template <int FIRST> struct EpiSaveFrom : EpiSave {  // raw rows to scratch from output tile FIRST on (the backward pass reads only omega's l >= 1 part), value unchanged
  __device__ __forceinline__ void tile_done(int ot, const f32x4 &acc) const { if (ot >= FIRST) EpiSave::tile_done(ot, acc); }
};
typedef EpiSaveFrom<2> EpiSaveFrom2;
struct EpiSavePark : EpiSave {  // raw rows to scratch (for the backward pass) and to the LDS park (next layer's forward)
  float *pk; int prow, lane;
  __device__ __forceinline__ void tile_done(int ot, const f32x4 &acc) const {
    EpiSave::tile_done(ot, acc);
    park_store(pk, prow + ot, acc, lane);
  }
};
struct EpiPark {                // raw rows to the LDS park only
  static constexpr bool STORES = false;
  float *pk; int prow, lane;
  __device__ __forceinline__ void tile_done(int ot, const f32x4 &acc) const { park_store(pk, prow + ot, acc, lane); }
  __device__ __forceinline__ float apply(int, int, float v) const { return v; }
  __device__ __forceinline__ void flush(int) const {}
};
// out = silu(z); the rows saved for the backward pass hold silu'(z) = s + silu (1 - s) (two extra VALU ops here, no
// exp/rcp and no z there): they are written when all 8 registers of a tile pair have gone through apply()
struct EpiSiluSaveD {
  static constexpr bool STORES = false;      // its rows are VALU results, stored from flush()
  __amdgpu_buffer_rsrc_t S; int row0, v16;
  f32x4 d[2];
  __device__ __forceinline__ void tile_done(int, const f32x4 &) const {}
  __device__ __forceinline__ float apply(int ot, int r, float z) {
    const float sg = sigmoidf_fast(z), y = z * sg;
    d[ot & 1][r] = fmaf(y, 1.f - sg, sg);
    return y;
  }
  __device__ __forceinline__ void flush(int ot0) const {
    bstore(S, v16, (row0 + ot0) * ROW * 4, d[0]);
    bstore(S, v16, (row0 + ot0 + 1) * ROW * 4, d[1]);
  }
};
struct EpiSaveScale : EpiSave {  // raw rows to scratch, out = c * v
  float c;
  __device__ __forceinline__ float apply(int, int, float v) const { return c * v; }
};
// The same two, with the rows going to register images inside the wave's own slots of the LDS staging tile (image q of lane (j, g) = floats 16 q + 4 g .. + 3 of
// slot row j: 8 images fit the 128 staged features).  The staging tile is idle between the last layer's environment sum and its backward tensor product, which
// is exactly the life of that layer's activation rows: they never travel to memory (k_fused).
__device__ __forceinline__ void stg_store(float *sq, int q, f32x4 v) { *(f32x4 *)(sq + 16 * q) = v; }
__device__ __forceinline__ f32x4 stg_load(const float *sq, int q) { return *(const f32x4 *)(sq + 16 * q); }
struct EpiSiluSaveDL {
  static constexpr bool STORES = false;
  float *sq; int q0;
  f32x4 d[2];
  __device__ __forceinline__ void tile_done(int, const f32x4 &) const {}
  __device__ __forceinline__ float apply(int ot, int r, float z) {
    const float sg = sigmoidf_fast(z), y = z * sg;
    d[ot & 1][r] = fmaf(y, 1.f - sg, sg);
    return y;
  }
  __device__ __forceinline__ void flush(int ot0) const { stg_store(sq, q0 + ot0, d[0]); stg_store(sq, q0 + ot0 + 1, d[1]); }
};
struct EpiSiluSaveZL {
  static constexpr bool STORES = false;
  float *sq; int q0;
  __device__ __forceinline__ void tile_done(int ot, const f32x4 &acc) const { stg_store(sq, q0 + ot, acc); }
  __device__ __forceinline__ float apply(int, int, float z) const { return z * sigmoidf_fast(z); }
  __device__ __forceinline__ void flush(int) const {}
};
template <int NT> struct EpiMulRows {        // out = v * d (d = the saved silu' rows)
  static constexpr bool STORES = false;
  const f32x4 (&d)[NT];
  __device__ __forceinline__ void tile_done(int, const f32x4 &) const {}
  __device__ __forceinline__ float apply(int ot, int r, float v) const { return v * d[ot][r]; }
  __device__ __forceinline__ void flush(int) const {}
};
// out = silu(z); the RAW pre-activation rows are saved (straight from the result registers): the backward pass rebuilds both silu(z) and silu'(z)
// from them (EpiMulSiluZ), which is what lets it do without the saved u rows of the layer above
struct EpiSiluSaveZ {
  static constexpr bool STORES = true;
  __amdgpu_buffer_rsrc_t S; int row0, v16;
  __device__ __forceinline__ void tile_done(int ot, const f32x4 &acc) const { bstore(S, v16, (row0 + ot) * ROW * 4, acc); }
  __device__ __forceinline__ float apply(int, int, float z) const { return z * sigmoidf_fast(z); }
  __device__ __forceinline__ void flush(int) const {}
};
// Backward twin: q = (g W^T) raw, z = the saved pre-activation rows: out = c q silu'(z), and acc += q silu(z) -- the contraction <u, g> of the layer's output
// u = silu(z) W with its gradient g, i.e. what dE/dfc needs of u, without u: <u, g> = <silu(z), g W^T>.
template <int NT> struct EpiMulSiluZ {
  static constexpr bool STORES = false;
  const f32x4 (&z)[NT]; float c; float &acc;
  __device__ __forceinline__ void tile_done(int, const f32x4 &) const {}
  __device__ __forceinline__ float apply(int ot, int r, float q) const {
    const float zz = z[ot][r], sg = sigmoidf_fast(zz), h = zz * sg;
    acc = fmaf(h, q, acc);
    return q * (c * fmaf(h, 1.f - sg, sg));
  }
  __device__ __forceinline__ void flush(int) const {}
};
template <int NT> struct EpiResidualNS {   // out = ra * xold + rbf * u, nothing saved
  static constexpr bool STORES = false;
  const f32x4 (&xold)[NT]; float ra, rbf;
  __device__ __forceinline__ void tile_done(int, const f32x4 &) const {}
  __device__ __forceinline__ float apply(int ot, int r, float v) const { return ra * xold[ot][r] + rbf * v; }
  __device__ __forceinline__ void flush(int) const {}
};
template <int NT> struct EpiResidual : EpiSave {   // raw u rows to scratch, out = ra * xold + rbf * u
  const f32x4 (&xold)[NT]; float ra, rbf;
  __device__ __forceinline__ float apply(int ot, int r, float v) const { return ra * xold[ot][r] + rbf * v; }
};

template <int KT, int NT, bool ACC, int RP, class Epi>
__device__ __forceinline__ void linear_s(__amdgpu_buffer_rsrc_t W, int &wp, const f32x4 (&in)[KT], f32x4 (&out)[NT], int v16,
                                         f32x4 (&ring)[RING], Epi epi) {
  static_assert(NT % 2 == 0, "output tiles are processed in pairs");
  constexpr int NP = NT / 2, NSTEP = NP * KT, NS = 2 * NSTEP;
  f32x4 acc0, acc1, prev0, prev1;
  // byte offset of the next fragment pair to request: ONE running scalar, advanced step by step.  Written as wp + constant, the
  // optimiser forms every offset of the whole tile (~700 of them) at the top of the tile loop, spills them to VGPR lanes and reads
  // each one back with v_readlane + hazard nops where it is used (1 450 lane operations per wave-tile in fused_lx2.hip).
  int wo = (wp + RING * 256) * 4;
  pin_s(wo);
#ifdef AHIP_LIN_PRIO
  __builtin_amdgcn_s_setprio(AHIP_LIN_PRIO);
#endif
#pragma unroll
  for (int s = 0; s < NSTEP; ++s) {
    const int p = s / KT, kt = s % KT;
    if (kt == 0) {
      if (ACC) { acc0 = out[2 * p]; acc1 = out[2 * p + 1]; }
      else { acc0 = f32x4{0.f, 0.f, 0.f, 0.f}; acc1 = f32x4{0.f, 0.f, 0.f, 0.f}; }
    }
    const f32x4 a0 = ring[(RP + 2 * s) % RING], a1 = ring[(RP + 2 * s + 1) % RING];
    ring[(RP + 2 * s) % RING] = bload_w(W, v16, wo);
    ring[(RP + 2 * s + 1) % RING] = bload_w(W, v16 + 1024, wo);           // + 1 KiB: the instruction's immediate offset
    wo += 2048;
    pin_s(wo);
    // (Rounds 2-3 waited here -- s_waitcnt vmcnt(2) where a tile pair begins -- until the previous pair's accumulator stores had completed: that
    // hid the store-data hazard now padded inside bstore(); with the pad the waits are redundant: 57.3 -> 56.9 ms at 1 M Si, soak green.)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        if (hh == 0) acc0 = mfma16(a0[r], in[kt][r], acc0);
        else acc1 = mfma16(a1[r], in[kt][r], acc1);
        if (p > 0) {
          const int idx = kt * 8 + 2 * r + hh;        // MFMA index inside this pair
          if (idx % KT == 0) {
            const int e = idx / KT;                   // 0..7: element of the previous pair
            if (e < 4) out[2 * (p - 1)][e] = epi.apply(2 * (p - 1), e, prev0[e]);
            else out[2 * (p - 1) + 1][e - 4] = epi.apply(2 * (p - 1) + 1, e - 4, prev1[e - 4]);
            if (e == 7) epi.flush(2 * (p - 1));
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (kt == KT - 1) {
      epi.tile_done(2 * p, acc0);
      epi.tile_done(2 * p + 1, acc1);
      if (p == NP - 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { out[2 * p][r] = epi.apply(2 * p, r, acc0[r]); out[2 * p + 1][r] = epi.apply(2 * p + 1, r, acc1[r]); }
        epi.flush(2 * p);
      } else { prev0 = acc0; prev1 = acc1; }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  wp += NS * 256;
  pin_s(wp);
#ifdef AHIP_LIN_PRIO
  __builtin_amdgcn_s_setprio(AHIP_LIN_PRIO_OUT);
#endif
}
// first RING fragments of the stream at wp into the ring
__device__ __forceinline__ void ring_prime(__amdgpu_buffer_rsrc_t W, int wp, int v16, f32x4 (&ring)[RING]) {
  int wo = wp * 4;
  pin_s(wo);
#pragma unroll
  for (int j = 0; j < RING; ++j) ring[j] = bload_w(W, v16 + (j & 3) * 1024, wo + (j >> 2) * 4096);
}

__device__ __forceinline__ float gsum(float v) {       // sum over the 4 lanes (groups) that share one edge
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}


// ---------------------------------------------------------------------------- tile packing
// Greedy packing of consecutive centre atoms into tiles (<= tile_slots edges, <= maxa atoms), done
// sequentially inside independent segments of SEG atoms so it parallelises.
// maxdeg != null: the tile shape is chosen HERE from the largest degree of the current edge list, which the host has not read back
// (k_fused: 64 slots / 6 centres while every centre has <= 64 edges, else 128 / 12; the host launches both shapes of the model kernel and the one
// that does not match returns at once).
__device__ __forceinline__ void pack_shape(const int *maxdeg, int &tile_slots, int &maxa) {
  if (maxdeg && *maxdeg > 64) { tile_slots = 128; maxa = 12; }
}
template <bool FILL>
static __global__ void k_pack_tiles(int inum, const int *eoff, int nseg, int *seg_count, const int *seg_base, int *tile_a0, int tile_slots, int maxa, const int *maxdeg = nullptr) {
  int sg = blockIdx.x * blockDim.x + threadIdx.x;
  if (sg >= nseg) return;
  pack_shape(maxdeg, tile_slots, maxa);
  int a = sg * SEG, end = min(inum, a + SEG);
  int nt = 0, cur_e = 0, cur_a = 0;
  int base = FILL ? seg_base[sg] : 0;
  if (FILL && a < end) tile_a0[base] = a;
  for (int at = a; at < end; ++at) {
    int deg = eoff[at + 1] - eoff[at];
    if (cur_a == maxa || cur_e + deg > tile_slots) {
      ++nt; cur_e = 0; cur_a = 0;
      if (FILL) tile_a0[base + nt] = at;
    }
    cur_e += deg; ++cur_a;
  }
  if (!FILL) seg_count[sg] = (a < end) ? nt + 1 : 0;
}
static __global__ void k_pack_finish(int inum, int nseg, const int *seg_base, int *tile_a0, int *ntiles) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    int n = seg_base[nseg];
    tile_a0[n] = inum;
    ntiles[0] = n;
    ntiles[1] = 0;          // the fused kernel's tile counter
  }
}
static __global__ void k_centre_info(int inum, const int *ilist, const int *mtype, int2 *centre) {
  int ii = blockIdx.x * blockDim.x + threadIdx.x;
  if (ii < inum) { const int i = ilist[ii]; centre[ii] = make_int2(i, mtype[i]); }
}
// Small systems (<= PACK_SMALL_ATOMS centres, below): the whole tile packing -- segment counts, their scan, the fill, the tile bounds
// and the per-centre {atom, type} records -- in ONE single-workgroup launch instead of six (a 10 648-atom step is launch-bound: 0.056 -> 0.03 ms).
static constexpr int PACK_SMALL_SEGS = 1024;
// One workgroup streams what one CU's L2 port delivers: up to one LDS chunk of centres the single launch wins (10 648 atoms: 29 us), beyond it the six
// launches of the multi-workgroup path do (125 000 atoms: 0.14 ms here; 1 M atoms: 0.065 ms there).
static constexpr int PACK_SMALL_ATOMS = 32768;
static constexpr int PACK_CH = 32768;          // atoms per LDS chunk of k_pack_small (a multiple of SEG): 129 KB of edge offsets at a time
// Round 4: the segment walks read the edge offsets from LDS.  A thread walking its 128 atoms straight from global memory touches one cache line
// per lane and step (the segments are 512 B apart): 48 us for 10 648 atoms, 5 % of a step of that system and paid three times per step by the
// overlapped multi-rank schedule.  Now the offsets of 32 768 atoms at a time are loaded coalesced into LDS (one pad word per segment: the 256
// walkers of a chunk then hit different banks), the walks, the tiles' first edges and the tile bounds come from there.
__device__ __forceinline__ int pack_lds_pos(int k) { return k + (k >> 7); }
static __global__ void __launch_bounds__(PACK_SMALL_SEGS) k_pack_small(int inum, const int *eoff, int nseg, int *tile_a0, int *tile_e0, int *ntiles, int tile_slots, int maxa,
                                                                       const int *ilist, const int *mtype, int2 *centre, const int *maxdeg = nullptr) {
  __shared__ int cnt[PACK_SMALL_SEGS];
  __shared__ int se[PACK_CH + PACK_CH / SEG + 2];
  __shared__ int total;
  const int sg = threadIdx.x;
  pack_shape(maxdeg, tile_slots, maxa);
  // per-centre {atom, type} records: independent of everything below, their two dependent loads run under the first chunk's staging
  // (eight independent load chains in flight per thread: one at a time, a 125 k-atom call spent 0.17 ms in this loop alone)
  for (int i0 = sg; i0 < inum; i0 += 8 * PACK_SMALL_SEGS) {
    int iv[8], tv[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { const int ii = i0 + q * PACK_SMALL_SEGS; iv[q] = ii < inum ? ilist[ii] : 0; }
#pragma unroll
    for (int q = 0; q < 8; ++q) tv[q] = mtype[iv[q]];
#pragma unroll
    for (int q = 0; q < 8; ++q) { const int ii = i0 + q * PACK_SMALL_SEGS; if (ii < inum) centre[ii] = make_int2(iv[q], tv[q]); }
  }
  const int nchunk = (inum + PACK_CH - 1) / PACK_CH;
  const int a = sg * SEG, end = min(inum, a + SEG);
  const int myc = a / PACK_CH, l0 = a - myc * PACK_CH;      // this thread's segment: chunk and first atom inside it
  auto stage = [&](int c) {
    const int c0 = c * PACK_CH, n = min(PACK_CH, inum - c0);
    for (int k0 = sg; k0 <= n; k0 += 8 * PACK_SMALL_SEGS) {
      int v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) { const int k = k0 + q * PACK_SMALL_SEGS; v[q] = k <= n ? eoff[c0 + k] : 0; }
#pragma unroll
      for (int q = 0; q < 8; ++q) { const int k = k0 + q * PACK_SMALL_SEGS; if (k <= n) se[pack_lds_pos(k)] = v[q]; }
    }
  };
  int nt = 0;
  for (int c = 0; c < nchunk; ++c) {
    if (c > 0) __syncthreads();
    stage(c);
    __syncthreads();
    if (sg < nseg && a < end && myc == c) {
      int cur_e = 0, cur_a = 0, prev = se[pack_lds_pos(l0)];
#pragma unroll 8
      for (int k = 1; k <= end - a; ++k) {
        const int nx = se[pack_lds_pos(l0 + k)], deg = nx - prev;
        prev = nx;
        if (cur_a == maxa || cur_e + deg > tile_slots) { ++nt; cur_e = 0; cur_a = 0; }
        cur_e += deg; ++cur_a;
      }
      ++nt;
    }
  }
  cnt[sg] = nt;
  __syncthreads();
  // inclusive scan (Hillis-Steele over the block)
  for (int off = 1; off < PACK_SMALL_SEGS; off <<= 1) {
    const int v = sg >= off ? cnt[sg - off] : 0;
    __syncthreads();
    cnt[sg] += v;
    __syncthreads();
  }
  const int base = cnt[sg] - nt;
  if (sg == PACK_SMALL_SEGS - 1) total = cnt[sg];
  for (int c = 0; c < nchunk; ++c) {
    if (nchunk > 1) { __syncthreads(); stage(c); __syncthreads(); }        // one chunk: its offsets are still staged
    if (sg < nseg && a < end && myc == c) {
      int k = 0, cur_e = 0, cur_a = 0, prev = se[pack_lds_pos(l0)];
      tile_a0[base] = a; tile_e0[base] = prev;
#pragma unroll 8
      for (int q = 1; q <= end - a; ++q) {
        const int nx = se[pack_lds_pos(l0 + q)], deg = nx - prev;
        if (cur_a == maxa || cur_e + deg > tile_slots) { ++k; cur_e = 0; cur_a = 0; tile_a0[base + k] = a + q - 1; tile_e0[base + k] = prev; }
        prev = nx;
        cur_e += deg; ++cur_a;
      }
    }
  }
  __syncthreads();
  if (sg == 0) { const int n = total; tile_a0[n] = inum; tile_e0[n] = eoff[inum]; ntiles[0] = n; ntiles[1] = 0; }
}
// packed per-edge types when the edge list did not come from the single-pass build (edges.hip writes them itself)
static __global__ void k_edge_types(long long E, const int *e_ii, const int *e_j, const int *ilist, const int *mtype, unsigned char *e_tt) {
  long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < E) e_tt[e] = (unsigned char)((mtype[ilist[e_ii[e]]] << 4) | mtype[e_j[e]]);
}
static __global__ void k_tile_e0(const int *ntiles, const int *tile_a0, const int *eoff, int *tile_e0) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t <= *ntiles) tile_e0[t] = eoff[tile_a0[t]];
}

// Fragment tiling of a [K][N] linear: KT input tiles x NT output tiles of 16 features; NT is padded to
// even (tile pairs), and a single-input-tile linear is padded to KT = 2 so that it consumes 8 fragments.
static void frag_dims(int K, int N, int &KT, int &NT) {
  KT = (K + 15) / 16;
  NT = (N + 15) / 16;
  NT += NT & 1;
  if (KT == 1) KT = 2;
}
// A-operand fragments of W [K][N] (row-major, x @ W) in consumption order [pair p][kt][half][lane][r]:
//   value = W[16 kt + 4 (lane>>4) + r][16 (2p+half) + (lane & 15)], zero padded.
static int append_frag(std::vector<float> &out, const double *W, int K, int N, int ldw) {
  int KT, NT;
  frag_dims(K, N, KT, NT);
  for (int p = 0; p < NT / 2; ++p)
    for (int kt = 0; kt < KT; ++kt)
      for (int half = 0; half < 2; ++half)
        for (int lane = 0; lane < 64; ++lane)
          for (int r = 0; r < 4; ++r) {
            int k = feat16(kt, r, lane >> 4), n = 16 * (2 * p + half) + (lane & 15);
            out.push_back((k < K && n < N) ? (float)W[(size_t)k * ldw + n] : 0.f);
          }
  return KT * NT;
}
static std::vector<double> transpose(const double *W, int K, int N) {
  std::vector<double> t((size_t)K * N);
  for (int k = 0; k < K; ++k)
    for (int n = 0; n < N; ++n) t[(size_t)n * K + k] = W[(size_t)k * N + n];
  return t;
}

// Tabulated two-body embedding: x0(d; type pair) = f_c * MLP([one-hots, Bessel * f_c]) depends on the edge only through
// (d, t_i, t_j), so it is evaluated here in float64 -- with its exact d/dd (forward mode) -- at NK + 1 knots per type pair and
// stored as cubic-Hermite coefficients [pair][NK intervals][tile 4][coef 4][16]; the kernels gather 16 x 16 B per lane.
static void append_two_body_table(std::vector<float> &w, const HostModel &h, const std::vector<double> &rcut_model_host, int NKin) {
  const int T = h.num_types;
  const HostTensor &w0 = h.get("tb.w0");          // [2T+B][64]
  const double *wc = w0.data.data() + (size_t)2 * T * 64;       // Bessel block [B][64], B = num_bessels: any count (the kernels never see the basis, only the table)
  auto T_ = [&](const std::string &name) -> const double * { return h.get(name).data.data(); };
  struct { const std::vector<double> &rcut_model_host; } m{rcut_model_host};
  const int NK = NKin;
  const double PI = 3.14159265358979323846;
  // hidden layers beyond the first: tb.w1 .. tb.w{depth-1} (64 x 64), output layer tb.w{depth}; depth = the model's MLP depth (1..3 on the fused paths)
  const int depth = h.mlp_depth;
  std::vector<const double *> Wh;
  for (int k = 1; k <= depth; ++k) Wh.push_back(T_("tb.w" + std::to_string(k)));
  auto silu = [](double z) { return z / (1.0 + std::exp(-z)); };
  auto dsilu = [](double z) { const double sg = 1.0 / (1.0 + std::exp(-z)); return sg * (1.0 + z * (1.0 - sg)); };
  std::vector<double> y((size_t)(NK + 1) * 64), dy((size_t)(NK + 1) * 64);
  for (int ti = 0; ti < T; ++ti)
    for (int tj = 0; tj < T; ++tj) {
      const double rc = m.rcut_model_host[(size_t)ti * T + tj];
      const double hstep = rc / NK;
      for (int k = 0; k <= NK; ++k) {
        const double d = k * hstep, xq = d / rc;
        double fcv = 0, dfc = 0;                       // cutoff envelope and d/dx
        if (xq < 1.0) {
          const int p = h.poly_p;
          const double xp1 = std::pow(xq, p - 1), xp = xp1 * xq;
          const double ca = 0.5 * (p + 1) * (p + 2), cb = (double)p * (p + 2), cc = 0.5 * p * (p + 1);
          fcv = 1.0 - ca * xp + cb * xp * xq - cc * xp * xq * xq;
          dfc = -ca * p * xp1 + cb * (p + 1) * xp - cc * (p + 2) * xp * xq;
        }
        double z[64], dz[64], hh[64], dh[64];
        for (int n = 0; n < 64; ++n) { z[n] = w0.data[(size_t)ti * 64 + n] + w0.data[(size_t)(T + tj) * 64 + n]; dz[n] = 0; }
        for (int b = 1; b <= h.num_bessels; ++b) {
          const double a = b * PI / rc;
          double sv, ds;                                // s = sin(a d)/d and ds/dd, series near 0
          if (a * d < 1e-4) { sv = a * (1.0 - a * a * d * d / 6.0); ds = -a * a * a * d / 3.0; }
          else { sv = std::sin(a * d) / d; ds = (a * d * std::cos(a * d) - std::sin(a * d)) / (d * d); }
          const double bf = 2.0 / rc * sv * fcv, dbf = 2.0 / rc * (ds * fcv + sv * dfc / rc);
          for (int n = 0; n < 64; ++n) { z[n] += wc[(size_t)(b - 1) * 64 + n] * bf; dz[n] += wc[(size_t)(b - 1) * 64 + n] * dbf; }
        }
        for (int n = 0; n < 64; ++n) { hh[n] = silu(z[n]); dh[n] = dsilu(z[n]) * dz[n]; }
        for (int l = 0; l + 1 < depth; ++l) {            // hidden layers 2 .. depth, value and d/dd (forward mode)
          for (int n = 0; n < 64; ++n) { z[n] = 0; dz[n] = 0; }
          for (int q = 0; q < 64; ++q)
            for (int n = 0; n < 64; ++n) { z[n] += hh[q] * Wh[l][(size_t)q * 64 + n]; dz[n] += dh[q] * Wh[l][(size_t)q * 64 + n]; }
          for (int n = 0; n < 64; ++n) { hh[n] = silu(z[n]); dh[n] = dsilu(z[n]) * dz[n]; }
        }
        const double *Wo = Wh[depth - 1];
        for (int n = 0; n < 64; ++n) {
          double u = 0, du = 0;
          for (int q = 0; q < 64; ++q) { u += hh[q] * Wo[(size_t)q * 64 + n]; du += dh[q] * Wo[(size_t)q * 64 + n]; }
          y[(size_t)k * 64 + n] = fcv * u;
          dy[(size_t)k * 64 + n] = dfc / rc * u + fcv * du;
        }
      }
      for (int k = 0; k < NK; ++k)
        for (int t = 0; t < 4; ++t)
          for (int c = 0; c < 4; ++c)
            for (int q = 0; q < 16; ++q) {
              const int f = 16 * t + q;               // lane (j, g) reads the 4 floats at 4 g: features 16 t + 4 g + r
              const double y0 = y[(size_t)k * 64 + f], y1 = y[(size_t)(k + 1) * 64 + f];
              const double m0 = hstep * dy[(size_t)k * 64 + f], m1 = hstep * dy[(size_t)(k + 1) * 64 + f];
              const double cf = c == 0 ? y0 : c == 1 ? m0 : c == 2 ? 3.0 * (y1 - y0) - 2.0 * m0 - m1 : 2.0 * (y0 - y1) + m0 + m1;
              w.push_back((float)cf);
            }
    }
}
}  // namespace ahip
