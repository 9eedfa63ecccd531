// Single-pass edge build for the float32 production path: cutoff filter of the skin-inflated full
// neighbor list + CSR->COO expansion (reference: preprocess() passes 1+2, /root/reference/
// pair_nequip_allegro.cpp:488-512,566-629; Kokkos K1,K2,K5, pair_nequip_allegro_kokkos.cpp:165-258).
//
//  * one wave per centre row: the row's neighbour indices are read coalesced (64 per instruction), the
//    gathered x_j / type_j are the only random traffic; `rsq <= cut^2` in float64 (host-path semantics);
//  * survivors are compacted with ballot/mbcnt and kept in registers (8 centres per wave), the loads of the 8 centres are
//    issued together (8 gathers per lane in flight) and the next unit's row bounds / indices are requested a unit ahead;
//  * the global edge offsets come from a decoupled look-back scan over 64-centre units (unit order taken
//    from an atomic ticket, so a unit's predecessors have always started), so the neighbour data is
//    gathered ONCE -- the two-pass version (generic_kernels.h) gathers it twice;
//  * every block then writes its edges to one contiguous range of e_ii / e_j / rvec.
// Output layout is identical to k_count_edges + scan + k_fill_edges: edges grouped by centre, list order.
// Rows longer than 128 entries raise the overflow flag and the caller re-runs the two-pass kernels.
#include <hip/hip_runtime.h>

#include <cstring>
#include <mutex>

#include "engine.h"

namespace ahip {

static constexpr int EB_ATOMS = 64;      // centres per scan unit (one status word each): 8 waves x 8 centres
static constexpr int EB_THREADS = EB_ATOMS / 8 * 64;
static constexpr int LB_PER_LANE = 1;   // look-back window = 64 * LB_PER_LANE predecessors per poll (8: 0.93 ms instead of 0.70 -- the polls themselves load the L2)
// The kernel is PERSISTENT: a grid of 8-wave workgroups the size of the residency (2 per CU = the 4 waves per SIMD its registers
// admit; 1 per CU for the two-chunk instance; queried with hipOccupancyMaxActiveBlocksPerMultiprocessor) walks the 64-centre scan
// units.  The look-back costs (units) x (half the grid / 64) dependent polls in total, i.e. it shrinks with the unit size: 64 centres.
// Unit order: a fixed stride per workgroup, or one atomic ticket per unit where co-residency of the grid is not a given (see the kernel).
static constexpr int EB_PER_WAVE = 8;    // centres per wave
// EB_CHUNKS (template parameter): 64-entry chunks of a list row held in registers: 1 when no row of the installed list is longer
// than 64 entries (half the registers: 5 instead of 3 waves per SIMD in flight for this latency-bound gather), else 2

// {x, y, z, types} per atom, 32-byte aligned: ONE sector per gathered neighbour instead of two or three (the 24-byte position
// straddles sectors, the type sits in another array).  Rewritten every step by k_pack_xt (a 60 MB stream at 1 M atoms).
struct __attribute__((aligned(32))) AtomXT { double x, y, z; int ft, mt; };

// It also clears the edge build's header + look-back status words and the caller's 7 energy / virial sums (two memset launches less per call:
// a small system's step is a chain of short launches).
__global__ void __launch_bounds__(256) k_pack_xt(int nall, const double *__restrict__ x, const int *__restrict__ ftype,
                                                  const int *__restrict__ mtype, AtomXT *__restrict__ xt, unsigned long long *clear, int nclear,
                                                  double *zero7) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < nclear) clear[i] = 0ull;
  if (i < 7) zero7[i] = 0.0;
  if (i >= nall) return;
  AtomXT a;
  a.x = x[3 * (size_t)i]; a.y = x[3 * (size_t)i + 1]; a.z = x[3 * (size_t)i + 2];
  a.ft = ftype[i]; a.mt = mtype[i];
  xt[i] = a;
}

__device__ __forceinline__ unsigned long long pack_state(unsigned long long state, unsigned long long v) { return (state << 62) | v; }

template <int EB_CHUNKS, bool DYN>
__global__ void __launch_bounds__(EB_THREADS) k_build_edges(int inum, const int *__restrict__ ilist, const int *__restrict__ nl_off,
                                                       const int *__restrict__ nl_j, const AtomXT *__restrict__ xt,
                                                       const double *__restrict__ cutsq, int nft, int nunits,
                                                       unsigned int *ticket, unsigned long long *status, int *eoff, int *e_ii,
                                                       int *e_j, float *rvec, int *maxdeg, int *overflow,
                                                       unsigned char *e_tt, int heavy_thresh,
                                                       int *heavy_cnt, int *heavy_list, int *total_out,
                                                       int pack_slots, int pack_maxa, int *tile_a0, int *tile_e0, int *ntiles, int2 *centre) {
  __shared__ int s_cnt2[2][EB_ATOMS];          // counters of two consecutive units (no barrier between the stores of one and the
  __shared__ int s_base2[2][EB_ATOMS + 1];     // counting of the next)
  __shared__ int s_blk, s_claim2[2];           // claims double-buffered by unit parity like the counters: a wave may still be reading one while thread 0 posts the next
  __shared__ long long s_prefix2[2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // Two unit schedules.  DYN = false: the workgroup takes ONE ticket t and walks the units t, t + G, t + 2 G ... (G = grid size).  The
  // look-back of a unit waits for the unit before it, i.e. for the previous ticket of the same round: safe only while all G workgroups
  // are resident (the host sizes the grid from the occupancy query and serialises the launches of this kernel inside the process).
  // DYN = true: units are claimed ONE AT A TIME from the ticket counter, two units ahead of their use (the claim's round trip runs
  // under a whole unit): every unit below a claimed one has then been claimed by a workgroup that is running, so the look-back cannot
  // wait for a workgroup that has not started -- whatever the grid size and whatever else occupies the chip (a second model on another
  // stream, ranks sharing the GPU, exchange kernels).  It costs one device-scope atomic per unit on one address: +0.19 ms at 1 M atoms
  // (0.55 -> 0.74 ms), so the host picks it only where co-residency is not a given (edges_build_f32).
  if (tid == 0) { s_blk = (int)atomicAdd(ticket, 1u); if (DYN) s_claim2[0] = (int)atomicAdd(ticket, 1u); }
  __syncthreads();
  const int b0 = s_blk;
  int bn = DYN ? s_claim2[0] : b0 + (int)gridDim.x;       // the unit after the current one
  const int uwave = __builtin_amdgcn_readfirstlane(wave);
  int run_max = 0;                       // largest edge count seen by this workgroup (thread 0), published once at the end
  constexpr int NB = EB_CHUNKS == 1 ? EB_PER_WAVE : 2;
  // Row bounds, centre indices and the first batch's neighbour indices of the NEXT unit are requested while the current one is
  // in its scan / look-back / store phases: at the top of the loop only the gathers (one memory round trip) are still to be
  // issued instead of a chain of three.
  int n_ci[EB_PER_WAVE], n_p0[EB_PER_WAVE], n_p1[EB_PER_WAVE], n_js[NB][EB_CHUNKS];
  int l_il = 0, l_off = 0;
  auto request_rows = [&](int bb) {      // lanes 0..7: centre indices, lanes 0..8: row offsets (clamped: past the end = empty rows)
    const int ii0 = bb * EB_ATOMS + uwave * EB_PER_WAVE;
    l_il = ilist[min(ii0 + min(lane, EB_PER_WAVE - 1), inum - 1)];
    l_off = nl_off[min(ii0 + min(lane, EB_PER_WAVE), inum)];
  };
  auto take_rows = [&]() {
#pragma unroll
    for (int q = 0; q < EB_PER_WAVE; ++q) {
      n_ci[q] = __builtin_amdgcn_readlane(l_il, q);
      n_p0[q] = __builtin_amdgcn_readlane(l_off, q);
      n_p1[q] = __builtin_amdgcn_readlane(l_off, q + 1);
    }
#pragma unroll
    for (int q = 0; q < NB; ++q)
#pragma unroll
      for (int c = 0; c < EB_CHUNKS; ++c) {
        const int p = n_p0[q] + c * 64 + lane;
        n_js[q][c] = p < n_p1[q] ? nl_j[p] : n_ci[q];          // lanes past the row end gather the centre itself (and drop it)
      }
  };
  request_rows(b0);
  take_rows();
  int par = 0, bnn = 0;
  for (int b = b0; b < nunits; b = bn, bn = bnn, par ^= 1) {
  const int a_begin = b * EB_ATOMS;
  if (DYN && tid == 0) s_claim2[par ^ 1] = (int)atomicAdd(ticket, 1u);        // the unit after next; read behind this unit's second barrier
  int *s_cnt = s_cnt2[par], *s_base = s_base2[par];
  long long &s_prefix = s_prefix2[par];

  // ---- gather + filter: results stay in registers -------------------------------------------------
  // Loads are issued in BATCHES of NB centres (8 gathers per lane in flight): first the row bounds and the centre records of
  // the batch (wave-uniform: scalar loads), then all neighbour indices, then all gathers, and only then the arithmetic and the
  // ballots.  One centre at a time (load -> dependent load -> ballot) left a single four-deep dependent chain per wave in flight
  // and made the kernel latency-bound at half its run time.
  int jj[EB_PER_WAVE][EB_CHUNKS];
  float dxs[EB_PER_WAVE][EB_CHUNKS], dys[EB_PER_WAVE][EB_CHUNKS], dzs[EB_PER_WAVE][EB_CHUNKS];
  int rank[EB_PER_WAVE][EB_CHUNKS];      // -1 = dropped, else position inside the centre's edge range
  int tts[EB_PER_WAVE][EB_CHUNKS];       // (model type of centre) << 4 | (model type of neighbour), for the fused kernel
  int kept_k[EB_PER_WAVE];
  int c_ci[EB_PER_WAVE], c_p0[EB_PER_WAVE], c_p1[EB_PER_WAVE];
#pragma unroll
  for (int q = 0; q < EB_PER_WAVE; ++q) { c_ci[q] = n_ci[q]; c_p0[q] = n_p0[q]; c_p1[q] = n_p1[q]; }
#pragma unroll
  for (int kb = 0; kb < EB_PER_WAVE; kb += NB) {
    int ci_i[NB], p0s[NB], p1s[NB];
    AtomXT cis[NB];
#pragma unroll
    for (int q = 0; q < NB; ++q) { ci_i[q] = c_ci[kb + q]; p0s[q] = c_p0[kb + q]; p1s[q] = c_p1[kb + q]; }
#pragma unroll
    for (int q = 0; q < NB; ++q) cis[q] = xt[ci_i[q]];
    int js[NB][EB_CHUNKS];
#pragma unroll
    for (int q = 0; q < NB; ++q) {
      if (p1s[q] - p0s[q] > 64 * EB_CHUNKS && lane == 0) atomicOr(overflow, 1);
#pragma unroll
      for (int c = 0; c < EB_CHUNKS; ++c) {
        if (kb == 0) js[q][c] = n_js[q][c];                    // requested during the previous unit
        else {
          const int p = p0s[q] + c * 64 + lane;
          js[q][c] = p < p1s[q] ? nl_j[p] : ci_i[q];
        }
      }
    }
    AtomXT cjs[NB][EB_CHUNKS];
#pragma unroll
    for (int q = 0; q < NB; ++q)
#pragma unroll
      for (int c = 0; c < EB_CHUNKS; ++c) cjs[q][c] = xt[js[q][c]];    // two 16-byte loads of one 32-byte sector
#pragma unroll
    for (int q = 0; q < NB; ++q) {
      const int k = kb + q;
      const double xi = cis[q].x, yi = cis[q].y, zi = cis[q].z;
      const double *crow = cutsq + (size_t)cis[q].ft * nft;
      const int mti = cis[q].mt;
      int kept = 0;
#pragma unroll
      for (int c = 0; c < EB_CHUNKS; ++c) {
        const bool valid = p0s[q] + c * 64 + lane < p1s[q];
        const AtomXT cj = cjs[q][c];
        const double ddx = cj.x - xi, ddy = cj.y - yi, ddz = cj.z - zi;
        const double rsq = ddx * ddx + ddy * ddy + ddz * ddz;
        const bool keep = valid && rsq <= crow[cj.ft];
        const unsigned long long mask = __ballot(keep);
        const int below = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
        jj[k][c] = js[q][c]; tts[k][c] = (mti << 4) | cj.mt;
        dxs[k][c] = (float)ddx; dys[k][c] = (float)ddy; dzs[k][c] = (float)ddz;       // neighbour - centre, f64 difference cast to f32
        rank[k][c] = keep ? kept + below : -1;
        kept += __popcll(mask);
      }
      kept_k[k] = kept;
      if (lane == 0) {
        const int la = uwave * EB_PER_WAVE + k;
        s_cnt[la] = kept;
        if (pack_slots > 0 && a_begin + la < inum) centre[a_begin + la] = make_int2(ci_i[q], mti);       // per-centre record of the fused kernels (tile packing below)
        // centres with more edges than a tile of the wide fused kernel holds: listed for the layer-at-a-time kernels
        if (heavy_thresh > 0 && kept > heavy_thresh) heavy_list[atomicAdd(heavy_cnt, 1)] = a_begin + la;
      }
    }
  }
  request_rows(bn);                      // answered during the scan and the look-back
  __syncthreads();

  // ---- block scan of the 64 counts, then decoupled look-back for the block's global offset ----------
  if (tid < 64) {
    int v = tid < EB_ATOMS ? s_cnt[tid] : 0;
    int inc = v;
#pragma unroll
    for (int off = 1; off < EB_ATOMS; off <<= 1) {
      int t = __shfl_up(inc, off, 64);
      if (lane >= off) inc += t;
    }
    if (tid < EB_ATOMS) s_base[tid + 1] = inc;
    if (tid == 0) s_base[0] = 0;
    int mx = v;
#pragma unroll
    for (int off = EB_ATOMS / 2; off > 0; off >>= 1) mx = max(mx, __shfl_xor(mx, off, 64));
    run_max = max(run_max, mx);
    // decoupled look-back, one WAVE wide: lane l polls predecessor b-1-l; the window closes at the nearest predecessor that has
    // published its inclusive prefix (state 2), everything nearer contributes its aggregate (state 1).  (A one-lane walk costs
    // one dependent global load per predecessor: with ~1000 blocks resident the chain was the kernel's critical path.)
    unsigned long long agg = (unsigned long long)__shfl(inc, EB_ATOMS - 1, 64);
    // ---- tile packing of the fused model kernels, fused into this pass (round 4) ----
    // Greedy packing of consecutive centres into tiles of <= pack_slots edges and <= pack_maxa centres, restarted at every unit (64 centres; the
    // stand-alone packing kernels restart every 128).  Lane k holds centre k's degree v and the prefixes ex / inc: the tile that would START at k
    // ends before nxt[k] = the first later centre that does not fit; the actual starts are the chain 0 -> nxt[0] -> nxt[nxt[0]] ..., walked on the
    // scalar unit (<= 64 readlanes).  The tile count travels through the look-back in the upper 26 bits of the same 62-bit value as the edge count
    // (sums of both stay inside their fields: < 2^36 edges, < 2^26 tiles), so a unit learns its first tile index together with its first edge.
    unsigned long long tmask = 0;
    const int ex = inc - v;
    const int nun = min(EB_ATOMS, inum - a_begin);
    if (pack_slots > 0) {
      int nx = lane + 1;
      bool open = lane < nun;
      for (int q = 1; q <= pack_maxa; ++q) {
        const int jn = lane + q;
        const int incj = __shfl(inc, min(jn, EB_ATOMS - 1), 64);
        if (open) {
          if (jn >= nun) { nx = nun; open = false; }
          else if (q == pack_maxa || incj - ex > pack_slots) { nx = jn; open = false; }
        }
        if (!__any(open)) break;
      }
      int k = 0;
      while (k < nun) { tmask |= 1ull << k; k = __builtin_amdgcn_readlane(nx, k); }
      agg |= (unsigned long long)__popcll(tmask) << 36;
    }
    const unsigned long long VMASK = (1ull << 62) - 1;
    unsigned long long prefix = 0;
    if (b == 0) {
      if (lane == 0) __hip_atomic_store(&status[0], pack_state(2, agg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      if (lane == 0) __hip_atomic_store(&status[b], pack_state(1, agg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int base = b - 1;
      for (;;) {
        // lane l polls the LB_PER_LANE predecessors base - LB_PER_LANE * l - q (independent loads, one round trip for a window of
        // 64 * LB_PER_LANE units = a whole round of the resident grid: the units of a round publish their aggregates at about
        // the same time, so a 64-wide window walked the round in ~8 dependent round trips)
        unsigned long long sum_pref = 0;
        bool ready_pref = true, has2 = false;
        unsigned long long v[LB_PER_LANE];
#pragma unroll
        for (int q = 0; q < LB_PER_LANE; ++q) {
          const int k = base - (lane * LB_PER_LANE + q);
          v[q] = k >= 0 ? __hip_atomic_load(&status[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : pack_state(2, 0);
        }
#pragma unroll
        for (int q = 0; q < LB_PER_LANE; ++q) {
          const unsigned st2 = (unsigned)(v[q] >> 62);
          if (!has2) { ready_pref = ready_pref && st2 != 0; sum_pref += v[q] & VMASK; has2 = st2 == 2; }
        }
        const unsigned long long incl = __ballot(has2), ready = __ballot(ready_pref);
        const int first2 = incl ? __builtin_ctzll(incl) : 64;                 // lane holding the nearest inclusive prefix
        const unsigned long long need = first2 < 63 ? ((2ull << first2) - 1) : ~0ull;
        if ((ready & need) == need) {
          unsigned long long c = lane <= first2 ? sum_pref : 0ull;
#pragma unroll
          for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
          prefix += c;
          if (first2 < 64) break;
          base -= 64 * LB_PER_LANE;
        } else __builtin_amdgcn_s_sleep(1);
      }
      if (lane == 0) __hip_atomic_store(&status[b], pack_state(2, prefix + agg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (lane == 0) s_prefix = (long long)(prefix & ((1ull << 36) - 1));
    if (pack_slots > 0) {
      const int tbase = (int)(prefix >> 36), gb = (int)(prefix & ((1ull << 36) - 1));
      if ((tmask >> lane) & 1ull) {
        const int ti = tbase + __popcll(tmask & ((1ull << lane) - 1));
        tile_a0[ti] = a_begin + lane;
        tile_e0[ti] = gb + ex;
      }
      if (lane == 0 && a_begin + EB_ATOMS >= inum) {            // last unit: closing bounds, tile total, the model kernel's tile counter
        const int nt = tbase + __popcll(tmask);
        tile_a0[nt] = inum;
        tile_e0[nt] = gb + (int)(agg & ((1ull << 36) - 1));
        ntiles[0] = nt; ntiles[1] = 0;
      }
    }
  }
  __syncthreads();
  const long long gbase = s_prefix;
  bnn = DYN ? s_claim2[par ^ 1] : bn + (int)gridDim.x;

  // ---- offsets + edges -------------------------------------------------------------------------------
  take_rows();                           // next unit's rows have arrived; its neighbour indices travel during the stores
  if (tid < EB_ATOMS && a_begin + tid < inum) eoff[a_begin + tid] = (int)(gbase + s_base[tid]);
  if (tid == 0 && a_begin + EB_ATOMS >= inum) { const int tot = (int)(gbase + s_base[min(EB_ATOMS, inum - a_begin)]); eoff[inum] = tot; *total_out = tot; }    // the total also next to the other counters: one read-back
#pragma unroll
  for (int k = 0; k < EB_PER_WAVE; ++k) {
    const int la = wave * EB_PER_WAVE + k;
    const int ii = a_begin + la;
    const long long ebase = gbase + s_base[la];
#pragma unroll
    for (int c = 0; c < EB_CHUNKS; ++c) {
      const int r = rank[k][c];
      if (r >= 0) {
        const long long e = ebase + r;
        e_ii[e] = ii;
        e_j[e] = jj[k][c];
        rvec[3 * e] = dxs[k][c]; rvec[3 * e + 1] = dys[k][c]; rvec[3 * e + 2] = dzs[k][c];
        e_tt[e] = (unsigned char)tts[k][c];
      }
    }
  }
  }
  if (tid == 0 && run_max > 0) atomicMax(maxdeg, run_max);
}

// h_back: pinned read-back words of the last BACK_N calls (8 ints each; [1] largest degree, [2] row-overflow flag, [3] heavy centres, [4] edge total), ev_back[k]:
// recorded behind the copy of slot k.  A ring, so that the overflow flag of EVERY call is looked at -- for free: by the time a later call (or edges_counts)
// comes by, the word has landed -- although no call waits for its own copy (the flag used to be read only when the row bound was unknown).
static constexpr int BACK_N = 8;
struct EdgeState {
  DevBuf flags, heavy, hoff, xt; int ncu = 0, occ[2] = {1, 1};
  int *h_back = nullptr; hipEvent_t ev_back[BACK_N] = {}, chain = nullptr; bool unchecked[BACK_N] = {}; int cur = 0;
};
// Looks at slot k's overflow word once its copy has completed (wait: block until it has).  A set flag means a list row was longer than the bound measured at the
// list's hand-over: the single-pass build read only the first 64 / 128 entries of that row, the forces of that call are wrong.
static void back_check(EdgeState &st, int k, bool wait) {
  if (!st.unchecked[k]) return;
  if (wait) AHIP_CHECK(hipEventSynchronize(st.ev_back[k]));
  else if (hipEventQuery(st.ev_back[k]) != hipSuccess) return;
  st.unchecked[k] = false;
  if (st.h_back[8 * k + 2] != 0)
    throw StateError("neighbor list row longer than the bound measured at its hand-over (ahip_neigh_update*): the CSR arrays handed to the library must not change "
                     "between hand-overs; the forces of the last evaluation are invalid");
}

// ---- compact copy of the edges of the listed ("heavy") centres: the edge list the layer-at-a-time kernels run on ----
static __global__ void k_heavy_offsets(int nh, const int *heavy, const int *eoff, int *hoff) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    int acc = 0;
    for (int k = 0; k < nh; ++k) { hoff[k] = acc; acc += eoff[heavy[k] + 1] - eoff[heavy[k]]; }
    hoff[nh] = acc;
  }
}
static __global__ void k_heavy_copy(int nh, const int *heavy, const int *ilist, const int *eoff, const int *e_j, const float *rvec,
                                    const int *hoff, int *h_ilist, int *h_eii, int *h_ej, float *h_rvec) {
  const int k = blockIdx.x;
  if (k >= nh) return;
  const int ii = heavy[k], e0 = eoff[ii], n = eoff[ii + 1] - e0, o = hoff[k];
  if (threadIdx.x == 0) h_ilist[k] = ilist[ii];
  for (int q = threadIdx.x; q < n; q += blockDim.x) {
    h_eii[o + q] = k;
    h_ej[o + q] = e_j[e0 + q];
    h_rvec[3 * (size_t)(o + q)] = rvec[3 * (size_t)(e0 + q)];
    h_rvec[3 * (size_t)(o + q) + 1] = rvec[3 * (size_t)(e0 + q) + 1];
    h_rvec[3 * (size_t)(o + q) + 2] = rvec[3 * (size_t)(e0 + q) + 2];
  }
}

static __global__ void k_max_row(int inum, const int *off, int *out) {
  int ii = blockIdx.x * blockDim.x + threadIdx.x;
  int v = ii < inum ? off[ii + 1] - off[ii] : 0;
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
  if ((threadIdx.x & 63) == 0 && v > 0) atomicMax(out, v);
}
int edges_max_row(Model &m, int inum, const int *offsets_dev) {
  if (inum <= 0) return 0;
  AHIP_CHECK(hipDeviceSynchronize());        // the list may have been written on any stream of the caller
  m.b_misc.reserve(64);
  int *out = m.b_misc.as<int>();
  AHIP_CHECK(hipMemsetAsync(out, 0, sizeof(int), nullptr));
  hipLaunchKernelGGL(k_max_row, dim3((inum + 255) / 256), dim3(256), 0, nullptr, inum, offsets_dev, out);
  AHIP_CHECK(hipGetLastError());            // a launch that did not happen would leave the bound at 0, i.e. "every row fits"
  int h = 0;
  AHIP_CHECK(hipMemcpy(&h, out, sizeof(int), hipMemcpyDeviceToHost));
  return h;
}

void edges_counts(Model &m) {
  if (!m.counts_pending) return;
  EdgeState &st = *(EdgeState *)m.edge_state;
  AHIP_CHECK(hipEventSynchronize(st.ev_back[st.cur]));       // the copy sits right behind the edge build in its stream: this does not wait for the model kernel
  const int *h3 = st.h_back + 8 * st.cur;
  back_check(st, st.cur, false);
  m.nedges = h3[4];
  m.nedges_hint = m.nedges;
  m.last_max_deg = h3[1];
  m.nheavy = m.heavy_thresh > 0 ? h3[3] : 0;
  m.counts_pending = false;
}

void edges_compact_heavy(Model &m, const ComputeArgs &a) {
  EdgeState &st = *(EdgeState *)m.edge_state;
  const int nh = m.nheavy;
  m.hv_eoff.reserve((size_t)(nh + 2) * sizeof(int));
  m.hv_ilist.reserve((size_t)(nh + 1) * sizeof(int));
  hipLaunchKernelGGL(k_heavy_offsets, dim3(1), dim3(64), 0, a.stream, nh, st.heavy.as<int>(), m.b_eoff.as<int>(), m.hv_eoff.as<int>());
  int tot = 0;
  AHIP_CHECK(hipMemcpyAsync(&tot, m.hv_eoff.as<int>() + nh, sizeof(int), hipMemcpyDeviceToHost, a.stream));
  AHIP_CHECK(hipStreamSynchronize(a.stream));
  m.hv_nedges = tot;
  const size_t E = (size_t)std::max(tot, 1);
  m.hv_eii.reserve(E * sizeof(int));
  m.hv_ej.reserve(E * sizeof(int));
  m.hv_rvec.reserve(E * 3 * sizeof(float));
  hipLaunchKernelGGL(k_heavy_copy, dim3(nh), dim3(64), 0, a.stream, nh, st.heavy.as<int>(), m.d_ilist, m.b_eoff.as<int>(), m.b_ej.as<int>(),
                     m.b_rvec.as<float>(), m.hv_eoff.as<int>(), m.hv_ilist.as<int>(), m.hv_eii.as<int>(), m.hv_ej.as<int>(), m.hv_rvec.as<float>());
  AHIP_CHECK(hipGetLastError());
}

bool edges_build_f32(Model &m, const ComputeArgs &a) {
  StageTimer tm(m, "edge_build", a.stream);
  const int inum = m.inum;
  const int nunits = (inum + EB_ATOMS - 1) / EB_ATOMS;
  if (!m.edge_state) m.edge_state = new EdgeState();
  EdgeState &st = *(EdgeState *)m.edge_state;
  if (st.ncu == 0) {                      // once per model: CU count and the residency of both instances
    hipDeviceProp_t prop;
    st.ncu = (hipGetDeviceProperties(&prop, m.device) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&st.occ[0], k_build_edges<1, false>, EB_THREADS, 0) != hipSuccess || st.occ[0] < 1) st.occ[0] = 1;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&st.occ[1], k_build_edges<2, false>, EB_THREADS, 0) != hipSuccess || st.occ[1] < 1) st.occ[1] = 1;
    AHIP_CHECK(hipHostMalloc((void **)&st.h_back, BACK_N * 8 * sizeof(int), hipHostMallocDefault));
    std::memset(st.h_back, 0, BACK_N * 8 * sizeof(int));
    for (int k = 0; k < BACK_N; ++k) AHIP_CHECK(hipEventCreateWithFlags(&st.ev_back[k], hipEventDisableTiming));
  }
  m.counts_pending = false;
  for (int k = 0; k < BACK_N; ++k) back_check(st, k, false);          // earlier calls whose words have landed: no wait
  // a row of more than 128 entries cannot go through the register-resident single pass: known from the list, no launch needed to find out
  if (m.max_list_row > 128) return false;
  const bool one_chunk = m.max_list_row >= 0 && m.max_list_row <= 64;
  // a grid the size of the residency (more workgroups would only queue)
  const int nblocks = std::max(1, std::min(nunits, st.ncu * st.occ[one_chunk ? 0 : 1] - m.reserve_wgs));
  // Unit schedule (see k_build_edges): the fixed stride needs the whole grid resident.  That holds for one model evaluating alone;
  // it is not a given when other kernels share the chip -- the overlapped multi-rank schedule (reserve_wgs > 0: exchange kernels and
  // the persistent model kernel of the other range run beside this one), a second model in the process, other processes on the
  // device (option edge_schedule=dynamic) -- and then units are claimed one at a time, which cannot wait on a workgroup that has not
  // started.  Launches with the fixed stride are additionally chained by an event, so that two of them never overlap in one process.
  const bool dyn = m.opt_edge_schedule == "dynamic" || (m.opt_edge_schedule == "auto" && (m.reserve_wgs > 0 || g_models_alive.load() > 1));
  // (one event per model, i.e. per device: ADVICE r03 -- a process-wide one belonged to whichever device ran first; with a second model alive the
  // dynamic schedule is chosen anyway)
  if (!dyn) {
    if (!st.chain) AHIP_CHECK(hipEventCreateWithFlags(&st.chain, hipEventDisableTiming));
    else AHIP_CHECK(hipStreamWaitEvent(a.stream, st.chain, 0));
  }
  // header: [0] ticket (u32), [1] maxdeg, [2] overflow, [3] number of heavy centres, [4] edge total; status array starts at byte 64
  const size_t bytes = 64 + (size_t)nunits * sizeof(unsigned long long);
  st.flags.reserve(bytes);
  const size_t cap = (size_t)std::max<long long>(m.nneigh, 1);            // upper bound: every list entry survives
  m.b_eoff.reserve((size_t)(inum + 2) * sizeof(int));
  m.b_eii.reserve(cap * sizeof(int));
  m.b_ej.reserve(cap * sizeof(int));
  m.b_rvec.reserve(cap * 3 * sizeof(float));
  m.b_ett.reserve(cap);
  m.edges_T_size = 4;
  int *hdr = st.flags.as<int>();
  m.tiles_packed = false;
  if (m.pack_slots > 0) {                              // tile packing for the fused kernel that will run, done by this kernel (see k_build_edges)
    m.b_tile_a0.reserve((size_t)(inum + 2) * sizeof(int));
    m.b_tile_e0.reserve((size_t)(inum + 2) * sizeof(int));
    m.b_centre.reserve((size_t)std::max(inum, 1) * sizeof(int2));
    m.b_ntiles.reserve(64);
  }
  if (m.heavy_thresh > 0) st.heavy.reserve((size_t)std::max(inum, 1) * sizeof(int));
  const int nall = std::max(m.nall, 1);
  st.xt.reserve((size_t)nall * sizeof(AtomXT));
  const int nclear = (int)(bytes / 8);
  hipLaunchKernelGGL(k_pack_xt, dim3((std::max(nall, nclear) + 255) / 256), dim3(256), 0, a.stream, m.nall, a.x, a.ftype, a.mtype, (AtomXT *)st.xt.p,
                     (unsigned long long *)st.flags.p, nclear, a.engvir);
#define EB_LAUNCH(CH, DY) hipLaunchKernelGGL((k_build_edges<CH, DY>), dim3(nblocks), dim3(EB_THREADS), 0, a.stream, inum, m.d_ilist, m.d_nloff, m.d_nlj,       \
                     (const AtomXT *)st.xt.p, a.cutsq, a.nft, nunits, (unsigned int *)hdr, (unsigned long long *)((char *)st.flags.p + 64),   \
                     m.b_eoff.as<int>(), m.b_eii.as<int>(), m.b_ej.as<int>(), m.b_rvec.as<float>(), hdr + 1, hdr + 2,                \
                     m.b_ett.as<unsigned char>(), m.heavy_thresh, hdr + 3, st.heavy.as<int>(), hdr + 4,                              \
                     m.pack_slots, m.pack_maxa, m.b_tile_a0.as<int>(), m.b_tile_e0.as<int>(), m.b_ntiles.as<int>(), m.b_centre.as<int2>())
  if (dyn) { if (one_chunk) EB_LAUNCH(1, true); else EB_LAUNCH(2, true); }
  else {
    if (one_chunk) EB_LAUNCH(1, false); else EB_LAUNCH(2, false);
    AHIP_CHECK(hipEventRecord(st.chain, a.stream));
  }
#undef EB_LAUNCH
  AHIP_CHECK(hipGetLastError());
  // The counters go to pinned memory asynchronously.  The reference's Kokkos path reads its edge total back in every step
  // (pair_nequip_allegro_kokkos.cpp:203-206); here nobody waits for the copy unless a value is needed on the host: with every row <= 128
  // entries (known since the list was installed) no row can overflow, the degree is bounded by the row length, and the kernels that follow
  // read the totals from device memory.
  st.cur = (st.cur + 1) % BACK_N;
  back_check(st, st.cur, true);                     // the slot of BACK_N calls ago: long complete
  int *h3 = st.h_back + 8 * st.cur;
  AHIP_CHECK(hipMemcpyAsync(h3, hdr, 5 * sizeof(int), hipMemcpyDeviceToHost, a.stream));       // ticket, max degree, overflow, heavy centres, edge total
  AHIP_CHECK(hipEventRecord(st.ev_back[st.cur], a.stream));
  st.unchecked[st.cur] = true;
  m.d_maxdeg = hdr + 1;
  m.tiles_packed = m.pack_slots > 0;
  m.have_ett = true;
  m.counts_pending = true;
  if (m.max_list_row < 0) {                         // row lengths unknown (cannot happen through the C-ABI, which measures them at every list hand-over): check the overflow flag now
    AHIP_CHECK(hipEventSynchronize(st.ev_back[st.cur]));
    st.unchecked[st.cur] = false;                   // handled here: the caller falls back to the two-pass build
    if (h3[2] != 0) { m.counts_pending = false; m.have_ett = false; return false; }
    edges_counts(m);
  }
  return true;
}

void edges_free(Model &m) {
  if (!m.edge_state) return;
  EdgeState *st = (EdgeState *)m.edge_state;
  st->flags.release();
  st->xt.release();
  st->heavy.release();
  st->hoff.release();
  if (st->h_back) (void)hipHostFree(st->h_back);
  for (int k = 0; k < BACK_N; ++k) if (st->ev_back[k]) (void)hipEventDestroy(st->ev_back[k]);
  if (st->chain) (void)hipEventDestroy(st->chain);
  delete st;
  m.edge_state = nullptr;
}

}  // namespace ahip
