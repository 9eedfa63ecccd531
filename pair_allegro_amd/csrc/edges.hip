// Single-pass edge build for the float32 production path: cutoff filter of the skin-inflated full
// neighbor list + CSR->COO expansion (reference: preprocess() passes 1+2, /root/reference/
// pair_nequip_allegro.cpp:488-512,566-629; Kokkos K1,K2,K5, pair_nequip_allegro_kokkos.cpp:165-258).
//
//  * one wave per centre row: the row's neighbour indices are read coalesced (64 per instruction), the
//    gathered x_j / type_j are the only random traffic; `rsq <= cut^2` in float64 (host-path semantics);
//  * survivors are compacted with ballot/mbcnt and kept in registers (8 centres per wave);
//  * the global edge offsets come from a decoupled look-back scan over 32-centre blocks (block order taken
//    from an atomic ticket, so a block's predecessors have always started), so the neighbour data is
//    gathered ONCE -- the two-pass version (generic_kernels.h) gathers it twice;
//  * every block then writes its edges to one contiguous range of e_ii / e_j / rvec.
// Output layout is identical to k_count_edges + scan + k_fill_edges: edges grouped by centre, list order.
// Rows longer than 128 entries raise the overflow flag and the caller re-runs the two-pass kernels.
#include <hip/hip_runtime.h>

#include "engine.h"

namespace ahip {

static constexpr int EB_ATOMS = 32;      // centres per scan unit (one status word each)
// The kernel is PERSISTENT: a grid of resident workgroups (<= 4 per CU, below the 5 its registers admit; 2 per CU for the
// two-chunk instance) each takes one ticket and then handles the scan units ticket, ticket + G, ticket + 2 G, ...  One ticket per
// 32 centres cost 0.35 ms at 1 M atoms (a single device-scope word takes ~88 atomics per microsecond).  The look-back of unit u
// waits for unit u - 1 = the previous ticket in the same round (or the last ticket of the previous round): all G workgroups are
// resident, so every predecessor is running -- a grid larger than the residency would deadlock here.
static constexpr int EB_PER_WAVE = 8;    // centres per wave
// EB_CHUNKS (template parameter): 64-entry chunks of a list row held in registers: 1 when no row of the installed list is longer
// than 64 entries (half the registers: 5 instead of 3 waves per SIMD in flight for this latency-bound gather), else 2

// {x, y, z, types} per atom, 32-byte aligned: ONE sector per gathered neighbour instead of two or three (the 24-byte position
// straddles sectors, the type sits in another array).  Rewritten every step by k_pack_xt (a 60 MB stream at 1 M atoms).
struct __attribute__((aligned(32))) AtomXT { double x, y, z; int ft, mt; };

__global__ void __launch_bounds__(256) k_pack_xt(int nall, const double *__restrict__ x, const int *__restrict__ ftype,
                                                  const int *__restrict__ mtype, AtomXT *__restrict__ xt) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= nall) return;
  AtomXT a;
  a.x = x[3 * (size_t)i]; a.y = x[3 * (size_t)i + 1]; a.z = x[3 * (size_t)i + 2];
  a.ft = ftype[i]; a.mt = mtype[i];
  xt[i] = a;
}

__device__ __forceinline__ unsigned long long pack_state(unsigned long long state, unsigned long long v) { return (state << 62) | v; }

template <int EB_CHUNKS>
__global__ void __launch_bounds__(256) k_build_edges(int inum, const int *__restrict__ ilist, const int *__restrict__ nl_off,
                                                       const int *__restrict__ nl_j, const AtomXT *__restrict__ xt,
                                                       const double *__restrict__ cutsq, int nft, int nunits,
                                                       unsigned int *ticket, unsigned long long *status, int *eoff, int *e_ii,
                                                       int *e_j, float *rvec, int *maxdeg, int *overflow,
                                                       unsigned char *e_tt, int heavy_thresh,
                                                       int *heavy_cnt, int *heavy_list) {
  __shared__ int s_cnt[EB_ATOMS];
  __shared__ int s_base[EB_ATOMS + 1];
  __shared__ int s_blk;
  __shared__ long long s_prefix;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) s_blk = (int)atomicAdd(ticket, 1u);
  __syncthreads();
  const int b0 = s_blk;
  for (int b = b0; b < nunits; b += (int)gridDim.x) {
  const int a_begin = b * EB_ATOMS;

  // ---- gather + filter: results stay in registers -------------------------------------------------
  int jj[EB_PER_WAVE][EB_CHUNKS];
  float dxs[EB_PER_WAVE][EB_CHUNKS], dys[EB_PER_WAVE][EB_CHUNKS], dzs[EB_PER_WAVE][EB_CHUNKS];
  int rank[EB_PER_WAVE][EB_CHUNKS];      // -1 = dropped, else position inside the centre's edge range
  int tts[EB_PER_WAVE][EB_CHUNKS];       // (model type of centre) << 4 | (model type of neighbour), for the fused kernel
  int kept_k[EB_PER_WAVE];
#pragma unroll
  for (int k = 0; k < EB_PER_WAVE; ++k) {
    const int la = wave * EB_PER_WAVE + k;
    const int ii = a_begin + la;
    int kept = 0;
    if (ii < inum) {
      const int i = ilist[ii];
      const AtomXT ci = xt[i];
      const double xi = ci.x, yi = ci.y, zi = ci.z;
      const double *crow = cutsq + (size_t)ci.ft * nft;
      const int mti = ci.mt;
      const int p0 = nl_off[ii], p1 = nl_off[ii + 1];
      if (p1 - p0 > 64 * EB_CHUNKS && lane == 0) atomicOr(overflow, 1);
#pragma unroll
      for (int c = 0; c < EB_CHUNKS; ++c) {
        const int p = p0 + c * 64 + lane;
        const bool valid = p < p1;
        int j = 0, tt = 0;
        bool keep = false;
        float fx = 0.f, fy = 0.f, fz = 0.f;
        if (valid) {
          j = nl_j[p];
          const AtomXT cj = xt[j];                              // two 16-byte loads of one 32-byte sector
          const double ddx = cj.x - xi, ddy = cj.y - yi, ddz = cj.z - zi;
          const double rsq = ddx * ddx + ddy * ddy + ddz * ddz;
          keep = rsq <= crow[cj.ft];
          tt = (mti << 4) | cj.mt;
          fx = (float)ddx; fy = (float)ddy; fz = (float)ddz;       // neighbour - centre, f64 difference cast to f32
        }
        const unsigned long long mask = __ballot(keep);
        const int below = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
        jj[k][c] = j; dxs[k][c] = fx; dys[k][c] = fy; dzs[k][c] = fz; tts[k][c] = tt;
        rank[k][c] = keep ? kept + below : -1;
        kept += __popcll(mask);
      }
    } else {
#pragma unroll
      for (int c = 0; c < EB_CHUNKS; ++c) { jj[k][c] = 0; dxs[k][c] = dys[k][c] = dzs[k][c] = 0.f; rank[k][c] = -1; tts[k][c] = 0; }
    }
    kept_k[k] = kept;
    if (lane == 0) {
      s_cnt[la] = kept;
      // centres with more edges than a tile of the wide fused kernel holds: listed for the layer-at-a-time kernels
      if (heavy_thresh > 0 && kept > heavy_thresh) heavy_list[atomicAdd(heavy_cnt, 1)] = ii;
    }
  }
  __syncthreads();

  // ---- block scan of the 32 counts, then decoupled look-back for the block's global offset ----------
  if (tid < 64) {
    int v = tid < EB_ATOMS ? s_cnt[tid] : 0;
    int inc = v;
#pragma unroll
    for (int off = 1; off < 32; off <<= 1) {
      int t = __shfl_up(inc, off, 64);
      if (lane >= off) inc += t;
    }
    if (tid < EB_ATOMS) s_base[tid + 1] = inc;
    if (tid == 0) s_base[0] = 0;
    int mx = v;
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) mx = max(mx, __shfl_xor(mx, off, 64));
    if (tid == 0 && mx > __hip_atomic_load(maxdeg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(maxdeg, mx);
  }
  __syncthreads();
  if (tid < 64) {
    // decoupled look-back, one WAVE wide: lane l polls predecessor b-1-l; the window closes at the nearest predecessor that has
    // published its inclusive prefix (state 2), everything nearer contributes its aggregate (state 1).  (A one-lane walk costs
    // one dependent global load per predecessor: with ~1000 blocks resident the chain was the kernel's critical path.)
    const unsigned long long agg = (unsigned long long)s_base[EB_ATOMS];
    const unsigned long long VMASK = (1ull << 62) - 1;
    unsigned long long prefix = 0;
    if (b == 0) {
      if (lane == 0) __hip_atomic_store(&status[0], pack_state(2, agg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      if (lane == 0) __hip_atomic_store(&status[b], pack_state(1, agg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int base = b - 1;
      for (;;) {
        const int k = base - lane;
        const unsigned long long v = k >= 0 ? __hip_atomic_load(&status[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : pack_state(2, 0);
        const unsigned st2 = (unsigned)(v >> 62);
        const unsigned long long ready = __ballot(st2 != 0), incl = __ballot(st2 == 2);
        const int first2 = incl ? __builtin_ctzll(incl) : 64;                 // nearest predecessor with an inclusive prefix
        const unsigned long long need = first2 < 63 ? ((2ull << first2) - 1) : ~0ull;
        if ((ready & need) == need) {
          unsigned long long c = lane <= first2 ? (v & VMASK) : 0ull;
#pragma unroll
          for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
          prefix += c;
          if (first2 < 64) break;
          base -= 64;
        } else __builtin_amdgcn_s_sleep(1);
      }
      if (lane == 0) __hip_atomic_store(&status[b], pack_state(2, prefix + agg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (lane == 0) s_prefix = (long long)prefix;
  }
  __syncthreads();
  const long long gbase = s_prefix;

  // ---- offsets + edges -------------------------------------------------------------------------------
  if (tid < EB_ATOMS && a_begin + tid < inum) eoff[a_begin + tid] = (int)(gbase + s_base[tid]);
  if (tid == 0 && a_begin + EB_ATOMS >= inum) eoff[inum] = (int)(gbase + s_base[min(EB_ATOMS, inum - a_begin)]);
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
  __syncthreads();                       // the LDS counters are reused by the next unit
  }
}

struct EdgeState { DevBuf flags, heavy, hoff, xt; };

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
  int ncu = 256;
  { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, m.device) == hipSuccess && prop.multiProcessorCount > 0) ncu = prop.multiProcessorCount; }
  const bool one_chunk = m.max_list_row >= 0 && m.max_list_row <= 64;
  const int nblocks = std::min(nunits, ncu * (one_chunk ? 4 : 2));          // resident grid, see k_build_edges
  if (!m.edge_state) m.edge_state = new EdgeState();
  EdgeState &st = *(EdgeState *)m.edge_state;
  // header: [0] ticket (u32), [1] maxdeg, [2] overflow, [3] number of heavy centres; status array starts at byte 64
  const size_t bytes = 64 + (size_t)nunits * sizeof(unsigned long long);
  st.flags.reserve(bytes);
  AHIP_CHECK(hipMemsetAsync(st.flags.p, 0, bytes, a.stream));
  const size_t cap = (size_t)std::max<long long>(m.nneigh, 1);            // upper bound: every list entry survives
  m.b_eoff.reserve((size_t)(inum + 2) * sizeof(int));
  m.b_eii.reserve(cap * sizeof(int));
  m.b_ej.reserve(cap * sizeof(int));
  m.b_rvec.reserve(cap * 3 * sizeof(float));
  m.b_ett.reserve(cap);
  m.edges_T_size = 4;
  int *hdr = st.flags.as<int>();
  if (m.heavy_thresh > 0) st.heavy.reserve((size_t)std::max(inum, 1) * sizeof(int));
  const int nall = std::max(m.nall, 1);
  st.xt.reserve((size_t)nall * sizeof(AtomXT));
  hipLaunchKernelGGL(k_pack_xt, dim3((nall + 255) / 256), dim3(256), 0, a.stream, m.nall, a.x, a.ftype, a.mtype, (AtomXT *)st.xt.p);
#define EB_LAUNCH(CH) hipLaunchKernelGGL(k_build_edges<CH>, dim3(nblocks), dim3(256), 0, a.stream, inum, m.d_ilist, m.d_nloff, m.d_nlj,       \
                     (const AtomXT *)st.xt.p, a.cutsq, a.nft, nunits, (unsigned int *)hdr, (unsigned long long *)((char *)st.flags.p + 64),   \
                     m.b_eoff.as<int>(), m.b_eii.as<int>(), m.b_ej.as<int>(), m.b_rvec.as<float>(), hdr + 1, hdr + 2,                \
                     m.b_ett.as<unsigned char>(), m.heavy_thresh, hdr + 3, st.heavy.as<int>())
  if (one_chunk) EB_LAUNCH(1); else EB_LAUNCH(2);
#undef EB_LAUNCH
  AHIP_CHECK(hipGetLastError());
  int h3[4] = {0, 0, 0, 0}, tot = 0;
  // the scalar read-back per step (the Kokkos path has the same one: pair_nequip_allegro_kokkos.cpp:203-206)
  AHIP_CHECK(hipMemcpyAsync(h3, hdr, 4 * sizeof(int), hipMemcpyDeviceToHost, a.stream));
  AHIP_CHECK(hipMemcpyAsync(&tot, m.b_eoff.as<int>() + inum, sizeof(int), hipMemcpyDeviceToHost, a.stream));
  AHIP_CHECK(hipStreamSynchronize(a.stream));
  if (h3[2] != 0) return false;                     // a row longer than 128 entries: caller uses the two-pass kernels
  m.nedges = tot;
  m.last_max_deg = h3[1];
  m.nheavy = m.heavy_thresh > 0 ? h3[3] : 0;
  m.have_ett = true;
  return true;
}

void edges_free(Model &m) {
  if (!m.edge_state) return;
  EdgeState *st = (EdgeState *)m.edge_state;
  st->flags.release();
  st->xt.release();
  st->heavy.release();
  st->hoff.release();
  delete st;
  m.edge_state = nullptr;
}

}  // namespace ahip
