// Device-wide primitives: single-pass decoupled scan is overkill for int32 counts of <= a few
// million atoms; a 3-kernel block scan (reduce, scan of block sums, down-sweep) keeps it simple,
// deterministic and graph-capturable (no temp-storage queries, no host sync).
#include "prims.h"

namespace ahip {

static constexpr int SCAN_BLOCK = 256;
static constexpr int SCAN_ITEMS = 8;                    // per thread
static constexpr int SCAN_TILE = SCAN_BLOCK * SCAN_ITEMS;

__device__ inline int wave_incl_scan(int v, int lane) {
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    int t = __shfl_up(v, off, 64);
    if (lane >= off) v += t;
  }
  return v;
}

// block-wide exclusive scan of one value per thread; returns exclusive prefix, total in *tot
__device__ inline int block_excl_scan(int v, int *tot) {
  __shared__ int wsum[SCAN_BLOCK / 64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int inc = wave_incl_scan(v, lane);
  if (lane == 63) wsum[w] = inc;
  __syncthreads();
  int base = 0, total = 0;
#pragma unroll
  for (int k = 0; k < SCAN_BLOCK / 64; ++k) {
    if (k < w) base += wsum[k];
    total += wsum[k];
  }
  __syncthreads();
  *tot = total;
  return base + inc - v;
}

__global__ void __launch_bounds__(SCAN_BLOCK) k_scan_tile_sums(const int *in, int n, int *tile_sums) {
  const long long base = (long long)blockIdx.x * SCAN_TILE;
  int s = 0;
  for (int k = 0; k < SCAN_ITEMS; ++k) {
    long long i = base + (long long)k * SCAN_BLOCK + threadIdx.x;
    if (i < n) s += in[i];
  }
  int tot;
  block_excl_scan(s, &tot);
  if (threadIdx.x == 0) tile_sums[blockIdx.x] = tot;
}

// one block scans the tile sums in place (exclusive) and writes the grand total to out_total
__global__ void __launch_bounds__(SCAN_BLOCK) k_scan_tiles(int *tile_sums, int ntiles, int *out_total) {
  int carry = 0;
  for (int b = 0; b < ntiles; b += SCAN_BLOCK) {
    int i = b + threadIdx.x;
    int v = i < ntiles ? tile_sums[i] : 0;
    int tot;
    int ex = block_excl_scan(v, &tot);
    if (i < ntiles) tile_sums[i] = carry + ex;
    carry += tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) *out_total = carry;
}

__global__ void __launch_bounds__(SCAN_BLOCK) k_scan_down(const int *in, int n, const int *tile_sums, int *out) {
  // thread owns SCAN_ITEMS consecutive items so that the block scan runs over per-thread sums
  const long long base = (long long)blockIdx.x * SCAN_TILE + (long long)threadIdx.x * SCAN_ITEMS;
  int v[SCAN_ITEMS];
  int s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) {
    long long i = base + k;
    v[k] = i < n ? in[i] : 0;
    s += v[k];
  }
  int tot;
  int ex = block_excl_scan(s, &tot) + tile_sums[blockIdx.x];
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) {
    long long i = base + k;
    if (i < n) out[i] = ex;
    ex += v[k];
  }
}

hipError_t prim_exclusive_scan_i32(PrimScratch &ps, const int *in, int *out, int n, hipStream_t s) {
  if (n <= 0) return hipMemsetAsync(out, 0, sizeof(int), s);
  int ntiles = (n + SCAN_TILE - 1) / SCAN_TILE;
  if ((size_t)ntiles > ps.tile_cap) {
    // growing: earlier scans of this model on this stream may still read the old buffer
    hipError_t e = hipStreamSynchronize(s);
    if (e != hipSuccess) return e;
    if (ps.tile) (void)hipFree(ps.tile);
    ps.tile = nullptr;
    ps.tile_cap = (size_t)ntiles * 2 + 1024;
    e = hipMalloc((void **)&ps.tile, ps.tile_cap * sizeof(int));
    if (e != hipSuccess) { ps.tile = nullptr; ps.tile_cap = 0; return e; }
  }
  int *g_tile_buf = ps.tile;
  hipLaunchKernelGGL(k_scan_tile_sums, dim3(ntiles), dim3(SCAN_BLOCK), 0, s, in, n, g_tile_buf);
  hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(SCAN_BLOCK), 0, s, g_tile_buf, ntiles, out + n);
  hipLaunchKernelGGL(k_scan_down, dim3(ntiles), dim3(SCAN_BLOCK), 0, s, in, n, g_tile_buf, out);
  return hipGetLastError();
}

// ---------------------------------------------------------------- column sums
static constexpr int RED_BLOCKS = 512;

__device__ inline double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

__global__ void __launch_bounds__(256) k_colsum_stage1(const double *in, long long nrow, int ncol, double *part) {
  __shared__ double sm[4][8];
  double acc[8];
  for (int c = 0; c < 8; ++c) acc[c] = 0;
  for (long long r = (long long)blockIdx.x * 256 + threadIdx.x; r < nrow; r += (long long)gridDim.x * 256)
    for (int c = 0; c < ncol; ++c) acc[c] += in[r * ncol + c];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int c = 0; c < ncol; ++c) {
    double v = wave_sum(acc[c]);
    if (lane == 0) sm[w][c] = v;
  }
  __syncthreads();
  if (threadIdx.x < ncol) part[blockIdx.x * 8 + threadIdx.x] = sm[0][threadIdx.x] + sm[1][threadIdx.x] + sm[2][threadIdx.x] + sm[3][threadIdx.x];
}

__global__ void __launch_bounds__(64) k_colsum_stage2(const double *part, int nblocks, int ncol, double *out) {
  for (int c = 0; c < ncol; ++c) {
    double v = 0;
    for (int b = threadIdx.x; b < nblocks; b += 64) v += part[b * 8 + c];
    v = wave_sum(v);
    if (threadIdx.x == 0) out[c] = v;
  }
}

hipError_t prim_sum_columns_f64(PrimScratch &ps, const double *in, long long nrow, int ncol, double *out, hipStream_t s) {
  if (!ps.part) {
    hipError_t e = hipMalloc((void **)&ps.part, RED_BLOCKS * 8 * sizeof(double));
    if (e != hipSuccess) { ps.part = nullptr; return e; }
  }
  double *g_part = ps.part;
  int nb = (int)((nrow + 255) / 256);
  if (nb > RED_BLOCKS) nb = RED_BLOCKS;
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL(k_colsum_stage1, dim3(nb), dim3(256), 0, s, in, nrow, ncol, g_part);
  hipLaunchKernelGGL(k_colsum_stage2, dim3(1), dim3(64), 0, s, g_part, nb, ncol, out);
  return hipGetLastError();
}

__global__ void __launch_bounds__(256) k_max_i32(const int *in, int n, int *out) {
  int m = 0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) m = max(m, in[i]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = max(m, __shfl_down(m, off, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(out, m);
}

hipError_t prim_max_i32(const int *in, int n, int *out, hipStream_t s) {
  hipError_t e = hipMemsetAsync(out, 0, sizeof(int), s);
  if (e != hipSuccess) return e;
  if (n <= 0) return hipSuccess;
  int nb = (n + 255) / 256;
  if (nb > 1024) nb = 1024;
  hipLaunchKernelGGL(k_max_i32, dim3(nb), dim3(256), 0, s, in, n, out);
  return hipGetLastError();
}

__global__ void __launch_bounds__(256) k_disp_max(const double *x, const double *xh, const double *v, int n, unsigned int *work) {
  float md = 0.f, mv = 0.f;                  // squared lengths; non-negative floats order like their bit patterns
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const double dx = x[3 * i] - xh[3 * i], dy = x[3 * i + 1] - xh[3 * i + 1], dz = x[3 * i + 2] - xh[3 * i + 2];
    const double vx = v[3 * i], vy = v[3 * i + 1], vz = v[3 * i + 2];
    md = fmaxf(md, (float)(dx * dx + dy * dy + dz * dz) * 1.000001f);      // rounded up: the criterion must not miss
    mv = fmaxf(mv, (float)(vx * vx + vy * vy + vz * vz) * 1.000001f);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { md = fmaxf(md, __shfl_down(md, off, 64)); mv = fmaxf(mv, __shfl_down(mv, off, 64)); }
  if ((threadIdx.x & 63) == 0) { atomicMax(&work[0], __float_as_uint(md)); atomicMax(&work[1], __float_as_uint(mv)); }
}
__global__ void k_disp_flag(unsigned int *work, double dt, double half_skin, int *flag) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const double reach = sqrt((double)__uint_as_float(work[0])) + 2.0 * dt * sqrt((double)__uint_as_float(work[1]));
    flag[0] = reach > half_skin ? 1 : 0;
    work[0] = 0u; work[1] = 0u;             // ready for the next step's maxima: no memset launch per step (the caller zeroes the words once)
  }
}
hipError_t prim_reneighbor_flag(const double *x, const double *xhold, const double *v, int n, double dt, double half_skin,
                                unsigned int *work, int *flag, hipStream_t s) {
  if (n > 0) {
    int nb = (n + 255) / 256;
    if (nb > 1024) nb = 1024;
    hipLaunchKernelGGL(k_disp_max, dim3(nb), dim3(256), 0, s, x, xhold, v, n, work);
  }
  hipLaunchKernelGGL(k_disp_flag, dim3(1), dim3(64), 0, s, work, dt, half_skin, flag);
  return hipGetLastError();
}

}  // namespace ahip
