// TEST INFRASTRUCTURE ONLY -- never part of liballegro_hip.so.
//
// Minimal single-threaded host stand-in for <hip/hip_runtime.h>, just enough to compile the
// *sync-free* HIP sources (generic kernels, API orchestration, neighbor builder) with g++ so that
// the `-m "not gpu"` tests can run the kernel LOGIC on the CPU under sanitizers, against the torch
// oracle, in a container without a GPU.  Kernels that use LDS, cross-lane ops or MFMA (prims.hip,
// fused.hip) are not compiled here; emu_parts.cpp provides loop equivalents of the two primitives.
// The product library is built only by hipcc for gfx950 and has no CPU path.
#pragma once
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <cstddef>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
extern dim3 threadIdx, blockIdx, blockDim, gridDim;

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorUnknown = 999 };
typedef struct ihipStream_t *hipStream_t;
struct EmuEvent { std::chrono::steady_clock::time_point t; };
typedef EmuEvent *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyHostToHost };

inline const char *hipGetErrorString(hipError_t) { return "emu error"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipMalloc(void **p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorUnknown; }
template <typename T> inline hipError_t hipMalloc(T **p, size_t n) { return hipMalloc((void **)p, n); }
inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
#define hipHostMallocDefault 0
#define hipHostMallocMapped 2
#define hipHostMallocPortable 1
inline hipError_t hipHostMalloc(void **p, size_t n, unsigned) { return hipMalloc(p, n); }
inline hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
inline hipError_t hipHostGetDevicePointer(void **d, void *h, unsigned) { *d = h; return hipSuccess; }
inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { if (n) memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind k, hipStream_t) { return hipMemcpy(d, s, n, k); }
inline hipError_t hipMemset(void *d, int v, size_t n) { if (n) memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { return hipMemset(d, v, n); }
inline hipError_t hipEventCreate(hipEvent_t *e) { *e = new EmuEvent(); return hipSuccess; }
#define hipEventDisableTiming 2
inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = new EmuEvent(); return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) {
  *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
  return hipSuccess;
}

template <typename T> inline T atomicAdd(T *p, T v) { T o = *p; *p = o + v; return o; }
template <typename T> inline T atomicMax(T *p, T v) { T o = *p; if (v > o) *p = v; return o; }

// sequential launch: valid for kernels without intra-block synchronisation
#define hipLaunchKernelGGL(k, g, b, sh, st, ...)                                   \
  do {                                                                             \
    dim3 _g = (g), _b = (b);                                                       \
    gridDim = _g; blockDim = _b;                                                   \
    for (unsigned _bx = 0; _bx < _g.x; ++_bx)                                      \
      for (unsigned _tx = 0; _tx < _b.x; ++_tx) {                                  \
        blockIdx.x = _bx; threadIdx.x = _tx;                                       \
        k(__VA_ARGS__);                                                            \
      }                                                                            \
  } while (0)
