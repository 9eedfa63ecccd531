// Device-wide primitives used by the engine (exclusive scan, column sums).
#pragma once
#include <hip/hip_runtime.h>

namespace ahip {
// out[0..n] = exclusive prefix sum of in[0..n), out[n] = total.  (prefix sum of
// /root/reference/pair_nequip_allegro.cpp:515-519; Kokkos K2 of pair_nequip_allegro_kokkos.cpp:196-202)
hipError_t prim_exclusive_scan_i32(const int *in, int *out, int n, hipStream_t s);
// out[c] = sum_r in[r*ncol + c], r < nrow  (deterministic two-stage tree), ncol <= 8.
hipError_t prim_sum_columns_f64(const double *in, long long nrow, int ncol, double *out, hipStream_t s);
// max over in[0..n) -> out[0]
hipError_t prim_max_i32(const int *in, int n, int *out, hipStream_t s);
}  // namespace ahip
