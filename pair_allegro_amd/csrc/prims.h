// Device-wide primitives used by the engine (exclusive scan, column sums).
#pragma once
#include <hip/hip_runtime.h>

namespace ahip {
// Scratch of the primitives: owned by ONE model (engine.h: Model::prim), so two models driven on different streams, devices or
// host threads never share it.  Calls on one model are ordered on that model's stream.
struct PrimScratch {
  int *tile = nullptr;          // scan: per-tile sums
  size_t tile_cap = 0;
  double *part = nullptr;       // column sums: stage-1 partials
  void release() {
    if (tile) (void)hipFree(tile);
    if (part) (void)hipFree(part);
    tile = nullptr; tile_cap = 0; part = nullptr;
  }
};
// out[0..n] = exclusive prefix sum of in[0..n), out[n] = total.  (prefix sum of
// /root/reference/pair_nequip_allegro.cpp:515-519; Kokkos K2 of pair_nequip_allegro_kokkos.cpp:196-202)
hipError_t prim_exclusive_scan_i32(PrimScratch &ps, const int *in, int *out, int n, hipStream_t s);
// out[c] = sum_r in[r*ncol + c], r < nrow  (deterministic two-stage tree), ncol <= 8.
hipError_t prim_sum_columns_f64(PrimScratch &ps, const double *in, long long nrow, int ncol, double *out, hipStream_t s);
// Re-neighboring criterion of the stand-alone MD driver: flag[0] = (max_i |x_i - xhold_i| + 2 dt max_i |v_i| > half_skin) ? 1 : 0
// (f64 positions / velocities [n][3]); work[2] is an unsigned scratch pair owned by the caller (the two maxima as float bits).
hipError_t prim_reneighbor_flag(const double *x, const double *xhold, const double *v, int n, double dt, double half_skin,
                                unsigned int *work, int *flag, hipStream_t s);
// max over in[0..n) -> out[0]
hipError_t prim_max_i32(const int *in, int n, int *out, hipStream_t s);
}  // namespace ahip
