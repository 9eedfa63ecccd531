// TEST INFRASTRUCTURE ONLY: loop equivalents of prims.hip and a stub of fused.hip for the
// host-emulation build (see hip/hip_runtime.h).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>

#include "engine.h"
#include "prims.h"

dim3 threadIdx, blockIdx, blockDim, gridDim;

namespace ahip {
hipError_t prim_exclusive_scan_i32(PrimScratch &, const int *in, int *out, int n, hipStream_t) {
  long long s = 0;
  for (int i = 0; i < n; ++i) { int v = in[i]; out[i] = (int)s; s += v; }
  out[n > 0 ? n : 0] = (int)s;
  return hipSuccess;
}
hipError_t prim_sum_columns_f64(PrimScratch &, const double *in, long long nrow, int ncol, double *out, hipStream_t) {
  for (int c = 0; c < ncol; ++c) {
    double s = 0;
    for (long long r = 0; r < nrow; ++r) s += in[r * ncol + c];
    out[c] = s;
  }
  return hipSuccess;
}
hipError_t prim_max_i32(const int *in, int n, int *out, hipStream_t) {
  int m = 0;
  for (int i = 0; i < n; ++i) m = std::max(m, in[i]);
  *out = m;
  return hipSuccess;
}
hipError_t prim_reneighbor_flag(const double *x, const double *xh, const double *v, int n, double dt, double half_skin,
                                unsigned int *, int *flag, hipStream_t) {
  double md = 0, mv = 0;
  for (int i = 0; i < n; ++i) {
    double d = 0, w = 0;
    for (int k = 0; k < 3; ++k) { const double a = x[3 * i + k] - xh[3 * i + k]; d += a * a; w += v[3 * i + k] * v[3 * i + k]; }
    md = std::max(md, d); mv = std::max(mv, w);
  }
  flag[0] = std::sqrt(md) + 2.0 * dt * std::sqrt(mv) > half_skin ? 1 : 0;
  return hipSuccess;
}
bool fused_model_supported(const Model &, std::string *why) { if (why) *why = "host emulation has no MFMA"; return false; }
bool fused_run(Model &, const ComputeArgs &, std::string *why) { if (why) *why = "host emulation has no MFMA"; return false; }
void fused_free(Model &) {}
bool fusedlx_model_supported(const Model &, std::string *why) { if (why) *why = "host emulation has no MFMA"; return false; }
bool fusedlx_run(Model &, const ComputeArgs &, std::string *why) { if (why) *why = "host emulation has no MFMA"; return false; }
void fusedlx_free(Model &) {}
bool fusedlx2_run(Model &, const ComputeArgs &, std::string *why) { if (why) *why = "host emulation has no MFMA"; return false; }
void fusedlx2_free(Model &) {}
void edges_compact_heavy(Model &, const ComputeArgs &) {}
bool edges_build_f32(Model &, const ComputeArgs &) { return false; }   // emulation runs the two-pass kernels
void edges_free(Model &) {}
void edges_counts(Model &) {}
int edges_max_row(Model &, int, const int *) { return -1; }
bool gemm_f32(hipStream_t, long long, int, int, const float *, int, const float *, int, bool, float *, int, bool, float *, const float *) { return false; }
bool latent_update_bwd_f32(hipStream_t, long long, int, const float *, const float *, const float *, const float *, float *, float *, float *) { return false; }
bool embed_bwd_Y_f32(hipStream_t, long long, int, int, const float *, const float *, float *) { return false; }
bool env_bwd_Y_f32(hipStream_t, long long, int, int, const float *, const int *, int, const float *, float *) { return false; }
bool tp_fwd_f32(hipStream_t, long long, int, bool, int, const float *, const float *, const float *, const int *, int, float *) { return false; }
bool tp_bwd_f32(hipStream_t, long long, int, bool, int, const float *, const float *, const float *, const int *, int, const float *, float *, float *) { return false; }
}  // namespace ahip

extern "C" int ahip_debug_fused_linear(int, int, const double *, const float *, float *) { return 5; }
extern "C" int ahip_debug_fused_edges(void *, float *, long long) { return 5; }

// TEST INFRASTRUCTURE ONLY (tests/test_host_logic.py): replaces a freshly loaded model by its zero-padded copy (model_io.cpp: pad_host_model, what the fused kernels of the
// product run for models narrower than their fixed widths) BEFORE anything was evaluated, so that the layer-at-a-time kernels of the emulation evaluate the padded model.
extern "C" int ahip_emu_pad_model(void *mh, int SF, int UF, int WF, int RF) {
  ahip::Model *m = (ahip::Model *)mh;
  if (!m) return 1;
  try { m->hm = ahip::pad_host_model(m->hm, SF, UF, WF, RF); } catch (...) { return 2; }
  return 0;
}
