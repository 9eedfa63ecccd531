// Internal engine state shared by the C-ABI (allegro_hip.hip), the generic path
// (generic_engine.h) and the fused MFMA path (fused.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <map>
#include <stdexcept>
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <string>
#include <vector>

#include "model_io.h"
#include "prims.h"

namespace ahip {

struct HipError : std::runtime_error {
  explicit HipError(const std::string &m) : std::runtime_error(m) {}
};
struct ArgError : std::runtime_error {
  explicit ArgError(const std::string &m) : std::runtime_error(m) {}
};
struct StateError : std::runtime_error {
  explicit StateError(const std::string &m) : std::runtime_error(m) {}
};
struct UnsupportedError : std::runtime_error {
  explicit UnsupportedError(const std::string &m) : std::runtime_error(m) {}
};

#define AHIP_CHECK(expr)                                                                       \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess)                                                                      \
      throw ahip::HipError(std::string(#expr) + " failed: " + hipGetErrorString(_e) + " (" +   \
                           __FILE__ + ":" + std::to_string(__LINE__) + ")");                   \
  } while (0)

// grow-only device buffer
struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
  template <typename T> T *as() const { return (T *)p; }
  void reserve(size_t bytes) {
    if (bytes <= cap) return;
    if (p) AHIP_CHECK(hipFree(p));
    p = nullptr; cap = 0;
    size_t want = bytes + bytes / 16 + 256;          // 6 % head-room: shape hysteresis like the
    AHIP_CHECK(hipMalloc(&p, want));                   // Kokkos path (pair_nequip_allegro_kokkos.cpp:218-229)
    cap = want;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

// bump allocator over one DevBuf; first pass measures, second pass hands out pointers
struct Arena {
  char *base = nullptr;
  size_t off = 0;
  bool measuring = true;
  template <typename T> T *get(size_t n) {
    size_t bytes = (n * sizeof(T) + 255) & ~size_t(255);
    T *r = measuring ? nullptr : (T *)(base + off);
    off += bytes;
    return r;
  }
};

template <typename T> struct DeviceWeights {
  std::map<std::string, T *> w;
  std::vector<void *> owned;
  bool ready = false;
  T *get(const std::string &n) const {
    auto it = w.find(n);
    if (it == w.end()) throw StateError("device weight '" + n + "' missing");
    return it->second;
  }
};

extern std::atomic<int> g_models_alive;     // models of this process (allegro_hip.hip)

// Stage timing (option timing=1): HIP events on the launch stream around every stage of a call.  Nothing waits for them when they are recorded
// (round 4: reading them back at the end of every call was a full synchronisation per ahip_compute_dev*); the pairs queue up in a ring and are
// turned into milliseconds when the caller asks (ahip_get_timings: the sum and the number of launches per stage since the last call of it).
struct TimingSlot {
  std::string name;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ring;      // event pairs, reused round-robin
  size_t head = 0, tail = 0;                                // pairs [tail, head) are recorded and not yet read
  double sum_ms = 0; long long count = 0;                   // read so far, not yet reported
};
static constexpr size_t TIMING_RING = 1024;

struct Model {
  HostModel hm;
  int device = 0;
  bool counted = false;                     // included in g_models_alive
  int D = 1, Ka = 0;
  std::vector<double> rcut_model_host;      // [T*T]

  // options
  std::string opt_path = "auto";            // auto | fused | generic
  std::string opt_precision = "model";      // model | float64
  std::string opt_fused_tb = "table";       // table | mlp: two-body embedding of the fused kernel from the spline table or as an MLP
  std::string opt_fused_arith = "auto";     // auto | f32 | f16x2 | bf16x3 | tf32eq: arithmetic of the fused kernels' linears.  auto = f16x2 (float32-equivalent, fused_h.h); tf32eq iff the
                                            // model file sets allow_tf32; f32 once auto has degraded (below)
  int last_fused_arith = 0;                 // what the last fused evaluation used: 0 f32, 1 bf16x3, 2 tf32eq, 3 f16x2
  // fused_arith=auto must never be less robust than the reference's float32: a model the f16x2 split cannot carry -- a weight beyond float16's
  // range, a linear whose weights sit in float16's subnormals, an activation that overflows, a first evaluation that disagrees with the f32 instance -- runs on the
  // f32-input MFMA instance for the rest of this model's life instead of failing.  Only an EXPLICIT fused_arith=f16x2 still reports those as errors.
  bool arith_degraded = false;              // auto has fallen back to f32 (sticky)
  bool arith_checked = false;               // the first-evaluation self-check of auto's f16x2 against the f32 instance has run (allegro_hip.hip: run_model)
  int arith_force = -1;                     // self-check only: -1 none, 0 = resolve auto to f32 for this dispatch
  int arith_check_attempts = 0;
  std::string arith_note;                   // what auto decided and why, one line (ahip_arith_note)
  DevBuf b_chk;                             // self-check: two force arrays + the 2-word reduction
  // A model narrower than a fused kernel's fixed widths runs on it zero-padded (round 6; model_io.h: pad_host_model): S <= 64 scalars, MLP width <= 64, read-out
  // width <= 32, U <= 32 tensor features for l_max = 1, U <= 32 / <= 64 for l_max = 2.  hm stays the model as loaded (layer-at-a-time kernels, metadata).
  HostModel hm_fused;
  bool hm_fused_ready = false;
  long long chunk_edges = 2000000;
  int reserve_wgs = 0;                      // workgroup slots the persistent fused kernels leave free (for kernels of other streams)
  bool timing = false;
  std::string opt_tile_pack = "auto";       // auto | separate | fused (auto = fused up to 262 144 centres per call): tile packing inside the single-pass edge build where the tile shape is known up front, or always by the stand-alone kernels (A/B, tests)
  std::string opt_edge_schedule = "auto";   // auto | static | dynamic: unit schedule of the single-pass edge build (edges.hip)
  bool cutoff_strict = false;               // edge kept iff rsq < cut^2 (the KOKKOS reference path) instead of rsq <= cut^2 (the host path)

  // weights
  DeviceWeights<float> wf;
  DeviceWeights<double> wd;
  void *cg_dev = nullptr;                    // AhipCgEntry[ncg_full]
  int ncg_full = 0, ncg_scalar = 0;
  double *rcut_model_dev = nullptr;          // [T*T]

  // neighbor list (device)
  int inum = 0, nall = 0;
  long long nneigh = 0;
  bool have_list = false;
  int max_list_row = -1;                     // longest row of the installed list (-1: unknown, e.g. a caller-owned device list)
  const int *d_ilist = nullptr, *d_nloff = nullptr, *d_nlj = nullptr;   // current (owned or borrowed)
  DevBuf b_ilist, b_nloff, b_nlj;
  std::vector<int> h_flat_j, h_off32, h_ilist;

  // per-call host-path staging
  DevBuf b_x, b_ftype, b_mtype, b_f, b_eatom, b_engvir, b_cutsq, b_flagwork;
  std::vector<int> h_ftype, h_mtype;
  std::vector<double> h_x, h_f, h_eatom, h_cutsq_dev;      // h_cutsq_dev: what b_cutsq currently holds (device path); h_x: page-locked copy of the caller's positions
  std::map<const void *, std::pair<void *, size_t>> pinned;   // host vectors currently page-locked (allegro_hip.hip: pin_host)

  // edge list + workspace
  DevBuf b_cnt, b_eoff, b_eii, b_ej, b_rvec, b_partial, b_ws, b_misc;
  DevBuf b_ett;                              // per-edge packed model types (centre << 4 | neighbour), fused path only
  bool have_ett = false;                     // b_ett matches the current edge list (written by the single-pass edge build)
  long long nedges = 0;
  int edges_T_size = 0;                      // sizeof(T) of b_rvec contents
  std::vector<int> h_eoff;
  int last_max_deg = 0;
  std::string last_path;
  // The single-pass edge build (edges.hip) leaves its counters -- edge total, largest degree, number of heavy centres -- on the device and
  // copies them to pinned memory asynchronously; the host waits for that copy (edges_counts) only where it needs one of the values, never on
  // the fused path of a bounded list (no hipStreamSynchronize per ahip_compute_dev* call).
  bool counts_pending = false;               // nedges / last_max_deg / nheavy are not valid yet: call edges_counts()
  long long nedges_hint = 0;                 // edge total of the last read-back that was waited for, else an estimate (heuristics only)
  const int *d_maxdeg = nullptr;             // device word holding the largest degree of the current edge list (single-pass build)
  // tile packing done by the single-pass edge build (edges.hip): requested shape (0 = none), whether the current edge list carries it, its arrays
  int pack_slots = 0, pack_maxa = 0;
  bool tiles_packed = false;
  DevBuf b_tile_a0, b_tile_e0, b_centre, b_ntiles;
  const int *d_ntiles_last = nullptr;        // device word with the tile count of the last fused launch; last_tile_slots: its edge slots per tile (0: chosen on the device from the largest degree)
  int last_tile_slots = 0;

  // Centres with more edges than a tile of the wide fused kernel holds (fused_lx.hip): found by the edge build when
  // heavy_thresh > 0, evaluated by the layer-at-a-time kernels on a compact copy of their edges (hv_*)
  int heavy_thresh = 0, nheavy = 0;
  long long hv_nedges = 0;
  DevBuf hv_eoff, hv_eii, hv_ej, hv_rvec, hv_ilist, hv_engvir;

  // `compute allegro`: registered output names and their values from the last host-path compute
  std::vector<std::string> custom_names;
  std::map<std::string, std::vector<double>> custom_out;

  // timings
  std::vector<TimingSlot> slots;
  std::string timing_names;
  std::vector<double> timing_ms, timing_counts;

  // scratch of the device-wide primitives (scan, column sums): per model, see prims.h
  PrimScratch prim;

  // fused path private state (fused.hip: model S shape; fused_lx.hip: l_max = 2 shapes)
  std::vector<hipEvent_t> f_events;          // host-pointer path: one event per returned chunk of f (allegro_hip.hip: ahip_compute)
  int *h_alarm = nullptr;                    // page-locked, device-mapped word raised by the f16x2 instances of the fused kernels (fused_h.h): see alarm_word()
  void *fused_state = nullptr;
  void *fusedlx_state = nullptr;
  void *fusedlx2_state = nullptr;            // wave-pair version of the 64-feature shape (fused_lx2.hip)

  // neighbor builder state
  void *nb_state = nullptr;
  // single-pass edge build state (edges.hip)
  void *edge_state = nullptr;
};

// ---- stage timing ------------------------------------------------------------------------------
void timing_drain(Model &m, TimingSlot &t, size_t keep);          // allegro_hip.hip: reads pairs until at most `keep` are outstanding (waits for them)
struct StageTimer {
  Model &m; hipStream_t s; int idx = -1; hipEvent_t eb = nullptr;
  StageTimer(Model &m_, const char *name, hipStream_t s_) : m(m_), s(s_) {
    if (!m.timing) return;
    for (size_t i = 0; i < m.slots.size(); ++i) if (m.slots[i].name == name) idx = (int)i;
    if (idx < 0) { TimingSlot t; t.name = name; m.slots.push_back(t); idx = (int)m.slots.size() - 1; }
    TimingSlot &t = m.slots[idx];
    if (t.head - t.tail >= TIMING_RING) timing_drain(m, t, TIMING_RING / 2);
    const size_t k = t.head % TIMING_RING;
    if (k >= t.ring.size()) {
      std::pair<hipEvent_t, hipEvent_t> e{nullptr, nullptr};
      AHIP_CHECK(hipEventCreate(&e.first)); AHIP_CHECK(hipEventCreate(&e.second));
      t.ring.push_back(e);
    }
    eb = t.ring[k].second;
    AHIP_CHECK(hipEventRecord(t.ring[k].first, s));
  }
  ~StageTimer() { if (idx >= 0) { (void)hipEventRecord(eb, s); ++m.slots[idx].head; } }
};

// ---- generic path (generic_engine.h via allegro_hip.hip) -------------------------------
struct ComputeArgs {
  int nlocal, nghost;
  const double *x;          // device [nall][3]
  const int *ftype;         // device [nall], index into cutsq
  const double *cutsq;      // device [nft*nft]
  int nft;
  const int *mtype;         // device [nall] model types
  double *f;                // device [nall][3], accumulated
  double *eatom;            // device [nall] or null
  double *engvir;           // device [7]
  hipStream_t stream;
};

// ---- fused path entry points (fused.hip; the host-emulation test build links a stub) ---------
// Returns true when the fused MFMA kernels support this model shape at all.
bool fused_model_supported(const Model &m, std::string *why);
// Runs forward+backward for all centres; edges already built (b_eoff/b_eii/b_ej/b_rvec as float).
// Returns false (and sets *why) if this particular list cannot be handled (e.g. too many edges per atom).
bool fused_run(Model &m, const ComputeArgs &a, std::string *why);
void fused_free(Model &m);
// f16x2 arithmetic (fused_h.h): device address of the model's alarm word (allocated on first use); fused_poll_alarm throws StateError when a kernel has
// raised it -- the host-pointer call polls behind its own synchronisation, device-resident callers meet it at their next evaluation
int *alarm_word(Model &m);
bool alarm_take(Model &m);                  // true (and cleared) when a kernel has raised the word since the last call
void fused_poll_alarm(Model &m);            // alarm_take + the policy: auto -> degrade to f32 and report once; explicit f16x2 -> StateError
// the fused kernels' fixed widths and whether a model fits them (exactly, or narrower: then padded)
inline int fused_UF(const HostModel &h) { return h.l_max == 1 ? 32 : (h.U <= 32 ? 32 : 64); }
inline bool fused_widths_fit(const HostModel &h) { return h.S >= 1 && h.S <= 64 && h.mlp_width >= 1 && h.mlp_width <= 64 && h.readout_width >= 1 && h.readout_width <= 32 && h.U >= 1 && h.U <= (h.l_max == 1 ? 32 : 64); }
inline const HostModel &fused_host_model(Model &m) {
  const HostModel &h = m.hm;
  if (h.S == 64 && h.mlp_width == 64 && h.readout_width == 32 && h.U == fused_UF(h)) return h;
  if (!m.hm_fused_ready) { m.hm_fused = pad_host_model(h, 64, fused_UF(h), 64, 32); m.hm_fused_ready = true; }
  return m.hm_fused;
}
// option fused_arith as it applies (the environment variable of the A/B tools wins)
inline std::string arith_option(const Model &m) {
  const char *ar = std::getenv("AHIP_FUSED_ARITH");
  return ar ? std::string(ar) : m.opt_fused_arith;
}
// auto -> f16x2 unless it has degraded (or the self-check is running its f32 pass)
inline bool arith_auto_is_f16x2(const Model &m) { return !m.arith_degraded && m.arith_force != 0; }
// arithmetic of the wide fused kernels' linears from option fused_arith: 3 = f16x2 (auto, f16x2), 0 = f32-input MFMA (f32; the bf16 splits exist in k_fused only)
inline int lx_arith_of(const Model &m) {
  const std::string a = arith_option(m);
  return (a == "f16x2" || (a == "auto" && arith_auto_is_f16x2(m))) ? 3 : 0;
}
// thrown by a fused kernel's prepare step when fused_arith=auto meets a model the f16x2 split cannot carry; run_model degrades the model and dispatches again
struct ArithDegraded {
  std::string why;
};
// range findings of append_frag_h (fused_h.h) over a model's weight stream
enum { H_RANGE_OVERFLOW = 1, H_RANGE_TINY = 2 };
// H_RANGE_TINY over a whole model: any [K][N] block of a dense-layer tensor whose largest weight is non-zero and below 2^-10 (fused_lx2.hip lays its stream
// out in sub-blocks of the matrices and asks here instead of per block)
inline int model_tiny_linear(const HostModel &h) {
  for (const auto &kv : h.tensors) {
    const HostTensor &t = kv.second;
    const std::string &n = kv.first;
    const bool dense = n.find(".lat.w") != std::string::npos || n.find(".env") != std::string::npos || n.find(".mix") != std::string::npos || n == "emb.w" || n == "out.w0" ||
                       (n.rfind("tb.w", 0) == 0);
    if (!dense || t.shape.size() < 2) continue;
    const long long blk = (long long)t.shape[t.shape.size() - 2] * t.shape[t.shape.size() - 1], nb = blk > 0 ? t.numel() / blk : 0;
    for (long long b = 0; b < nb; ++b) {
      double wmax = 0.0;
      for (long long i = 0; i < blk; ++i) wmax = std::max(wmax, std::fabs(t.data[(size_t)(b * blk + i)]));
      if (wmax > 0.0 && wmax < 0x1p-10) return H_RANGE_TINY;
    }
  }
  return 0;
}
// what a prepare step does with them: explicit f16x2 -> the overflow is an error (the tiny linear is the caller's choice); auto -> ArithDegraded
inline void arith_range_verdict(const Model &m, int flags) {
  if (!flags) return;
  const char *what = (flags & H_RANGE_OVERFLOW) ? "a weight of this model exceeds float16's range" : "a linear of this model has all its weights below 2^-10 (float16 subnormal territory for the split)";
  if (arith_option(m) == "auto") throw ArithDegraded{what};
  if (flags & H_RANGE_OVERFLOW) throw UnsupportedError(std::string("fused_arith=f16x2: ") + what + "; use fused_arith=f32 (or auto)");
}
// f16x2 arithmetic: the backward pass is linear in its upstream gradient scale[type] / sqrt(avg_num_neighbors) and runs scaled by the power of two that brings that
// gradient into [0.5, 1) -- PER CENTRE TYPE since round 6 (the kernels derive it from scale[t_i] with frexp: energy scales that differ by orders of magnitude
// between species each get their own; one global power of two from the largest left the small species in float16's subnormals)
// the same three for the wide shapes (l_max = 2; fused_lx.hip)
bool fusedlx_model_supported(const Model &m, std::string *why);
bool fusedlx_run(Model &m, const ComputeArgs &a, std::string *why);
void fusedlx_free(Model &m);
// wave-pair kernel for l_max = 2, 64 tensor features (fused_lx2.hip): two waves per SIMD, each wave half of the channels
bool fusedlx2_run(Model &m, const ComputeArgs &a, std::string *why);
void fusedlx2_free(Model &m);

// ---- single-pass float32 edge build (edges.hip; the host-emulation build links a stub returning false) ----
// Fills m.nedges, m.last_max_deg, b_eoff/b_eii/b_ej/b_rvec exactly like build_edges<float>; false = a list
// row is too long for the register-resident version and the caller must run the two-pass kernels.
bool edges_build_f32(Model &m, const ComputeArgs &a);
// waits for the counters of the last edges_build_f32 if they are still in flight (m.counts_pending) and installs them in m
// Synchronous copies between PAGEABLE host memory and the device, staged through the library's own page-locked buffer.  Handed a pageable pointer, the
// HIP runtime pins the range on the fly for copies above 1 MiB and caches that pinning by address; in a long-lived process the heap hands pages back to the
// system and gets them back later (malloc trimming), the cached pinning then points at nothing, and the copy engine faults on a HOST address -- seen as an
// intermittent "Memory access fault by GPU ... on address 0x5555..." during the weight upload of the n-th model of a process (round 4, tests -m gpu:
// 4 of 10 full runs).  Page-locked memory of our own never takes that path.
void copy_h2d(void *dst_dev, const void *src_host, size_t bytes);
void copy_d2h(void *dst_host, const void *src_dev, size_t bytes);
void edges_counts(Model &m);
// longest row of a device-resident CSR list (list hand-over only: one small kernel + a 4-byte read-back)
int edges_max_row(Model &m, int inum, const int *offsets_dev);
void edges_free(Model &m);
// compact copy (m.hv_*) of the edges of the m.nheavy centres the last edges_build_f32 listed
void edges_compact_heavy(Model &m, const ComputeArgs &a);

// ---- float32 dense layers of the generic path on the matrix cores (gemm.hip; the host-emulation build links a stub
// returning false and keeps the one-thread-per-output kernels) ----
// C[e][n] (+)= sum_k A[e][k] * (transB ? W[n][k] : W[k][n])
bool gemm_f32(hipStream_t s, long long E, int K, int N, const float *A, int lda, const float *W, int ldw, bool transB, float *C,
              int ldc, bool accumulate, float *silu_out = nullptr, const float *dsilu_z = nullptr);
// wave-per-row versions of k_latent_update_bwd / k_embed_bwd_Y (same stubs in the emulation build)
bool latent_update_bwd_f32(hipStream_t s, long long E, int S, const float *dx, const float *u, const float *fc, const float *res,
                           float *du, float *dfc, float *dxprev);
bool embed_bwd_Y_f32(hipStream_t s, long long E, int D, int U, const float *dV, const float *w, float *dY);
bool env_bwd_Y_f32(hipStream_t s, long long E, int D, int U, const float *denv, const int *e_ii, int c0, const float *om, float *dY);
// tensor product with the CG table unrolled at compile time (l_max 1, 2); scalar_only = last layer (l3 = 0 outputs)
bool tp_fwd_f32(hipStream_t s, long long E, int L, bool scalar_only, int U, const float *pw, const float *V, const float *env,
                const int *e_ii, int c0, float *Vp);
bool tp_bwd_f32(hipStream_t s, long long E, int L, bool scalar_only, int U, const float *pw, const float *V, const float *env,
                const int *e_ii, int c0, const float *dVp, float *dV, float *denv_e);

// ---- neighbor builder (neigh.hip; stubbed in the host-emulation build) -------------------------
void neigh_build(Model &m, int nlocal, int nall, const double *x_dev, const double *lo, const double *hi,
                 double rc_list, hipStream_t s);
void neigh_from_table(Model &m, int inum, int nall, const int *ilist_dev, const int *numneigh_dev, const int *table_dev,
                      long long stride_atom, long long stride_slot, int mask, hipStream_t s);
void map_types(Model &m, int n, const int *type_dev, int ntypes, const int *mapper_host, int *out_dev, hipStream_t s);
void neigh_free(Model &m);
// periodic images of a single rank's own atoms inside the halo (neigh.hip); returns their number (may exceed `capacity`: then the arrays hold the first `capacity`)
int borders_local(Model &m, int nlocal, const double *x, const int *mtype, const double *lo, const double *hi, const double *box, double rc, int capacity,
                  double *xg, int *mtg, long long *src, double *shift, hipStream_t s);
void nve_first_step(int n, int nall, double *x, double *v, double *f, const int *mtype, const double *mass_host, int ntypes, double dt, double ftm2v,
                    hipStream_t s);
void nve_step(int mode, int n, double *x, double *v, const double *f, const int *mtype, const double *mass_dev_or_host,
              int ntypes, double dt, double ftm2v, hipStream_t s);

}  // namespace ahip
