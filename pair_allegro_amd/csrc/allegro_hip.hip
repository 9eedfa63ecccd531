// liballegro_hip.so -- C-ABI implementation (include/allegro_hip.h).
//
// Host orchestration of one force evaluation = PairNequIPAllegro<false>::compute
// (/root/reference/pair_nequip_allegro.cpp:333-407): list ingestion, cutoff filter + edge build,
// model forward/backward (generic or fused kernels), force/energy/virial read-out.
// No libtorch, no Kokkos, no CPU fallback: every compute entry point needs a HIP device.
#include "../../include/allegro_hip.h"

#include <pthread.h>
#include <sched.h>

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "engine.h"
#include "generic_engine.h"

using namespace ahip;

struct ahip_model : public ahip::Model {};

static thread_local std::string g_err;

const char *ahip_last_error(void) { return g_err.c_str(); }
namespace ahip {
void set_error(const std::string &s) { g_err = s; }          // comm.hip
std::atomic<int> g_models_alive{0};                          // edges.hip: unit schedule of the edge build
}

template <typename F> static int guarded(F &&fn) {
  try {
    g_err.clear();
    fn();
    return AHIP_OK;
  } catch (const ArgError &e) { g_err = e.what(); return AHIP_ERR_ARG; }
  catch (const StateError &e) { g_err = e.what(); return AHIP_ERR_STATE; }
  catch (const UnsupportedError &e) { g_err = e.what(); return AHIP_ERR_UNSUPPORTED; }
  catch (const HipError &e) { g_err = e.what(); return AHIP_ERR_DEVICE; }
  catch (const std::bad_alloc &) { g_err = "out of host memory"; return AHIP_ERR_DEVICE; }
  catch (const std::exception &e) { g_err = e.what(); return AHIP_ERR_FILE; }
}

int ahip_device_count(int *count) {
  return guarded([&] {
    if (!count) throw ArgError("ahip_device_count: count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { (void)hipGetLastError(); n = 0; }
    *count = n;
  });
}

static void require_model(const ahip_model *m) { if (!m) throw ArgError("model handle is NULL"); }

static void validate_shape(const HostModel &h) {
  if (h.l_max < 0 || h.l_max > 3) throw UnsupportedError("l_max must be 0 .. 3 (got " + std::to_string(h.l_max) + ")");      // 3: layer-at-a-time kernels only
  if (h.num_layers < 1) throw UnsupportedError("num_layers must be >= 1");
  if (h.num_bessels < 1 || h.S < 1 || h.U < 1 || h.mlp_width < 1) throw UnsupportedError("bad model dimensions");
  if (h.mlp_depth < 0 || h.readout_depth < 0) throw UnsupportedError("bad MLP depth");
  if (h.poly_p < 2) throw UnsupportedError("polynomial_cutoff_p must be >= 2");
  if (!(h.r_max > 0)) throw UnsupportedError("r_max must be positive");
  // every tensor must be present with the documented shape
  const int T = h.num_types, B = h.num_bessels, S = h.S, U = h.U, L = h.l_max, W = h.mlp_width, R = h.readout_width;
  auto need = [&](const std::string &n, std::vector<int> shape) {
    const HostTensor &t = h.get(n);
    if (t.shape != shape) throw std::runtime_error("model file: tensor '" + n + "' has unexpected shape");
  };
  auto need_mlp = [&](const std::string &pre, int din, int depth, int width, int dout) {
    std::vector<int> dims{din};
    for (int k = 0; k < depth; ++k) dims.push_back(width);
    dims.push_back(dout);
    for (size_t k = 0; k + 1 < dims.size(); ++k) need(pre + ".w" + std::to_string(k), {dims[k], dims[k + 1]});
  };
  const int npf = L == 0 ? AHIP_CG_L0_NPATHS : (L == 1 ? AHIP_CG_L1_NPATHS : (L == 2 ? AHIP_CG_L2_NPATHS : AHIP_CG_L3_NPATHS));
  const int nps = L == 0 ? AHIP_CG_L0_NPATHS_SCALAR : (L == 1 ? AHIP_CG_L1_NPATHS_SCALAR : (L == 2 ? AHIP_CG_L2_NPATHS_SCALAR : AHIP_CG_L3_NPATHS_SCALAR));
  need_mlp("tb", 2 * T + B, h.mlp_depth, W, S);
  need("emb.w", {S, U * (L + 1)});
  for (int k = 1; k <= h.num_layers; ++k) {
    const std::string lk = "l" + std::to_string(k);
    need(lk + ".env", {S, U * (L + 1)});
    need(lk + ".tp", {k == h.num_layers ? nps : npf, U});
    need_mlp(lk + ".lat", S + U, h.mlp_depth, W, S);
    need(lk + ".res", {2});
    if (k < h.num_layers) need(lk + ".mix", {L + 1, U, U});
  }
  need_mlp("out", S, h.readout_depth, R, 1);
  need("scale", {T});
  need("shift", {T});
}

int ahip_model_load(const char *path, int device, ahip_model **out) {
  return guarded([&] {
    if (!path || !out) throw ArgError("ahip_model_load: NULL argument");
    *out = nullptr;
    HostModel hm = load_model_file(path);                  // throws runtime_error -> AHIP_ERR_FILE
    validate_shape(hm);
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
      (void)hipGetLastError();
      throw HipError("no HIP device visible: allegro-hip has no CPU fallback (hipGetDeviceCount failed or returned 0)");
    }
    if (device < 0 || device >= n)
      throw ArgError("pair_allegro: mismatch between number of ranks and number of available GPUs (device " +
                     std::to_string(device) + " of " + std::to_string(n) + ")");
    AHIP_CHECK(hipSetDevice(device));
    ahip_model *m = new ahip_model();
    try {
      m->hm = std::move(hm);
      m->device = device;
      m->D = (m->hm.l_max + 1) * (m->hm.l_max + 1);
      m->Ka = 2 * m->hm.num_types + m->hm.num_bessels;
      const int T = m->hm.num_types;
      m->rcut_model_host.assign((size_t)T * T, m->hm.r_max);
      if (!m->hm.per_edge_type_cutoff.empty()) m->rcut_model_host = m->hm.per_edge_type_cutoff;
      AHIP_CHECK(hipMalloc((void **)&m->rcut_model_dev, (size_t)T * T * sizeof(double)));
      AHIP_CHECK(hipMemcpy(m->rcut_model_dev, m->rcut_model_host.data(), (size_t)T * T * sizeof(double), hipMemcpyHostToDevice));
      const int lm_ = m->hm.l_max;
      const AhipCgEntry *tab = lm_ == 0 ? ahip_cg_l0 : (lm_ == 1 ? ahip_cg_l1 : (lm_ == 2 ? ahip_cg_l2 : ahip_cg_l3));
      m->ncg_full = lm_ == 0 ? AHIP_CG_L0_N : (lm_ == 1 ? AHIP_CG_L1_N : (lm_ == 2 ? AHIP_CG_L2_N : AHIP_CG_L3_N));
      m->ncg_scalar = lm_ == 0 ? AHIP_CG_L0_NSCALAR : (lm_ == 1 ? AHIP_CG_L1_NSCALAR : (lm_ == 2 ? AHIP_CG_L2_NSCALAR : AHIP_CG_L3_NSCALAR));
      AHIP_CHECK(hipMalloc(&m->cg_dev, (size_t)m->ncg_full * sizeof(AhipCgEntry)));
      AHIP_CHECK(hipMemcpy(m->cg_dev, tab, (size_t)m->ncg_full * sizeof(AhipCgEntry), hipMemcpyHostToDevice));
    } catch (...) { ahip_model_free(m); throw; }
    *out = m;
    m->counted = true;
    g_models_alive.fetch_add(1);
  });
}

namespace ahip {
int *alarm_word(Model &m) {
  if (!m.h_alarm) {
    AHIP_CHECK(hipHostMalloc((void **)&m.h_alarm, 64, hipHostMallocMapped));
    *m.h_alarm = 0;
  }
  int *d = nullptr;
  AHIP_CHECK(hipHostGetDevicePointer((void **)&d, m.h_alarm, 0));
  return d;
}
bool alarm_take(Model &m) {
  if (m.h_alarm && *(volatile int *)m.h_alarm != 0) { *m.h_alarm = 0; return true; }
  return false;
}
// fused_arith=auto falls back to the f32-input MFMA instances for the rest of this model's life (engine.h): the prepared f16x2 weight streams go, the next
// dispatch prepares the f32 ones; said once on stderr and kept for ahip_arith_note
static void arith_degrade(Model &m, const std::string &why) {
  m.arith_degraded = true;
  m.arith_note = "fused_arith=auto: float32 instance (f32-input MFMA) selected: " + why;
  fused_free(m); fusedlx_free(m); fusedlx2_free(m);
  std::fprintf(stderr, "[allegro-hip] %s\n", m.arith_note.c_str());
}
// An alarm found at the START of an evaluation, or by an accessor, was raised by an EARLIER device-resident evaluation that nobody waited for: its forces were
// not finite and have been handed out.  That is reported either way; under auto the model also switches to the f32 instance, so the error is reported once.
void fused_poll_alarm(Model &m) {
  if (!alarm_take(m)) return;
  if (arith_option(m) == "auto" && !m.arith_degraded) {
    arith_degrade(m, "an earlier evaluation produced a non-finite edge gradient on the f16x2 arithmetic (an activation left float16's range)");
    throw StateError("fused_arith=auto: an EARLIER evaluation produced non-finite forces on the f16x2 arithmetic (an activation left float16's range, or the input was "
                     "not finite); this model now runs on the float32 instance -- re-evaluate from the last valid state");
  }
  throw StateError("fused_arith=f16x2: an edge gradient was not finite (an activation left float16's range, or the input was not finite): the forces of that "
                   "evaluation are invalid; set option fused_arith=f32 (or auto) for this model");
}
}  // namespace ahip

void ahip_model_free(ahip_model *m) {
  if (!m) return;
  (void)hipSetDevice(m->device);
  (void)hipDeviceSynchronize();
  if (m->h_alarm) {
    // an alarm of the LAST evaluation of a run is not lost either (void function: said on stderr)
    if (alarm_take(*m)) std::fprintf(stderr, "[allegro-hip] WARNING: the last device-resident evaluation of this model produced non-finite forces on the f16x2 arithmetic "
                                             "(float16 range exceeded); set fused_arith=f32\n");
    (void)hipHostFree(m->h_alarm); m->h_alarm = nullptr;
  }
  fused_free(*m);
  fusedlx_free(*m);
  fusedlx2_free(*m);
  neigh_free(*m);
  edges_free(*m);
  m->prim.release();
  free_weights(m->wf);
  free_weights(m->wd);
  if (m->cg_dev) (void)hipFree(m->cg_dev);
  if (m->rcut_model_dev) (void)hipFree(m->rcut_model_dev);
  for (DevBuf *b : {&m->b_flagwork, &m->b_ilist, &m->b_nloff, &m->b_nlj, &m->b_x, &m->b_ftype, &m->b_mtype, &m->b_f, &m->b_eatom,
                    &m->b_engvir, &m->b_cutsq, &m->b_cnt, &m->b_eoff, &m->b_eii, &m->b_ej, &m->b_rvec, &m->b_ett, &m->b_partial,
                    &m->b_ws, &m->b_misc, &m->b_chk, &m->hv_eoff, &m->hv_eii, &m->hv_ej, &m->hv_rvec, &m->hv_ilist, &m->hv_engvir, &m->b_tile_a0, &m->b_tile_e0, &m->b_centre, &m->b_ntiles})
    b->release();
  for (auto &t : m->slots) for (auto &e : t.ring) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
  for (hipEvent_t e : m->f_events) (void)hipEventDestroy(e);
#ifndef AHIP_HOST_EMU
  for (auto &kv : m->pinned) (void)hipHostUnregister(kv.second.first);
#endif
  if (m->counted) g_models_alive.fetch_sub(1);
  delete m;
}

int ahip_model_allow_tf32(const ahip_model *m, int *allow) {
  return guarded([&] {
    if (!m || !allow) throw ArgError("ahip_model_allow_tf32: null argument");
    *allow = m->hm.allow_tf32;
  });
}

int ahip_model_meta(const ahip_model *m, double *r_max, int *num_types, const char **type_names,
                    const double **per_edge_type_cutoff, int *l_max, int *num_tensor_features,
                    int *num_scalar_features, int *num_layers, const char **model_dtype) {
  return guarded([&] {
    require_model(m);
    if (r_max) *r_max = m->hm.r_max;
    if (num_types) *num_types = m->hm.num_types;
    if (type_names) *type_names = m->hm.type_names_joined.c_str();
    if (per_edge_type_cutoff) *per_edge_type_cutoff = m->hm.per_edge_type_cutoff.empty() ? nullptr : m->hm.per_edge_type_cutoff.data();
    if (l_max) *l_max = m->hm.l_max;
    if (num_tensor_features) *num_tensor_features = m->hm.U;
    if (num_scalar_features) *num_scalar_features = m->hm.S;
    if (num_layers) *num_layers = m->hm.num_layers;
    if (model_dtype) *model_dtype = m->hm.model_dtype.c_str();
  });
}

int ahip_set_option(ahip_model *m, const char *key, const char *value) {
  return guarded([&] {
    require_model(m);
    if (!key || !value) throw ArgError("ahip_set_option: NULL key/value");
    const std::string k(key), v(value);
    if (k == "path") {
      if (v != "auto" && v != "fused" && v != "generic") throw ArgError("option path: expected auto|fused|generic");
      m->opt_path = v;
    } else if (k == "precision") {
      if (v != "model" && v != "float64") throw ArgError("option precision: expected model|float64");
      m->opt_precision = v;
    } else if (k == "fused_arith") {
      if (v != "bf16x3" && v != "f32" && v != "tf32eq" && v != "f16x2" && v != "auto") throw ArgError("option fused_arith: expected auto|f32|f16x2|bf16x3|tf32eq");
      if (v != m->opt_fused_arith) { m->opt_fused_arith = v; fused_free(*m); fusedlx_free(*m); fusedlx2_free(*m); }     // weight streams are rebuilt on the next compute
    } else if (k == "fused_tb") {
      if (v != "table" && v != "mlp") throw ArgError("option fused_tb: expected table|mlp");
      if (v != m->opt_fused_tb) { m->opt_fused_tb = v; fused_free(*m); }
    } else if (k == "chunk_edges") {
      long long n = std::atoll(value);
      if (n < 1) throw ArgError("option chunk_edges: expected a positive integer");
      m->chunk_edges = n;
    } else if (k == "reserve_wgs") {
      int n = std::atoi(value);
      if (n < 0 || n > 128) throw ArgError("option reserve_wgs: expected 0..128");
      m->reserve_wgs = n;
    } else if (k == "cutoff_compare") {
      if (v != "le" && v != "lt") throw ArgError("option cutoff_compare: expected le|lt");
      m->cutoff_strict = v == "lt";
      m->h_cutsq_dev.clear();
    } else if (k == "edge_schedule") {
      if (v != "auto" && v != "static" && v != "dynamic") throw ArgError("option edge_schedule: expected auto|static|dynamic");
      m->opt_edge_schedule = v;
    } else if (k == "tile_pack") {
      if (v != "auto" && v != "separate" && v != "fused") throw ArgError("option tile_pack: expected auto|separate|fused");
      m->opt_tile_pack = v;
    } else if (k == "timing") {
      m->timing = (v == "1" || v == "on" || v == "true");
    } else throw ArgError("unknown option '" + k + "'");
  });
}

namespace ahip {
namespace {
struct Staging {
  std::mutex mu;
  void *p = nullptr;
  // One event pair PER DEVICE: an event belongs to the device that was current when it was created and hipEventRecord refuses a stream of another
  // device; a process that loads models on two devices (ahip_model_load takes the device) copies through the same page-locked buffer with that device's pair.
  std::map<int, std::array<hipEvent_t, 2>> evs;
  hipEvent_t *ev = nullptr;                 // the current device's pair, valid while `mu` is held (set by get())
  static constexpr size_t BYTES = 8u << 20, HALF = BYTES / 2;
  void *get() {
    if (!p) AHIP_CHECK(hipHostMalloc(&p, BYTES, hipHostMallocPortable));
    int dev = 0;
    AHIP_CHECK(hipGetDevice(&dev));
    auto it = evs.find(dev);
    if (it == evs.end()) {
      std::array<hipEvent_t, 2> e{nullptr, nullptr};
      for (int k = 0; k < 2; ++k) AHIP_CHECK(hipEventCreateWithFlags(&e[k], hipEventDisableTiming));
      it = evs.emplace(dev, e).first;
    }
    ev = it->second.data();
    return p;
  }
};
Staging g_staging;      // process-wide, never freed (the runtime may be gone when static destructors run)
}  // namespace

// ---- host-side loops of the host-pointer path, on a few threads ----
// The reference's own host loops are OpenMP loops (pair_nequip_allegro.cpp:371, 488, 566); this library links nothing but libdl, so the three O(nall)
// loops of ahip_compute (type map, position copy into page-locked memory, f +=) and the staging copies run on std::threads instead: AHIP_HOST_THREADS
// (default: min(16, hardware threads)), one thread below 65 536 items.  fn(begin, end) must be safe to run concurrently on disjoint ranges.
int host_threads() {
  static const int n = [] {
    if (const char *e = std::getenv("AHIP_HOST_THREADS")) return std::max(1, std::atoi(e));
    // the CPUs this process may run on, not the machine's: under `mpirun --bind-to core` a rank owns one core, and sixteen workers
    // taking turns on it are slower than one memcpy
    unsigned hw = std::thread::hardware_concurrency();
#ifdef __linux__
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof(set), &set) == 0 && CPU_COUNT(&set) > 0) hw = (unsigned)CPU_COUNT(&set);
#endif
    return (int)std::max(1u, std::min(16u, hw ? hw : 1u));
  }();
  return n;
}
// A persistent pool (created at the first use, never destroyed: worker threads may outlive static destructors): forking and joining eight
// std::threads costs ~0.3 ms, four times per call, which was half of what the loops themselves take at 1 M atoms.
namespace {
struct HostPool {
  std::mutex mu;
  std::condition_variable cv_go, cv_done;
  std::function<void(size_t, size_t)> job;
  size_t n = 0, per = 0;
  unsigned long long gen = 0;
  int pending = 0, nworkers = 0;
  explicit HostPool(int nw) : nworkers(nw) {
    for (int w = 1; w <= nw; ++w) std::thread([this, w] { work(w); }).detach();
  }
  void work(int w) {
    unsigned long long seen = 0;
    for (;;) {
      std::function<void(size_t, size_t)> fn;
      size_t b, e;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv_go.wait(lk, [&] { return gen != seen; });
        seen = gen;
        fn = job;
        b = std::min(n, (size_t)w * per); e = std::min(n, b + per);
      }
      if (b < e) fn(b, e);
      {
        std::lock_guard<std::mutex> lk(mu);
        if (--pending == 0) cv_done.notify_one();
      }
    }
  }
  void run(size_t count, const std::function<void(size_t, size_t)> &fn) {        // one caller at a time (run_mu)
    {
      std::lock_guard<std::mutex> lk(mu);
      job = fn; n = count; per = (count + nworkers) / (nworkers + 1); pending = nworkers; ++gen;
    }
    cv_go.notify_all();
    fn((size_t)0, std::min(count, per));
    std::unique_lock<std::mutex> lk(mu);
    cv_done.wait(lk, [&] { return pending == 0; });
  }
};
std::mutex g_pool_run_mu;
HostPool *g_pool = nullptr;
// a forked child has none of the parent's worker threads: it starts a pool of its own at its first large loop instead of waiting for ever on the parent's
// (the old pool object is leaked on purpose: its mutex may be held by a thread that does not exist in the child)
void pool_atfork_child() { g_pool = nullptr; new (&g_pool_run_mu) std::mutex(); }
const int g_pool_atfork = pthread_atfork(nullptr, nullptr, pool_atfork_child);
}  // namespace
template <class F> static void parallel_for(size_t n, F fn) {
  const int nt = n < 65536 ? 1 : host_threads();
  if (nt <= 1) { fn((size_t)0, n); return; }
  std::lock_guard<std::mutex> lk(g_pool_run_mu);
  if (!g_pool) g_pool = new HostPool(nt - 1);
  g_pool->run(n, std::function<void(size_t, size_t)>(fn));
}
static void parallel_memcpy(void *dst, const void *src, size_t bytes) {
  parallel_for(bytes, [=](size_t b, size_t e) { std::memcpy((char *)dst + b, (const char *)src + b, e - b); });
}

// Pageable host memory <-> device through the page-locked staging buffer, in two halves: the DMA of one half runs while the host copies the other
// (a 184 MB neighbor list used to be 23 memcpy + synchronous-DMA pairs in a row).
void copy_h2d(void *dst_dev, const void *src_host, size_t bytes) {
  std::lock_guard<std::mutex> lk(g_staging.mu);
  char *st = (char *)g_staging.get();
  int k = 0;
  bool used[2] = {false, false};
  for (size_t o = 0; o < bytes; o += Staging::HALF, k ^= 1) {
    const size_t n = std::min(Staging::HALF, bytes - o);
    if (used[k]) AHIP_CHECK(hipEventSynchronize(g_staging.ev[k]));            // the DMA that last read this half
    parallel_memcpy(st + k * Staging::HALF, (const char *)src_host + o, n);
    AHIP_CHECK(hipMemcpyAsync((char *)dst_dev + o, st + k * Staging::HALF, n, hipMemcpyHostToDevice, nullptr));
    AHIP_CHECK(hipEventRecord(g_staging.ev[k], nullptr));
    used[k] = true;
  }
  AHIP_CHECK(hipStreamSynchronize(nullptr));
}
void copy_d2h(void *dst_host, const void *src_dev, size_t bytes) {
  std::lock_guard<std::mutex> lk(g_staging.mu);
  char *st = (char *)g_staging.get();
  // half k holds chunk c: request chunk c + 1 into the other half, then copy chunk c out while that DMA runs
  const size_t nchunk = (bytes + Staging::HALF - 1) / Staging::HALF;
  auto req = [&](size_t c) {
    const size_t o = c * Staging::HALF, n = std::min(Staging::HALF, bytes - o);
    AHIP_CHECK(hipMemcpyAsync(st + (c & 1) * Staging::HALF, (const char *)src_dev + o, n, hipMemcpyDeviceToHost, nullptr));
    AHIP_CHECK(hipEventRecord(g_staging.ev[c & 1], nullptr));
  };
  if (nchunk) req(0);
  for (size_t c = 0; c < nchunk; ++c) {
    AHIP_CHECK(hipEventSynchronize(g_staging.ev[c & 1]));
    if (c + 1 < nchunk) req(c + 1);
    const size_t o = c * Staging::HALF, n = std::min(Staging::HALF, bytes - o);
    parallel_memcpy((char *)dst_host + o, st + (c & 1) * Staging::HALF, n);
  }
}
}  // namespace ahip

// ------------------------------------------------------------------------------------ neighbor list
static void install_list_host(ahip_model *m, int inum, int nall) {
  AHIP_CHECK(hipSetDevice(m->device));
  const size_t nn = m->h_flat_j.size();
  m->b_ilist.reserve(std::max<size_t>(inum, 1) * sizeof(int));
  m->b_nloff.reserve(((size_t)inum + 1) * sizeof(int));
  m->b_nlj.reserve(std::max<size_t>(nn, 1) * sizeof(int));
  copy_h2d(m->b_ilist.p, m->h_ilist.data(), (size_t)inum * sizeof(int));
  copy_h2d(m->b_nloff.p, m->h_off32.data(), ((size_t)inum + 1) * sizeof(int));
  copy_h2d(m->b_nlj.p, m->h_flat_j.data(), nn * sizeof(int));
  m->d_ilist = m->b_ilist.as<int>();
  m->d_nloff = m->b_nloff.as<int>();
  m->d_nlj = m->b_nlj.as<int>();
  m->inum = inum; m->nall = nall; m->nneigh = (long long)nn; m->have_list = true;
  m->max_list_row = 0;
  for (int ii = 0; ii < inum; ++ii) m->max_list_row = std::max(m->max_list_row, m->h_off32[ii + 1] - m->h_off32[ii]);
}

static void check_list_dims(int inum, int nall) {
  if (inum < 0 || nall < inum) throw ArgError("neighbor list: need 0 <= inum <= nall");
}

int ahip_neigh_update(ahip_model *m, int inum, int nall, const int *ilist, const int *numneigh,
                      const int *const *firstneigh, int neighmask) {
  return guarded([&] {
    require_model(m);
    check_list_dims(inum, nall);
    if (inum > 0 && (!ilist || !numneigh || !firstneigh)) throw ArgError("ahip_neigh_update: NULL list pointer");
    m->h_ilist.assign(ilist, ilist + inum);
    m->h_off32.resize((size_t)inum + 1);
    long long tot = 0;
    for (int ii = 0; ii < inum; ++ii) {
      int i = ilist[ii];
      if (i < 0 || i >= nall) throw ArgError("neighbor list: ilist entry out of range");
      m->h_off32[ii] = (int)tot;
      tot += numneigh[i];
      if (tot > 2147483000LL) throw UnsupportedError("neighbor list too large for 32-bit offsets; use more ranks");
    }
    m->h_off32[inum] = (int)tot;
    m->h_flat_j.resize((size_t)tot);
    for (int ii = 0; ii < inum; ++ii) {
      int i = ilist[ii];
      const int *jl = firstneigh[i];
      int *dst = m->h_flat_j.data() + m->h_off32[ii];
      for (int jj = 0; jj < numneigh[i]; ++jj) {
        int j = jl[jj] & neighmask;                          // pair_nequip_allegro.cpp:496
        if (j < 0 || j >= nall) throw ArgError("neighbor list: neighbour index out of range");
        dst[jj] = j;
      }
    }
    install_list_host(m, inum, nall);
  });
}

int ahip_neigh_update_csr(ahip_model *m, int inum, int nall, const int *ilist, const long long *offsets,
                          const int *neigh, int neighmask) {
  return guarded([&] {
    require_model(m);
    check_list_dims(inum, nall);
    if (inum > 0 && (!ilist || !offsets)) throw ArgError("ahip_neigh_update_csr: NULL list pointer");
    m->h_ilist.assign(ilist, ilist + inum);
    m->h_off32.resize((size_t)inum + 1);
    long long tot = inum > 0 ? offsets[inum] : 0;
    if (tot > 2147483000LL) throw UnsupportedError("neighbor list too large for 32-bit offsets; use more ranks");
    if (tot > 0 && !neigh) throw ArgError("ahip_neigh_update_csr: NULL neigh");
    for (int ii = 0; ii <= inum; ++ii) {
      long long o = inum > 0 ? offsets[ii] : 0;
      if (o < 0 || o > tot || (ii > 0 && o < offsets[ii - 1])) throw ArgError("neighbor list: offsets not monotone");
      m->h_off32[ii] = (int)o;
    }
    for (int ii = 0; ii < inum; ++ii)
      if (ilist[ii] < 0 || ilist[ii] >= nall) throw ArgError("neighbor list: ilist entry out of range");
    m->h_flat_j.resize((size_t)tot);
    for (long long p = 0; p < tot; ++p) {
      int j = neigh[p] & neighmask;
      if (j < 0 || j >= nall) throw ArgError("neighbor list: neighbour index out of range");
      m->h_flat_j[(size_t)p] = j;
    }
    install_list_host(m, inum, nall);
  });
}

int ahip_neigh_update_dev(ahip_model *m, int inum, int nall, const int *ilist_dev, const int *offsets_dev,
                          const int *neigh_dev, long long nneigh_total) {
  return guarded([&] {
    require_model(m);
    check_list_dims(inum, nall);
    if (inum > 0 && (!ilist_dev || !offsets_dev)) throw ArgError("ahip_neigh_update_dev: NULL list pointer");
    m->d_ilist = ilist_dev; m->d_nloff = offsets_dev; m->d_nlj = neigh_dev;
    m->inum = inum; m->nall = nall; m->nneigh = nneigh_total; m->have_list = true;
    // longest row: bounds the degree of every centre until the next hand-over, which is what lets the per-step calls run without a read-back
    AHIP_CHECK(hipSetDevice(m->device));
    m->max_list_row = edges_max_row(*m, inum, offsets_dev);
  });
}

// ------------------------------------------------------------------------------------ compute
namespace ahip {
void timing_drain(Model &m, TimingSlot &t, size_t keep) {
  (void)m;
  while (t.head - t.tail > keep) {
    const auto &e = t.ring[t.tail % TIMING_RING];
    float ms = 0;
    if (hipEventSynchronize(e.second) == hipSuccess && hipEventElapsedTime(&ms, e.first, e.second) == hipSuccess) { t.sum_ms += ms; ++t.count; }
    else (void)hipGetLastError();
    ++t.tail;
  }
}
}  // namespace ahip
// everything recorded since the last report -> timing_names / timing_ms (sums) / timing_counts; waits for the recorded stages
static void collect_timings(ahip_model *m) {
  m->timing_names.clear();
  m->timing_ms.clear();
  m->timing_counts.clear();
  if (!m->timing) return;
  for (auto &t : m->slots) {
    timing_drain(*m, t, 0);
    if (t.count == 0) continue;
    if (!m->timing_names.empty()) m->timing_names += ";";
    m->timing_names += t.name;
    m->timing_ms.push_back(t.sum_ms);
    m->timing_counts.push_back((double)t.count);
    t.sum_ms = 0; t.count = 0;
  }
}

// The kernels keep an edge iff rsq <= bound.  bound = cut^2 gives the host path's `<=` (pair_nequip_allegro.cpp:507); the largest
// double below cut^2 gives the KOKKOS path's strict `<` (pair_nequip_allegro_kokkos.cpp:174) without touching the kernels.
static double cutsq_bound(const ahip_model *m, double c) {
  const double c2 = c * c;
  return m->cutoff_strict ? std::nextafter(c2, 0.0) : c2;
}

static __global__ void k_add_n(long long n, double *dst, const double *src) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) dst[t] += src[t];
}
static __global__ void k_copy_centres(int inum, const int *ilist, double *dst, const double *src) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < inum) dst[ilist[t]] = src[ilist[t]];
}
static __global__ void k_add7(double *dst, const double *src) { if (threadIdx.x < 7) dst[threadIdx.x] += src[threadIdx.x]; }

// The few centres that do not fit a tile of the wide fused kernel: layer-at-a-time float32 kernels on a compact copy of their
// edges; forces and per-atom energies accumulate into the same arrays, energy / virial partial sums are added.
static void heavy_generic(ahip_model *m, const ComputeArgs &a) {
  edges_compact_heavy(*m, a);
  m->hv_engvir.reserve(8 * sizeof(double));
  auto swap_in = [&]() {
    std::swap(m->b_eoff, m->hv_eoff); std::swap(m->b_eii, m->hv_eii); std::swap(m->b_ej, m->hv_ej); std::swap(m->b_rvec, m->hv_rvec);
  };
  const int *il = m->d_ilist;
  const int inum = m->inum;
  const long long ne = m->nedges;
  swap_in();
  m->d_ilist = m->hv_ilist.as<int>();
  m->inum = m->nheavy; m->nedges = m->hv_nedges;
  ComputeArgs a2 = a;
  a2.engvir = m->hv_engvir.as<double>();
  try { generic_run<float>(*m, a2); }
  catch (...) { swap_in(); m->d_ilist = il; m->inum = inum; m->nedges = ne; throw; }
  swap_in(); m->d_ilist = il; m->inum = inum; m->nedges = ne;
  hipLaunchKernelGGL(k_add7, dim3(1), dim3(64), 0, a.stream, a.engvir, a2.engvir);
}

static void run_model_once(ahip_model *m, const ComputeArgs &a);
// dispatch with the auto fallback: a prepare step that finds the model outside the f16x2 split's reach (ArithDegraded) costs one more dispatch, on f32
static void run_model_dispatch(ahip_model *m, const ComputeArgs &a) {
  try { run_model_once(m, a); }
  catch (const ArithDegraded &d) {
    arith_degrade(*m, d.why);
    run_model_once(m, a);
  }
}
static __global__ void k_chk_reduce(long long n, const double *a, const double *b, unsigned long long *out) {
  // out[0] = max |a - b|, out[1] = max |a| as bit patterns (non-negative doubles order like their bits); a non-finite difference maps to +inf
  double d = 0.0, f = 0.0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    double di = fabs(a[i] - b[i]);
    if (!(di < 1.0e300)) di = 1.0e300;
    d = fmax(d, di); f = fmax(f, fabs(a[i]));
  }
  atomicMax(out, __builtin_bit_cast(unsigned long long, d));          // one pair of atomics per thread of a 1024 x 256 grid, once per model: no wave reduction needed
  atomicMax(out + 1, __builtin_bit_cast(unsigned long long, f));
}
// First evaluation of a model whose fused_arith=auto resolves to f16x2 (VERDICT r05 #3c): the same centres once on the f32 instance, once on f16x2, forces
// compared; f16x2 stays only if max|dF| <= 1e-5 max|F| (float32-equivalence on THIS model and THIS configuration, not on the builder's samples) and no
// alarm was raised.  Cost: two extra evaluations and two weight-stream builds, once per model (per pair_coeff).  Energies / virial / per-atom energies are the
// second pass's (or the fallback's); f gets the chosen pass added, as always.
static void run_model_selfcheck(ahip_model *m, const ComputeArgs &a) {
  m->arith_checked = true;
  const long long nf = 3LL * (a.nlocal + a.nghost);
  m->b_chk.reserve((size_t)(2 * nf + 16) * sizeof(double));
  double *f32f = m->b_chk.as<double>(), *f16f = f32f + nf, *ev = f16f + nf;
  unsigned long long *red = (unsigned long long *)(ev + 8);
  hipStream_t s = a.stream;
  AHIP_CHECK(hipMemsetAsync(m->b_chk.p, 0, (size_t)(2 * nf + 16) * sizeof(double), s));
  ComputeArgs a1 = a;
  a1.f = f32f; a1.eatom = nullptr; a1.engvir = ev;
  m->arith_force = 0;
  fused_free(*m); fusedlx_free(*m); fusedlx2_free(*m);
  try { run_model_dispatch(m, a1); } catch (...) { m->arith_force = -1; fused_free(*m); fusedlx_free(*m); fusedlx2_free(*m); throw; }
  m->arith_force = -1;                       // (a shape without a float32 fused instance -- MLP depth 1 / 3 -- was just evaluated by the layer-at-a-time float32 kernels)
  fused_free(*m); fusedlx_free(*m); fusedlx2_free(*m);
  ComputeArgs a2 = a;
  a2.f = f16f;
  run_model_dispatch(m, a2);                 // (a prepare-time fallback inside lands on f32 as well: the comparison below is then trivially green)
  if (m->last_path != "fused_f16x2" && !m->arith_degraded) {
    // this LIST went down the layer-at-a-time path (a centre with more edges than a tile holds): nothing was checked; a later list gets its chance, three times at most
    if (++m->arith_check_attempts < 3) m->arith_checked = false;
    if (nf > 0) hipLaunchKernelGGL(k_add_n, dim3((unsigned)((nf + 255) / 256)), dim3(256), 0, s, nf, a.f, f16f);
    return;
  }
  hipLaunchKernelGGL(k_chk_reduce, dim3(1024), dim3(256), 0, s, nf, f32f, f16f, red);
  unsigned long long hred[2] = {0, 0};
  AHIP_CHECK(hipMemcpyAsync(hred, red, sizeof(hred), hipMemcpyDeviceToHost, s));
  AHIP_CHECK(hipStreamSynchronize(s));
  double dmax, fmax_;
  std::memcpy(&dmax, &hred[0], 8); std::memcpy(&fmax_, &hred[1], 8);
  const bool alarm = alarm_take(*m);
  const bool ok = !alarm && dmax <= 1.0e-5 * fmax_ + 1.0e-30;
  char buf[256];
  if (m->arith_degraded) {                   // fell back while preparing: f16f holds the float32 result
    if (nf > 0) hipLaunchKernelGGL(k_add_n, dim3((unsigned)((nf + 255) / 256)), dim3(256), 0, s, nf, a.f, f16f);
    return;
  }
  if (ok) {
    std::snprintf(buf, sizeof(buf), "fused_arith=auto: f16x2 kept: first evaluation within %.2e max|F| of the float32 instance (max|dF| %.3e, max|F| %.3e; bar 1e-5)",
                  fmax_ > 0 ? dmax / fmax_ : 0.0, dmax, fmax_);
    m->arith_note = buf;
    if (nf > 0) hipLaunchKernelGGL(k_add_n, dim3((unsigned)((nf + 255) / 256)), dim3(256), 0, s, nf, a.f, f16f);
    return;
  }
  std::snprintf(buf, sizeof(buf), "first-evaluation self-check: f16x2 %s (max|dF| %.3e vs the float32 instance, max|F| %.3e; bar 1e-5 max|F|)",
                alarm ? "raised the float16-range alarm" : "disagrees with the float32 instance", dmax, fmax_);
  arith_degrade(*m, buf);
  run_model_dispatch(m, a);                  // float32, straight into the caller's arrays
}
static void run_model(ahip_model *m, const ComputeArgs &a) {
  fused_poll_alarm(*m);                      // raised by an EARLIER device-resident evaluation (nobody waits for those kernels)
  const bool f64 = (m->opt_precision == "float64") || (m->hm.model_dtype == "float64");
  const bool wants_check = !m->arith_checked && !m->arith_degraded && m->inum > 0 && !f64 && m->opt_path != "generic" && arith_option(*m) == "auto" &&
                           !m->hm.allow_tf32 && (fused_model_supported(*m, nullptr) || fusedlx_model_supported(*m, nullptr)) &&
                           std::getenv("AHIP_NO_ARITH_SELFCHECK") == nullptr;
  if (wants_check) {
    run_model_selfcheck(m, a);
    if (m->arith_checked) m->b_chk.release();      // two force arrays of the whole system: not kept for the model's life (hipFree waits for the kernels that still read them)
  } else run_model_dispatch(m, a);
}
static void run_model_once(ahip_model *m, const ComputeArgs &a) {
  m->nedges = 0;
  const bool f64 = (m->opt_precision == "float64") || (m->hm.model_dtype == "float64");
  // the 7 energy / virial sums start from zero: the single-pass edge build clears them in its first kernel, every other way here
  if (m->inum == 0 || f64) AHIP_CHECK(hipMemsetAsync(a.engvir, 0, 7 * sizeof(double), a.stream));
  if (m->inum == 0) return;                                   // empty sub-domain (pair_nequip_allegro.cpp:340-341)
  if (f64) {
    if (m->opt_path == "fused") throw UnsupportedError("the fused MFMA path computes in float32; use path=generic for float64");
    build_edges<double>(*m, a);
    generic_run<double>(*m, a);
    m->last_path = "generic_f64";
    return;
  }
  m->have_ett = false;
  m->nheavy = 0;
  m->heavy_thresh = (m->opt_path != "generic" && !fused_model_supported(*m, nullptr) && fusedlx_model_supported(*m, nullptr)) ? 64 : 0;
  // Tile packing rides on the edge build when the tile shape is known before it runs: k_fused with every list row <= 64 entries (4-wave tiles:
  // 64 slots, 6 centres), the wide kernels always (64 slots, 4 centres).  Otherwise (shape chosen on the device, two-pass edge build) the
  // stand-alone packing kernels run after it, as before.
  m->pack_slots = m->pack_maxa = 0;
  // ... and up to 262 144 centres per call: the packing runs on the scanning wave of every unit, i.e. serially inside the edge build, and costs there what
  // the stand-alone kernels cost beside it once they have a chip to spread over (1 M atoms: 0.062 vs 0.064 ms); below that the six launches they need are
  // the cost (10 648 atoms: 0.026 ms, 125 000: 0.056 ms, three times per step in the overlapped multi-rank schedule)
  if (m->opt_path != "generic" && m->opt_tile_pack != "separate" && (m->inum <= 262144 || m->opt_tile_pack == "fused")) {
    if (fused_model_supported(*m, nullptr)) { if (m->max_list_row >= 0 && m->max_list_row <= 64) { m->pack_slots = 64; m->pack_maxa = 6; } }
    else if (fusedlx_model_supported(*m, nullptr)) { m->pack_slots = 64; m->pack_maxa = 4; }
  }
  m->tiles_packed = false;
  if (!edges_build_f32(*m, a)) { m->nheavy = 0; m->heavy_thresh = 0; AHIP_CHECK(hipMemsetAsync(a.engvir, 0, 7 * sizeof(double), a.stream)); build_edges<float>(*m, a); }
#ifdef AHIP_EXPERIMENT_SWITCHES      // never in the product build: a switch that skips the model returns no forces
  static const bool edges_only = std::getenv("AHIP_EDGES_ONLY") != nullptr;     // timing experiments on the edge build alone
  if (edges_only) { m->last_path = "edges_only"; return; }
#endif
  std::string why;
  bool fused_ok = false;
  if (m->opt_path != "generic") {
    m->last_fused_arith = 0;
    if (fused_model_supported(*m, &why)) fused_ok = fused_run(*m, a, &why);
    else {
      std::string why2;
      if (fusedlx_model_supported(*m, &why2)) { fused_ok = fusedlx_run(*m, a, &why2); why = why2; }
      else why += "; " + why2;
    }
    if (!fused_ok && m->opt_path == "fused") throw UnsupportedError("fused path unavailable: " + why);
  }
  if (fused_ok) {
    // wide kernels: the number of centres left to the layer-at-a-time kernels is read now, with the model kernel already enqueued
    // (the copy sits in front of it in the stream, so the host waits for the edge build only); k_fused (heavy_thresh = 0) never asks
    if (m->heavy_thresh > 0) edges_counts(*m);
    if (m->nheavy > 0) heavy_generic(m, a);
    // "fused_tf32eq": the two-term bf16 split the model file licensed with allow_tf32 = 1 (fused.hip); everything else is float32-exact
    // "fused_f16x2" / "fused_bf16x3": float32-equivalent splits on the f16 / bf16 matrix cores (fused_h.h, fused.hip); "fused_f32": exact fmaf chains
    m->last_path = m->last_fused_arith == 2 ? "fused_tf32eq" : m->last_fused_arith == 3 ? "fused_f16x2" : m->last_fused_arith == 1 ? "fused_bf16x3" : "fused_f32";
    return;
  }
  edges_counts(*m);
  generic_run<float>(*m, a);
  m->last_path = "generic_f32";
}

// Page-locks a persistent host vector for as long as it keeps its storage: copies from / into pageable memory are staged by the runtime
// and "asynchronous" in name only; the vectors of the host-pointer path only ever grow, so this registers once per growth.
// resize_pinned: a growing resize frees the old storage, so the registration of that storage is dropped FIRST (unregistering a freed
// range afterwards, or a new allocation landing on the still-registered old range, was possible at re-neighborings that grow nall)
template <class T> static void unpin_if_growing(ahip::Model *m, std::vector<T> &v, size_t n) {
#ifndef AHIP_HOST_EMU
  if (n <= v.capacity()) return;
  auto it = m->pinned.find((const void *)&v);
  if (it != m->pinned.end()) { (void)hipHostUnregister(it->second.first); m->pinned.erase(it); }
#endif
}
template <class T> static void pin_host(ahip::Model *m, std::vector<T> &v) {
#ifndef AHIP_HOST_EMU
  void *p = (void *)v.data();
  const size_t bytes = v.capacity() * sizeof(T);
  auto it = m->pinned.find((const void *)&v);
  if (it != m->pinned.end() && it->second.first == p && it->second.second == bytes) return;
  if (it != m->pinned.end()) { (void)hipHostUnregister(it->second.first); m->pinned.erase(it); }
  if (p && bytes && hipHostRegister(p, bytes, hipHostRegisterDefault) == hipSuccess) m->pinned[(const void *)&v] = {p, bytes};
  else (void)hipGetLastError();          // registration is an optimisation: pageable copies still work
#endif
}

int ahip_compute(ahip_model *m, int nlocal, int nghost, const double *x, const int *type, int ntypes,
                 const int *type_mapper, const double *cutoff_matrix, double *f, double *eatom, double *eng,
                 double *virial) {
  return guarded([&] {
    require_model(m);
    if (!m->have_list) throw StateError("ahip_compute called before ahip_neigh_update");
    if (nlocal < 0 || nghost < 0) throw ArgError("ahip_compute: negative atom count");
    const int nall = nlocal + nghost;
    if (nall != m->nall) throw StateError("ahip_compute: nlocal+nghost differs from the neighbor list's nall; call ahip_neigh_update after re-neighboring");
    if (nall > 0 && (!x || !type || !f)) throw ArgError("ahip_compute: NULL x/type/f");
    if (ntypes <= 0 || !type_mapper || !cutoff_matrix) throw ArgError("ahip_compute: bad ntypes/type_mapper/cutoff_matrix");
    if (!eng) throw ArgError("ahip_compute: eng is NULL");
    AHIP_CHECK(hipSetDevice(m->device));
    hipStream_t s = nullptr;
    *eng = 0;
    if (virial) for (int k = 0; k < 6; ++k) virial[k] = 0;
    m->custom_out.clear();
    if (m->inum == 0) return;          // empty domain: nothing is stored (compute_allegro.cpp:106-112)
    for (const std::string &nm : m->custom_names)                      // the reference's output.at(name) (:404-405)
      if (nm != "atomic_energy" && nm != "forces" && nm != "virial" && nm != "total_energy")
        throw ArgError("model output '" + nm + "' not found (this model returns atomic_energy, forces, virial, total_energy)");
    const bool want_eatom = eatom != nullptr || !m->custom_names.empty();

    // types: LAMMPS 1-based -> filter index (type-1) and model type (pair_nequip_allegro.cpp:576)
    unpin_if_growing(m, m->h_ftype, (size_t)nall); unpin_if_growing(m, m->h_mtype, (size_t)nall);
    m->h_ftype.resize(nall); m->h_mtype.resize(nall);
    {
      std::atomic<int> bad_range{0}, bad_map{0};           // first offending LAMMPS type of each kind (0: none): the threads report, this thread throws
      const int nmt = m->hm.num_types;
      int *const hft = m->h_ftype.data(), *const hmt = m->h_mtype.data();
      parallel_for((size_t)nall, [&, hft, hmt](size_t b, size_t e) {
        for (size_t i = b; i < e; ++i) {
          const int t = type[i];
          if (t < 1 || t > ntypes) { bad_range.store(t ? t : -1); hft[i] = 0; hmt[i] = 0; continue; }
          const int mt = type_mapper[t - 1];
          if (mt < 0 || mt >= nmt) { bad_map.store(t); hft[i] = 0; hmt[i] = 0; continue; }
          hft[i] = t - 1; hmt[i] = mt;
        }
      });
      if (bad_range.load()) throw ArgError("ahip_compute: atom type out of range");
      if (bad_map.load()) throw ArgError("ahip_compute: LAMMPS type " + std::to_string(bad_map.load()) + " is not mapped to a model type (all pair coeffs are not set)");
    }
    std::vector<double> cutsq((size_t)ntypes * ntypes);
    for (size_t k = 0; k < cutsq.size(); ++k) cutsq[k] = cutsq_bound(m, cutoff_matrix[k]);
    m->b_x.reserve((size_t)nall * 3 * sizeof(double));
    m->b_ftype.reserve((size_t)nall * sizeof(int));
    m->b_mtype.reserve((size_t)nall * sizeof(int));
    m->b_f.reserve((size_t)nall * 3 * sizeof(double));
    m->b_eatom.reserve((size_t)nall * sizeof(double));
    m->b_engvir.reserve(8 * sizeof(double));
    m->b_cutsq.reserve(cutsq.size() * sizeof(double));
    // positions: one host copy into a page-locked buffer, then a true DMA (LAMMPS' atom->x itself is pageable and not ours to register)
    unpin_if_growing(m, m->h_x, (size_t)nall * 3);
    m->h_x.resize((size_t)nall * 3);
    pin_host(m, m->h_x); pin_host(m, m->h_ftype); pin_host(m, m->h_mtype);
    {
      // in chunks: the DMA of chunk c runs while the threads copy chunk c + 1 into the page-locked buffer
      const size_t tot = (size_t)nall * 3 * sizeof(double), CH = 16u << 20;
      for (size_t o = 0; o < tot; o += CH) {
        const size_t n = std::min(CH, tot - o);
        parallel_memcpy((char *)m->h_x.data() + o, (const char *)x + o, n);
        AHIP_CHECK(hipMemcpyAsync((char *)m->b_x.p + o, (const char *)m->h_x.data() + o, n, hipMemcpyHostToDevice, s));
      }
    }
    AHIP_CHECK(hipMemcpyAsync(m->b_ftype.p, m->h_ftype.data(), (size_t)nall * sizeof(int), hipMemcpyHostToDevice, s));
    AHIP_CHECK(hipMemcpyAsync(m->b_mtype.p, m->h_mtype.data(), (size_t)nall * sizeof(int), hipMemcpyHostToDevice, s));
    AHIP_CHECK(hipMemcpyAsync(m->b_cutsq.p, cutsq.data(), cutsq.size() * sizeof(double), hipMemcpyHostToDevice, s));
    m->h_cutsq_dev.clear();
    unpin_if_growing(m, m->h_f, (size_t)nall * 3);
    m->h_f.resize((size_t)nall * 3);
    pin_host(m, m->h_f);
    double ev[7];
    constexpr size_t FCH = 8u << 20;
    const size_t ftot = (size_t)nall * 3 * sizeof(double), nfch = (ftot + FCH - 1) / FCH;
    while (m->f_events.size() < nfch + 1) { hipEvent_t e; AHIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); m->f_events.push_back(e); }
    // (twice at most: an evaluation whose f16x2 kernels raise the float16-range alarm under fused_arith=auto is repeated on the float32 instance)
    for (int attempt = 0;; ++attempt) {
    AHIP_CHECK(hipMemsetAsync(m->b_f.p, 0, (size_t)nall * 3 * sizeof(double), s));
    if (want_eatom) AHIP_CHECK(hipMemsetAsync(m->b_eatom.p, 0, (size_t)nall * sizeof(double), s));

    ComputeArgs a{nlocal, nghost, m->b_x.as<double>(), m->b_ftype.as<int>(), m->b_cutsq.as<double>(), ntypes,
                  m->b_mtype.as<int>(), m->b_f.as<double>(), want_eatom ? m->b_eatom.as<double>() : nullptr,
                  m->b_engvir.as<double>(), s};
    run_model(m, a);

    // Order on the stream: the 7 energy / virial sums and an event (the kernel has finished once it fires: the float16-range alarm of THIS evaluation is
    // valid and is reported by this call, before f is touched), then the forces in chunks, each behind an event: the host adds chunk c into f while
    // the DMA of chunk c + 1 runs; the per-atom energies last.
    AHIP_CHECK(hipMemcpyAsync(ev, m->b_engvir.p, 7 * sizeof(double), hipMemcpyDeviceToHost, s));
    AHIP_CHECK(hipEventRecord(m->f_events[nfch], s));
    for (size_t c = 0; c < nfch; ++c) {
      const size_t o = c * FCH, n = std::min(FCH, ftot - o);
      AHIP_CHECK(hipMemcpyAsync((char *)m->h_f.data() + o, (const char *)m->b_f.p + o, n, hipMemcpyDeviceToHost, s));
      AHIP_CHECK(hipEventRecord(m->f_events[c], s));
    }
    if (want_eatom) {
      unpin_if_growing(m, m->h_eatom, (size_t)nall);
      m->h_eatom.resize(nall);
      pin_host(m, m->h_eatom);
      AHIP_CHECK(hipMemcpyAsync(m->h_eatom.data(), m->b_eatom.p, (size_t)nall * sizeof(double), hipMemcpyDeviceToHost, s));
    }
    AHIP_CHECK(hipEventSynchronize(m->f_events[nfch]));
    if (alarm_take(*m)) {                    // raised by THIS evaluation: nothing of it has touched the caller's arrays yet
      (void)hipStreamSynchronize(s);         // (the copies into the page-locked vectors finish before anybody may free or refill them)
      if (arith_option(*m) == "auto" && !m->arith_degraded && attempt == 0) {
        arith_degrade(*m, "an activation left float16's range (non-finite edge gradient on the f16x2 arithmetic); the evaluation was repeated");
        continue;
      }
      throw StateError("fused_arith=f16x2: an edge gradient was not finite (an activation left float16's range, or the input was not finite): the forces of this "
                       "evaluation are invalid and were not added to f; set option fused_arith=f32 (or auto) for this model");
    }
    break;
    }
    // scatter: f[i] += forces[i] for locals AND ghosts (pair_nequip_allegro.cpp:370-377)
    {
      const double *const hf = m->h_f.data();
      for (size_t c = 0; c < nfch; ++c) {
        AHIP_CHECK(hipEventSynchronize(m->f_events[c]));
        const size_t k0 = c * (FCH / sizeof(double)), k1 = std::min((size_t)nall * 3, k0 + FCH / sizeof(double));
        parallel_for(k1 - k0, [=](size_t b, size_t e) { for (size_t k = k0 + b; k < k0 + e; ++k) f[k] += hf[k]; });
      }
    }
    AHIP_CHECK(hipStreamSynchronize(s));
    if (eatom) {
      const double *const he = m->h_eatom.data();
      const int *const il = m->h_ilist.empty() ? nullptr : m->h_ilist.data();
      parallel_for((size_t)m->inum, [=](size_t b, size_t e) { for (size_t ii = b; ii < e; ++ii) { const int i = il ? il[ii] : (int)ii; eatom[i] = he[i]; } });
    }
    *eng = ev[0];
    if (virial) for (int k = 0; k < 6; ++k) virial[k] = ev[1 + k];
    if (!m->custom_names.empty()) {
      // the model's output dict entries (pair_nequip_allegro.cpp:403-406): rows for locals AND ghosts
      std::vector<double> ae(nall);
      const HostTensor &shift = m->hm.get("shift");
      for (int i = 0; i < nall; ++i) ae[i] = shift.data[m->h_mtype[i]];        // ghosts: no centre edges -> shift only
      for (int ii = 0; ii < m->inum; ++ii) { int i = m->h_ilist.empty() ? ii : m->h_ilist[ii]; ae[i] = m->h_eatom[i]; }
      double tot = 0;
      for (double v : ae) tot += v;
      for (const std::string &nm : m->custom_names) {
        if (nm == "atomic_energy") m->custom_out[nm] = ae;
        else if (nm == "forces") m->custom_out[nm] = m->h_f;
        else if (nm == "total_energy") m->custom_out[nm] = {tot};
        else m->custom_out[nm] = {ev[1], ev[4], ev[5], ev[4], ev[2], ev[6], ev[5], ev[6], ev[3]};    // [3][3] from xx yy zz xy xz yz
      }
    }
  });
}

int ahip_output_register(ahip_model *m, const char *name) {
  return guarded([&] {
    require_model(m);
    if (!name || !*name) throw ArgError("ahip_output_register: empty name");
    m->custom_names.push_back(name);
  });
}

int ahip_output_get(ahip_model *m, const char *name, double *out, long long capacity, long long *count) {
  return guarded([&] {
    require_model(m);
    if (!name || !count) throw ArgError("ahip_output_get: NULL name/count");
    auto it = m->custom_out.find(name);
    if (it == m->custom_out.end())
      throw StateError(std::string("output '") + name + "' is not stored: register it with ahip_output_register before ahip_compute / "
                       "ahip_compute_dev (nlocal > 0)");
    *count = (long long)it->second.size();
    if (!out) return;
    if (capacity < *count) throw ArgError("ahip_output_get: buffer too small");
    std::copy(it->second.begin(), it->second.end(), out);
  });
}

int ahip_compute_dev(ahip_model *m, int nlocal, int nghost, const double *x_dev, const int *mtype_dev,
                     const double *cutoff_matrix_model, double *f_dev, double *eatom_dev, double *eng_vir_dev,
                     void *stream) {
  return guarded([&] {
    require_model(m);
    if (!m->have_list) throw StateError("ahip_compute_dev called before a neighbor list was installed");
    const int nall = nlocal + nghost;
    if (nall != m->nall) throw StateError("ahip_compute_dev: nlocal+nghost differs from the neighbor list's nall");
    if (!x_dev || !mtype_dev || !f_dev || !eng_vir_dev) throw ArgError("ahip_compute_dev: NULL device pointer");
    AHIP_CHECK(hipSetDevice(m->device));
    hipStream_t s = (hipStream_t)stream;
    const int T = m->hm.num_types;
    std::vector<double> cutsq((size_t)T * T);
    for (size_t k = 0; k < cutsq.size(); ++k) {
      double c = cutoff_matrix_model ? cutoff_matrix_model[k] : m->rcut_model_host[k];
      cutsq[k] = cutsq_bound(m, c);
    }
    if (cutsq != m->h_cutsq_dev) {                           // upload (and synchronise) only when the matrix changes
      m->b_cutsq.reserve(cutsq.size() * sizeof(double));
      AHIP_CHECK(hipMemcpyAsync(m->b_cutsq.p, cutsq.data(), cutsq.size() * sizeof(double), hipMemcpyHostToDevice, s));
      AHIP_CHECK(hipStreamSynchronize(s));                   // cutsq is a stack vector
      m->h_cutsq_dev = cutsq;
    }
    if (m->custom_names.empty()) {
      ComputeArgs a{nlocal, nghost, x_dev, mtype_dev, m->b_cutsq.as<double>(), T, mtype_dev, f_dev, eatom_dev, eng_vir_dev, s};
      run_model(m, a);
    } else {
      // `compute allegro` on the device path (the reference's Kokkos class keeps output.at(name) too,
      // pair_nequip_allegro_kokkos.cpp:342-344): evaluate into the library's own zeroed force / energy arrays, add them to the
      // caller's, and keep host copies of the named entries.  Costs two extra array passes and a read-back, only when a
      // compute is registered.
      for (const std::string &nm : m->custom_names)
        if (nm != "atomic_energy" && nm != "forces" && nm != "virial" && nm != "total_energy")
          throw ArgError("model output '" + nm + "' not found (this model returns atomic_energy, forces, virial, total_energy)");
      m->b_f.reserve((size_t)std::max(nall, 1) * 3 * sizeof(double));
      m->b_eatom.reserve((size_t)std::max(nall, 1) * sizeof(double));
      AHIP_CHECK(hipMemsetAsync(m->b_f.p, 0, (size_t)nall * 3 * sizeof(double), s));
      AHIP_CHECK(hipMemsetAsync(m->b_eatom.p, 0, (size_t)nall * sizeof(double), s));
      ComputeArgs a{nlocal, nghost, x_dev, mtype_dev, m->b_cutsq.as<double>(), T, mtype_dev, m->b_f.as<double>(), m->b_eatom.as<double>(), eng_vir_dev, s};
      run_model(m, a);
      const int inum = m->inum;
      if (nall > 0) hipLaunchKernelGGL(k_add_n, dim3((unsigned)((3LL * nall + 255) / 256)), dim3(256), 0, s, 3LL * nall, f_dev, m->b_f.as<double>());
      if (eatom_dev && inum > 0) hipLaunchKernelGGL(k_copy_centres, dim3((unsigned)((inum + 255) / 256)), dim3(256), 0, s, inum, m->d_ilist, eatom_dev, m->b_eatom.as<double>());
      AHIP_CHECK(hipGetLastError());
      unpin_if_growing(m, m->h_f, (size_t)nall * 3); unpin_if_growing(m, m->h_eatom, (size_t)nall); unpin_if_growing(m, m->h_mtype, (size_t)nall);
      m->h_f.resize((size_t)nall * 3); m->h_eatom.resize(nall); m->h_mtype.resize(nall);
      std::vector<int> il(inum);
      double ev[7];
      AHIP_CHECK(hipStreamSynchronize(s));
      copy_d2h(m->h_f.data(), m->b_f.p, (size_t)nall * 3 * sizeof(double));          // pageable vectors: staged (engine.h)
      copy_d2h(m->h_eatom.data(), m->b_eatom.p, (size_t)nall * sizeof(double));
      copy_d2h(m->h_mtype.data(), mtype_dev, (size_t)nall * sizeof(int));
      if (inum > 0) copy_d2h(il.data(), m->d_ilist, (size_t)inum * sizeof(int));
      copy_d2h(ev, eng_vir_dev, 7 * sizeof(double));
      std::vector<double> ae(nall);
      const HostTensor &shift = m->hm.get("shift");
      for (int i = 0; i < nall; ++i) ae[i] = shift.data[m->h_mtype[i]];        // ghosts: no centre edges -> shift only
      for (int ii = 0; ii < inum; ++ii) ae[il[ii]] = m->h_eatom[il[ii]];
      double tot = 0;
      for (double v : ae) tot += v;
      for (const std::string &nm : m->custom_names) {
        if (nm == "atomic_energy") m->custom_out[nm] = ae;
        else if (nm == "forces") m->custom_out[nm] = m->h_f;
        else if (nm == "total_energy") m->custom_out[nm] = {tot};
        else m->custom_out[nm] = {ev[1], ev[4], ev[5], ev[4], ev[2], ev[6], ev[5], ev[6], ev[3]};
      }
    }
  });
}

int ahip_compute_dev_range(ahip_model *m, int centre_begin, int centre_end, int nlocal, int nghost, const double *x_dev,
                           const int *mtype_dev, const double *cutoff_matrix_model, double *f_dev, double *eatom_dev,
                           double *eng_vir_dev, void *stream) {
  return guarded([&] {
    require_model(m);
    if (!m->have_list) throw StateError("ahip_compute_dev_range called before a neighbor list was installed");
    if (centre_begin < 0 || centre_end < centre_begin || centre_end > m->inum)
      throw ArgError("ahip_compute_dev_range: need 0 <= centre_begin <= centre_end <= inum");
    if (!m->custom_names.empty()) throw StateError("ahip_compute_dev_range: registered model outputs (compute allegro) need the whole-list call ahip_compute_dev");
    // narrow the installed CSR list to the centre range (offsets are absolute into the neighbour array), evaluate, restore
    const int inum = m->inum;
    const int *il = m->d_ilist, *off = m->d_nloff;
    m->d_ilist = il + centre_begin; m->d_nloff = off + centre_begin; m->inum = centre_end - centre_begin;
    const int rc = ahip_compute_dev(m, nlocal, nghost, x_dev, mtype_dev, cutoff_matrix_model, f_dev, eatom_dev, eng_vir_dev, stream);
    const std::string err = g_err;
    m->d_ilist = il; m->d_nloff = off; m->inum = inum;
    if (rc == AHIP_ERR_ARG) throw ArgError(err);
    if (rc == AHIP_ERR_STATE) throw StateError(err);
    if (rc == AHIP_ERR_UNSUPPORTED) throw UnsupportedError(err);
    if (rc != AHIP_OK) throw HipError(err);
  });
}

long long ahip_last_list_size(ahip_model *m) { return m ? m->nneigh : 0; }

int ahip_get_edges(ahip_model *m, long long *nedges, long long *edge_index, double *rij) {
  return guarded([&] {
    require_model(m);
    if (!nedges) throw ArgError("ahip_get_edges: nedges is NULL");
    edges_counts(*m);
    fused_poll_alarm(*m);                    // (an accessor that waits for the device also reports an alarm nobody has collected)
    *nedges = m->nedges;
    if (!edge_index && !rij) return;
    const size_t E = (size_t)m->nedges;
    if (E == 0) return;
    AHIP_CHECK(hipSetDevice(m->device));
    AHIP_CHECK(hipDeviceSynchronize());
    std::vector<int> eii(E), ej(E), il(m->inum);
    copy_d2h(eii.data(), m->b_eii.p, E * sizeof(int));
    copy_d2h(ej.data(), m->b_ej.p, E * sizeof(int));
    copy_d2h(il.data(), m->d_ilist, (size_t)m->inum * sizeof(int));
    if (edge_index)
      for (size_t e = 0; e < E; ++e) { edge_index[e] = il[eii[e]]; edge_index[E + e] = ej[e]; }
    if (rij) {
      if (m->edges_T_size == 8) {
        std::vector<double> r(E * 3);
        copy_d2h(r.data(), m->b_rvec.p, E * 3 * sizeof(double));
        for (size_t e = 0; e < E; ++e) rij[e] = std::sqrt(r[3 * e] * r[3 * e] + r[3 * e + 1] * r[3 * e + 1] + r[3 * e + 2] * r[3 * e + 2]);
      } else {
        std::vector<float> r(E * 3);
        copy_d2h(r.data(), m->b_rvec.p, E * 3 * sizeof(float));
        for (size_t e = 0; e < E; ++e) {
          double a = r[3 * e], b = r[3 * e + 1], c = r[3 * e + 2];
          rij[e] = std::sqrt(a * a + b * b + c * c);
        }
      }
    }
  });
}

int ahip_debug_dump_edges(ahip_model *m, const int *tag) {
  return guarded([&] {
    require_model(m);
    long long E = 0;
    std::vector<long long> ei;
    std::vector<double> r;
    if (ahip_get_edges(m, &E, nullptr, nullptr) != AHIP_OK) throw std::runtime_error(g_err);
    ei.resize((size_t)2 * E); r.resize((size_t)E);
    if (E > 0 && ahip_get_edges(m, &E, ei.data(), r.data()) != AHIP_OK) throw std::runtime_error(g_err);
    // exact format of pair_nequip_allegro.cpp:564,625,632
    std::printf("Allegro edges: i j rij\n");
    for (long long e = 0; e < E; ++e) {
      long long i = ei[e], j = ei[E + e];
      if (tag) std::printf("%d %d %.10g\n", tag[i] - 1, tag[j] - 1, r[e]);
      else std::printf("%lld %lld %.10g\n", i, j, r[e]);
    }
    std::printf("end Allegro edges\n");
    std::fflush(stdout);
  });
}

int ahip_get_timings(ahip_model *m, const char **names, const double **ms, int *n) {
  return guarded([&] {
    require_model(m);
    AHIP_CHECK(hipSetDevice(m->device));
    collect_timings(m);
    fused_poll_alarm(*m);
    if (names) *names = m->timing_names.c_str();
    if (ms) *ms = m->timing_ms.data();
    if (n) *n = (int)m->timing_ms.size();
  });
}
int ahip_get_timing_counts(ahip_model *m, const double **counts, int *n) {
  return guarded([&] {
    require_model(m);
    if (counts) *counts = m->timing_counts.data();
    if (n) *n = (int)m->timing_counts.size();
  });
}

// Diagnostics: edge slots of the tiles of the last fused evaluation and how many of them held an edge (the padding tax of the tile packing:
// bulk Si 0.875, Li3PO4 ~0.77, water ~0.82).  Reads two device words (synchronises the default stream); 0 / 0 when the last path was not a fused one.
extern "C" int ahip_last_tile_occupancy(ahip_model *m, long long *slots_used, long long *slots_total) {
  return guarded([&] {
    require_model(m);
    if (!slots_used || !slots_total) throw ArgError("ahip_last_tile_occupancy: NULL argument");
    *slots_used = 0; *slots_total = 0;
    if (!m->d_ntiles_last || m->last_path.rfind("fused", 0) != 0) return;
    AHIP_CHECK(hipSetDevice(m->device));
    AHIP_CHECK(hipDeviceSynchronize());        // the tile count may come from the stand-alone packing kernels on a non-blocking stream: a null-stream copy does not wait for those
    edges_counts(*m);
    int nt = 0;
    AHIP_CHECK(hipMemcpy(&nt, m->d_ntiles_last, sizeof(int), hipMemcpyDeviceToHost));
    const int slots = m->last_tile_slots ? m->last_tile_slots : (m->last_max_deg <= 64 ? 64 : 128);
    *slots_used = m->nedges; *slots_total = (long long)nt * slots;
  });
}

// last kernel family used ("generic_f32" | "generic_f64" | "fused_f32" | "fused_tf32eq")
extern "C" const char *ahip_last_path(ahip_model *m) { return m ? m->last_path.c_str() : ""; }
// what fused_arith=auto decided for this model and why (empty until the first evaluation): "f16x2 kept: ..." or "float32 instance selected: ..."
extern "C" const char *ahip_arith_note(const ahip_model *m) { return m ? m->arith_note.c_str() : ""; }
extern "C" int ahip_last_max_degree(ahip_model *m) {
  if (!m) return 0;
  if (guarded([&] { edges_counts(*m); }) != 0) return -1;
  return m->last_max_deg;
}

int ahip_build_neighbors_dev(ahip_model *m, int nlocal, int nall, const double *x_dev, const double *lo,
                             const double *hi, double rc_list, void *stream) {
  return guarded([&] {
    require_model(m);
    if (nlocal < 0 || nall < nlocal) throw ArgError("ahip_build_neighbors_dev: need 0 <= nlocal <= nall");
    if (!x_dev || !lo || !hi || !(rc_list > 0)) throw ArgError("ahip_build_neighbors_dev: bad argument");
    AHIP_CHECK(hipSetDevice(m->device));
    neigh_build(*m, nlocal, nall, x_dev, lo, hi, rc_list, (hipStream_t)stream);
  });
}

int ahip_neigh_update_dev_table(ahip_model *m, int inum, int nall, const int *ilist_dev, const int *numneigh_dev,
                                const int *neighbors_dev, long long stride_atom, long long stride_slot, int neighmask, void *stream) {
  return guarded([&] {
    require_model(m);
    check_list_dims(inum, nall);
    if (inum > 0 && (!ilist_dev || !numneigh_dev || !neighbors_dev)) throw ArgError("ahip_neigh_update_dev_table: NULL list pointer");
    if (stride_atom < 1 || stride_slot < 1) throw ArgError("ahip_neigh_update_dev_table: strides must be positive");
    AHIP_CHECK(hipSetDevice(m->device));
    neigh_from_table(*m, inum, nall, ilist_dev, numneigh_dev, neighbors_dev, stride_atom, stride_slot, neighmask, (hipStream_t)stream);
  });
}

int ahip_map_types_dev(ahip_model *m, int n, const int *type_dev, int ntypes, const int *type_mapper, int *mtype_dev, void *stream) {
  return guarded([&] {
    require_model(m);
    if (n < 0 || ntypes < 1 || !type_mapper || (n > 0 && (!type_dev || !mtype_dev))) throw ArgError("ahip_map_types_dev: bad argument");
    AHIP_CHECK(hipSetDevice(m->device));
    map_types(*m, n, type_dev, ntypes, type_mapper, mtype_dev, (hipStream_t)stream);
  });
}

int ahip_reneighbor_flag_dev(ahip_model *m, int n, const double *x_dev, const double *xhold_dev, const double *v_dev, double dt,
                             double half_skin, int *flag_dev, void *stream) {
  return guarded([&] {
    require_model(m);
    if (n < 0 || !flag_dev || (n > 0 && (!x_dev || !xhold_dev || !v_dev))) throw ArgError("ahip_reneighbor_flag_dev: bad argument");
    AHIP_CHECK(hipSetDevice(m->device));
    if (!m->b_flagwork.p) { m->b_flagwork.reserve(64); AHIP_CHECK(hipMemsetAsync(m->b_flagwork.p, 0, 64, (hipStream_t)stream)); }      // the two maxima start from zero; the flag kernel resets them after every use
    AHIP_CHECK(prim_reneighbor_flag(x_dev, xhold_dev, v_dev, n, dt, half_skin, m->b_flagwork.as<unsigned int>(), flag_dev, (hipStream_t)stream));
  });
}

int ahip_borders_local_dev(ahip_model *m, int nlocal, const double *x_dev, const int *mtype_dev, const double *lo, const double *hi, const double *box, double rc,
                           int capacity, double *xg_dev, int *mtg_dev, long long *src_dev, double *shift_dev, int *nghost, void *stream) {
  return guarded([&] {
    require_model(m);
    if (nlocal < 0 || capacity < 0 || !lo || !hi || !box || !nghost || !(rc > 0.0) || (nlocal > 0 && (!x_dev || !mtype_dev)) ||
        (capacity > 0 && (!xg_dev || !mtg_dev || !src_dev || !shift_dev)))
      throw ArgError("ahip_borders_local_dev: bad argument");
    AHIP_CHECK(hipSetDevice(m->device));
    *nghost = borders_local(*m, nlocal, x_dev, mtype_dev, lo, hi, box, rc, capacity, xg_dev, mtg_dev, src_dev, shift_dev, (hipStream_t)stream);
  });
}

int ahip_nve_first_dev(ahip_model *m, int nlocal, int nall, double *x_dev, double *v_dev, double *f_dev, const int *mtype_dev,
                       const double *mass_by_mtype, double dt, double ftm2v, void *stream) {
  return guarded([&] {
    require_model(m);
    if (nlocal < 0 || nall < nlocal || (nall > 0 && !f_dev) || (nlocal > 0 && (!x_dev || !v_dev || !mtype_dev || !mass_by_mtype)))
      throw ArgError("ahip_nve_first_dev: bad argument");
    AHIP_CHECK(hipSetDevice(m->device));
    nve_first_step(nlocal, nall, x_dev, v_dev, f_dev, mtype_dev, mass_by_mtype, m->hm.num_types, dt, ftm2v, (hipStream_t)stream);
  });
}

int ahip_nve_dev(ahip_model *m, int mode, int n, double *x_dev, double *v_dev, const double *f_dev,
                 const int *mtype_dev, const double *mass_by_mtype, double dt, double ftm2v, void *stream) {
  return guarded([&] {
    require_model(m);
    if (n < 0 || (n > 0 && (!x_dev || !v_dev || !f_dev || !mtype_dev || !mass_by_mtype))) throw ArgError("ahip_nve_dev: bad argument");
    if (mode != 0 && mode != 1) throw ArgError("ahip_nve_dev: mode must be 0 or 1");
    AHIP_CHECK(hipSetDevice(m->device));
    nve_step(mode, n, x_dev, v_dev, f_dev, mtype_dev, mass_by_mtype, m->hm.num_types, dt, ftm2v, (hipStream_t)stream);
  });
}
