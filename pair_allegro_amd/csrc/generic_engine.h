// Generic path orchestration: edge build (shared with the fused path) and the layer-at-a-time
// forward / backward over chunks of centre atoms.
#pragma once
#include <type_traits>

#include "engine.h"
#include "generic_kernels.h"
#include "prims.h"

namespace ahip {

template <typename... KArgs, typename... Args>
static inline void launch(void (*k)(KArgs...), long long nthreads, hipStream_t s, Args... args) {
  if (nthreads <= 0) return;
  const unsigned block = 256;
  const unsigned grid = (unsigned)((nthreads + block - 1) / block);
  hipLaunchKernelGGL(k, dim3(grid), dim3(block), 0, s, args...);
}

template <typename T> DeviceWeights<T> &weights_of(Model &m);
template <> inline DeviceWeights<float> &weights_of<float>(Model &m) { return m.wf; }
template <> inline DeviceWeights<double> &weights_of<double>(Model &m) { return m.wd; }

template <typename T> static void upload_weights(Model &m) {
  DeviceWeights<T> &dw = weights_of<T>(m);
  if (dw.ready) return;
  auto put = [&](const std::string &name, const std::vector<double> &src) {
    std::vector<T> tmp(src.size());
    for (size_t i = 0; i < src.size(); ++i) tmp[i] = (T)src[i];
    void *p = nullptr;
    AHIP_CHECK(hipMalloc(&p, std::max<size_t>(tmp.size(), 1) * sizeof(T)));
    copy_h2d(p, tmp.data(), tmp.size() * sizeof(T));
    dw.owned.push_back(p);
    dw.w[name] = (T *)p;
  };
  for (const auto &kv : m.hm.tensors) put(kv.first, kv.second.data);
  put("res.identity", std::vector<double>{0.0, 1.0});        // x0 = 0*() + 1*fc*u0
  dw.ready = true;
}

template <typename T> static void free_weights(DeviceWeights<T> &dw) {
  for (void *p : dw.owned) (void)hipFree(p);
  dw.owned.clear(); dw.w.clear(); dw.ready = false;
}

// ---- edge build: K1/K2/K5 of the reference's Kokkos path, a2-a4 of SURVEY section 8 -------------
// After this: m.nedges, m.b_eoff (int[inum+1]), m.b_eii, m.b_ej, m.b_rvec (T[E][3]).
template <typename T> static void build_edges(Model &m, const ComputeArgs &a) {
  StageTimer tm(m, "edge_build", a.stream);
  const int inum = m.inum;
  m.b_cnt.reserve((size_t)(inum + 1) * sizeof(int));
  m.b_eoff.reserve((size_t)(inum + 2) * sizeof(int));
  launch(k_count_edges, inum, a.stream, inum, m.d_ilist, m.d_nloff, m.d_nlj, a.x, a.ftype, a.cutsq, a.nft,
         m.b_cnt.as<int>());
  AHIP_CHECK(prim_exclusive_scan_i32(m.prim, m.b_cnt.as<int>(), m.b_eoff.as<int>(), inum, a.stream));
  m.b_misc.reserve(64);
  AHIP_CHECK(prim_max_i32(m.b_cnt.as<int>(), inum, m.b_misc.as<int>(), a.stream));
  int tot = 0, mx = 0;
  // the one scalar read-back per step (the Kokkos path has the same: pair_nequip_allegro_kokkos.cpp:203-206)
  AHIP_CHECK(hipMemcpyAsync(&tot, m.b_eoff.as<int>() + inum, sizeof(int), hipMemcpyDeviceToHost, a.stream));
  AHIP_CHECK(hipMemcpyAsync(&mx, m.b_misc.as<int>(), sizeof(int), hipMemcpyDeviceToHost, a.stream));
  AHIP_CHECK(hipStreamSynchronize(a.stream));
  m.nedges = tot;
  m.last_max_deg = mx;
  const size_t E = (size_t)std::max(tot, 1);
  m.b_eii.reserve(E * sizeof(int));
  m.b_ej.reserve(E * sizeof(int));
  m.b_rvec.reserve(E * 3 * sizeof(T));
  m.edges_T_size = (int)sizeof(T);
  launch(k_fill_edges<T>, inum, a.stream, inum, m.d_ilist, m.d_nloff, m.d_nlj, a.x, a.ftype, a.cutsq, a.nft,
         m.b_eoff.as<int>(), m.b_eii.as<int>(), m.b_ej.as<int>(), m.b_rvec.as<T>());
}

// ---- one chunk of centres [c0, c0+nc), edges [e0, e0+Ec) ----------------------------------------
template <typename T>
static void generic_chunk(Model &m, const ComputeArgs &a, Arena &A, int c0, int nc, long long e0, long long Ec) {
  const HostModel &h = m.hm;
  const DeviceWeights<T> &W = weights_of<T>(m);
  hipStream_t s = a.stream;
  const bool go = !A.measuring;
  const int S = h.S, U = h.U, L = h.l_max, D = m.D, nl = L + 1, NL = h.num_layers, Wd = h.mlp_width;
  const int depth = h.mlp_depth, rdepth = h.readout_depth, R = h.readout_width, Ka = m.Ka, Tn = h.num_types;
  const T cenv = (T)(1.0 / std::sqrt(h.avg_num_neighbors));
  const int *eoff = m.b_eoff.as<int>();
  const int *e_ii = m.b_eii.as<int>() + e0;
  const int *e_j = m.b_ej.as<int>() + e0;
  const T *rvec = m.b_rvec.as<T>() + 3 * e0;
  const AhipCgEntry *cg = (const AhipCgEntry *)m.cg_dev;
  GeomParams gp{h.num_bessels, h.poly_p, L, D, Tn, h.r_max};
  const size_t E = (size_t)Ec;

#define RUN(...) do { if (go) { launch(__VA_ARGS__); } } while (0)
  // dense layers: MFMA GEMM for float32 (gemm.hip), the portable kernels otherwise
  auto linear_fwd = [&](int K, int N, const T *in, int ldin, const T *Wp, T *out, int ldout) {
    if (!go) return;
    if constexpr (std::is_same<T, float>::value)
      if (gemm_f32(s, Ec, K, N, in, ldin, Wp, N, false, out, ldout, false)) return;
    launch(k_linear<T>, Ec * N, s, Ec, K, N, in, ldin, Wp, out, ldout);
  };
  auto linear_bwd = [&](int K, int N, const T *dout, int lddout, const T *Wp, T *din, int lddin, int accumulate) {
    if (!go) return;
    if constexpr (std::is_same<T, float>::value)
      if (gemm_f32(s, Ec, N, K, dout, lddout, Wp, N, true, din, lddin, accumulate != 0)) return;
    launch(k_linear_bwd<T>, Ec * K, s, Ec, K, N, dout, lddout, Wp, din, lddin, accumulate);
  };
  auto lofl = [](int lm) { return lm == 0 ? 0 : (lm < 4 ? 1 : (lm < 9 ? 2 : 3)); };
  // channel mixing V[e][lm][:] = Vp[e][lm][:] @ mix[l]: one GEMM per lm with row stride D*U (float32), else k_mix
  auto mix_fwd = [&](int Dm, const T *Vp_, const T *mixw, T *Vout) {
    if (!go) return;
    if constexpr (std::is_same<T, float>::value) {
      bool ok = true;
      for (int lm = 0; lm < Dm && ok; ++lm)
        ok = gemm_f32(s, Ec, U, U, Vp_ + lm * U, Dm * U, mixw + (size_t)lofl(lm) * U * U, U, false, Vout + lm * U, Dm * U, false);
      if (ok) return;
    }
    launch(k_mix<T>, Ec * Dm * U, s, Ec, Dm, U, Vp_, mixw, Vout);
  };
  // dVp[e][lm][:] = dV[e][lm][:] @ mix[l]^T (+ ds on lm = 0)
  auto mix_bwd = [&](int Dm, const T *dV_, const T *mixw, const T *ds, int ldds, T *dVp_) {
    if (!go) return;
    if constexpr (std::is_same<T, float>::value) {
      if (dV_) {
        bool ok = true;
        for (int lm = 0; lm < Dm && ok; ++lm)
          ok = gemm_f32(s, Ec, U, U, dV_ + lm * U, Dm * U, mixw + (size_t)lofl(lm) * U * U, U, true, dVp_ + lm * U, Dm * U, false);
        if (ok) {
          if (ds) launch(k_add_cols<T>, Ec * U, s, Ec, U, ds, ldds, dVp_, Dm * U);
          return;
        }
      }
    }
    launch(k_mix_bwd<T>, Ec * Dm * U, s, Ec, Dm, U, dV_, mixw, ds, ldds, dVp_);
  };
  auto latent_bwd = [&](const T *dx_, const T *u_, const T *fc_, const T *resw, T *du_, T *dfc_, T *dxprev_) {
    if (!go) return;
    if constexpr (std::is_same<T, float>::value)
      if (latent_update_bwd_f32(s, Ec, S, dx_, u_, fc_, resw, du_, dfc_, dxprev_)) return;
    launch(k_latent_update_bwd<T>, Ec, s, Ec, S, dx_, u_, fc_, resw, du_, dfc_, dxprev_);
  };
  auto embed_bwd_Y = [&](const T *dV_, const T *w_, T *dY_) {
    if (!go) return;
    if constexpr (std::is_same<T, float>::value)
      if (embed_bwd_Y_f32(s, Ec, D, U, dV_, w_, dY_)) return;
    launch(k_embed_bwd_Y<T>, Ec * D, s, Ec, D, U, dV_, w_, dY_);
  };

  // ---------------- forward ----------------
  T *fc = A.get<T>(E), *Y = A.get<T>(E * D), *a_in = A.get<T>(E * Ka);
  RUN(k_geom_fwd<T>, Ec, s, Ec, gp, rvec, e_ii, e_j, m.d_ilist, a.mtype, m.rcut_model_dev, fc, Y, a_in);

  auto mlp_fwd = [&](const std::string &pre, int nhidden, const T *in, int din, int width, int dout,
                     std::vector<T *> &zs, T *&out) {
    const T *cur = in; int K = din;
    for (int k = 0; k < nhidden; ++k) {
      T *z = A.get<T>(E * width), *hh = A.get<T>(E * width);
      bool fusedsilu = false;                       // float32: the GEMM epilogue also writes h = silu(z)
      if constexpr (std::is_same<T, float>::value)
        if (go) fusedsilu = gemm_f32(s, Ec, K, width, cur, K, W.get(pre + ".w" + std::to_string(k)), width, false, z, width, false, hh, nullptr);
      if (!fusedsilu) {
        linear_fwd(K, width, cur, K, go ? W.get(pre + ".w" + std::to_string(k)) : nullptr, z, width);
        RUN(k_silu<T>, Ec * width, s, Ec * width, z, hh);
      }
      zs.push_back(z); cur = hh; K = width;
    }
    out = A.get<T>(E * dout);
    linear_fwd(K, dout, cur, K, go ? W.get(pre + ".w" + std::to_string(nhidden)) : nullptr, out, dout);
  };
  // backward of mlp: dout [E][dout] -> din [E][din]; returns pointer to din
  auto mlp_bwd = [&](const std::string &pre, int nhidden, int din, int width, int dout, const std::vector<T *> &zs,
                     T *dout_p) -> T * {
    T *d = dout_p; int N = dout;
    for (int k = nhidden; k >= 0; --k) {
      int K = (k == 0) ? din : width;
      T *dprev = A.get<T>(E * K);
      bool fusedsilu = false;                       // float32: the GEMM epilogue multiplies by silu'(z) of the layer below
      if constexpr (std::is_same<T, float>::value)
        if (go && k > 0) fusedsilu = gemm_f32(s, Ec, N, K, d, N, W.get(pre + ".w" + std::to_string(k)), N, true, dprev, K, false, nullptr, zs[k - 1]);
      if (!fusedsilu) {
        linear_bwd(K, N, d, N, go ? W.get(pre + ".w" + std::to_string(k)) : nullptr, dprev, K, 0);
        if (k > 0) RUN(k_silu_bwd<T>, Ec * K, s, Ec * K, zs[k - 1], dprev, dprev);
      }
      d = dprev; N = K;
    }
    return d;
  };

  std::vector<T *> z_tb;
  T *u0 = nullptr;
  mlp_fwd("tb", depth, a_in, Ka, Wd, S, z_tb, u0);
  std::vector<T *> x(NL + 1), u(NL + 1), om(NL + 1), env(NL + 1), V(NL + 1);
  std::vector<std::vector<T *>> z_lat(NL + 1);
  x[0] = A.get<T>(E * S);
  RUN(k_latent_update<T>, Ec * S, s, Ec, S, (const T *)nullptr, u0, fc, go ? W.get("res.identity") : nullptr, x[0]);
  T *w0 = A.get<T>(E * nl * U);
  linear_fwd(S, nl * U, x[0], S, go ? W.get("emb.w") : nullptr, w0, nl * U);
  V[0] = A.get<T>(E * D * U);
  RUN(k_embed<T>, Ec * D * U, s, Ec, D, U, w0, Y, V[0]);

  for (int k = 1; k <= NL; ++k) {
    const bool last = (k == NL);
    const std::string lk = "l" + std::to_string(k);
    const int Dout = last ? 1 : D;
    const int ncg = last ? m.ncg_scalar : m.ncg_full;
    om[k] = A.get<T>(E * nl * U);
    linear_fwd(S, nl * U, x[k - 1], S, go ? W.get(lk + ".env") : nullptr, om[k], nl * U);
    env[k] = A.get<T>((size_t)nc * D * U);
    RUN(k_env_reduce<T>, (long long)nc * D * U, s, nc, c0, eoff, e0, D, U, om[k], Y, cenv, env[k]);
    T *Vp = A.get<T>(E * Dout * U);
    {
      bool done = !go;
      if constexpr (std::is_same<T, float>::value)
        if (go) done = tp_fwd_f32(s, Ec, m.hm.l_max, last, U, W.get(lk + ".tp"), V[k - 1], env[k], e_ii, c0, Vp);
      if (!done) launch(k_tp_fwd<T>, Ec * U, s, Ec, D, Dout, U, cg, ncg, W.get(lk + ".tp"), V[k - 1], env[k], e_ii, c0, Vp);
    }
    T *cat = A.get<T>(E * (S + U));
    RUN(k_concat<T>, Ec * (S + U), s, Ec, S, U, x[k - 1], Vp, Dout * U, cat);
    mlp_fwd(lk + ".lat", depth, cat, S + U, Wd, S, z_lat[k], u[k]);
    x[k] = A.get<T>(E * S);
    RUN(k_latent_update<T>, Ec * S, s, Ec, S, x[k - 1], u[k], fc, go ? W.get(lk + ".res") : nullptr, x[k]);
    if (!last) {
      V[k] = A.get<T>(E * D * U);
      mix_fwd(D, Vp, go ? W.get(lk + ".mix") : nullptr, V[k]);
    }
  }
  std::vector<T *> z_out;
  T *eps = nullptr;
  mlp_fwd("out", rdepth, x[NL], S, R, 1, z_out, eps);

  // ---------------- backward ----------------
  T *deps = A.get<T>(E);
  RUN(k_seed_deps<T>, Ec, s, Ec, e_ii, m.d_ilist, a.mtype, go ? W.get("scale") : nullptr, cenv, deps);
  T *dx = mlp_bwd("out", rdepth, S, R, 1, z_out, deps);
  T *dfc = A.get<T>(E), *dY = A.get<T>(E * D);
  RUN(k_fill<T>, Ec, s, Ec, dfc, T(0));
  RUN(k_fill<T>, Ec * D, s, Ec * D, dY, T(0));
  T *dV = nullptr;
  for (int k = NL; k >= 1; --k) {
    const bool last = (k == NL);
    const std::string lk = "l" + std::to_string(k);
    const int Dout = last ? 1 : D;
    const int ncg = last ? m.ncg_scalar : m.ncg_full;
    T *du = A.get<T>(E * S), *dxprev = A.get<T>(E * S);
    latent_bwd(dx, u[k], fc, go ? W.get(lk + ".res") : nullptr, du, dfc, dxprev);
    T *dcat = mlp_bwd(lk + ".lat", depth, S + U, Wd, S, z_lat[k], du);
    RUN(k_add_cols<T>, Ec * S, s, Ec, S, dcat, S + U, dxprev, S);
    T *dVp = A.get<T>(E * Dout * U);
    mix_bwd(Dout, (const T *)(last ? nullptr : dV), (const T *)((last || !go) ? nullptr : W.get(lk + ".mix")), (const T *)(dcat + S),
            S + U, dVp);
    T *dVprev = A.get<T>(E * D * U), *denv_e = A.get<T>(E * D * U);
    {
      bool done = !go;
      if constexpr (std::is_same<T, float>::value)
        if (go) done = tp_bwd_f32(s, Ec, m.hm.l_max, last, U, W.get(lk + ".tp"), V[k - 1], env[k], e_ii, c0, dVp, dVprev, denv_e);
      if (!done) launch(k_tp_bwd<T>, Ec * U, s, Ec, D, Dout, U, cg, ncg, W.get(lk + ".tp"), V[k - 1], env[k], e_ii, c0, dVp, dVprev, denv_e);
    }
    T *denv = A.get<T>((size_t)nc * D * U);
    RUN(k_segment_sum<T>, (long long)nc * D * U, s, nc, c0, eoff, e0, D * U, denv_e, cenv, denv);
    T *dom = A.get<T>(E * nl * U);
    RUN(k_env_bwd_om<T>, Ec * nl * U, s, Ec, D, U, denv, e_ii, c0, Y, dom);
    {
      bool done = !go;
      if constexpr (std::is_same<T, float>::value)
        if (go) done = env_bwd_Y_f32(s, Ec, D, U, denv, e_ii, c0, om[k], dY);
      if (!done) launch(k_env_bwd_Y<T>, Ec * D, s, Ec, D, U, denv, e_ii, c0, om[k], dY);
    }
    linear_bwd(S, nl * U, dom, nl * U, go ? W.get(lk + ".env") : nullptr, dxprev, S, 1);
    dx = dxprev; dV = dVprev;
  }
  T *dw0 = A.get<T>(E * nl * U);
  RUN(k_embed_bwd_w<T>, Ec * nl * U, s, Ec, D, U, dV, Y, dw0);
  embed_bwd_Y(dV, w0, dY);
  linear_bwd(S, nl * U, dw0, nl * U, go ? W.get("emb.w") : nullptr, dx, S, 1);
  T *du0 = A.get<T>(E * S);
  latent_bwd(dx, u0, fc, go ? W.get("res.identity") : nullptr, du0, dfc, (T *)nullptr);
  T *da = mlp_bwd("tb", depth, Ka, Wd, S, z_tb, du0);
  T *g = A.get<T>(E * 3);
  RUN(k_geom_bwd<T>, Ec, s, Ec, gp, rvec, e_ii, e_j, m.d_ilist, a.mtype, m.rcut_model_dev, da, dfc, dY, g);
  RUN(k_readout<T>, nc, s, nc, c0, eoff, e0, m.d_ilist, a.mtype, e_j, rvec, eps, g, go ? W.get("scale") : nullptr,
      go ? W.get("shift") : nullptr, cenv, a.f, a.eatom, m.b_partial.as<double>());
#undef RUN
}

// ---- all centres, chunked -----------------------------------------------------------------------
template <typename T> static void generic_run(Model &m, const ComputeArgs &a) {
  upload_weights<T>(m);
  const int inum = m.inum;
  m.b_partial.reserve((size_t)std::max(inum, 1) * 7 * sizeof(double));
  // chunk boundaries (whole centres per chunk)
  std::vector<int> cuts{0};
  if (m.nedges <= m.chunk_edges) {
    cuts.push_back(inum);
    m.h_eoff.assign({0});
  } else {
    m.h_eoff.resize((size_t)inum + 1);
    AHIP_CHECK(hipStreamSynchronize(a.stream));
    copy_d2h(m.h_eoff.data(), m.b_eoff.p, ((size_t)inum + 1) * sizeof(int));
    int c = 0;
    while (c < inum) {
      int c1 = c + 1;                                         // at least one centre per chunk
      while (c1 < inum && (long long)m.h_eoff[c1 + 1] - m.h_eoff[c] <= m.chunk_edges) ++c1;
      cuts.push_back(c1);
      c = c1;
    }
  }
  auto eoff_at = [&](int c) -> long long {
    if (m.h_eoff.size() == 1) return c == 0 ? 0 : m.nedges;
    return m.h_eoff[c];
  };
  // measure the largest chunk
  size_t need = 0;
  for (size_t q = 0; q + 1 < cuts.size(); ++q) {
    Arena A; A.measuring = true;
    generic_chunk<T>(m, a, A, cuts[q], cuts[q + 1] - cuts[q], eoff_at(cuts[q]), eoff_at(cuts[q + 1]) - eoff_at(cuts[q]));
    need = std::max(need, A.off);
  }
  m.b_ws.reserve(need + 256);
  {
    StageTimer tm(m, "model_generic", a.stream);
    for (size_t q = 0; q + 1 < cuts.size(); ++q) {
      Arena A; A.measuring = false; A.base = (char *)m.b_ws.p;
      long long e0 = eoff_at(cuts[q]), Ec = eoff_at(cuts[q + 1]) - e0;
      generic_chunk<T>(m, a, A, cuts[q], cuts[q + 1] - cuts[q], e0, Ec);
    }
  }
  AHIP_CHECK(prim_sum_columns_f64(m.prim, m.b_partial.as<double>(), inum, 7, a.engvir, a.stream));
  AHIP_CHECK(hipGetLastError());
}

}  // namespace ahip
