// Round-5 micro-benchmark / probe for the f16x2 arithmetic (csrc/fused_h.h) and the two-groups-per-fragment linears.  Not part of the product.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics -fno-slp-vectorize -I pair_allegro_amd/csrc pair_allegro_amd/tools/g2_rate.hip \
//         pair_allegro_amd/csrc/{allegro_hip,prims,neigh,edges,model_io,gemm,fused_lx,fused_lx2,comm,fused_bf}.o -ldl -o gpurun_out/g2_rate
// 1. does the f16 / bf16 MFMA honour subnormal inputs?  (decides whether unscaled low terms would be usable at all)
// 2. one 64x64 linear through linear_h against a float64 product: error of the f16x2 arithmetic next to plain float32
// 3. rates: cycles per 64x64 linear with the SiLU + save + split epilogue, one and two groups per fragment, one and two waves per SIMD
#include "../csrc/fused.hip"
#include "../csrc/fused_h.h"
namespace ahip {
// (micro-benchmark only: the packed-f32 SiLU epilogue measured in round 5; the kernels use the scalar one)
// out = silu(z) and the saved rows silu'(z) as EpiSiluSaveD, on register pairs: five packed operations and four transcendentals per two values instead of
// seven plain operations per value (the f16x2 linears are VALU-bound: their MFMAs take a fifth of the f32-input form's time)
struct EpiSiluSaveD2 {
  static constexpr bool STORES = false;
  __amdgpu_buffer_rsrc_t S; int row0, v16;
  f32x4 d[2];
  __device__ __forceinline__ void tile_done(int, const f32x4 &) const {}
  __device__ __forceinline__ f32x2 apply2(int ot, int r, f32x2 z) {
    const f32x2 t = z * -1.4426950408889634f;
    const f32x2 o = f32x2{__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])} + 1.f;
    const f32x2 sg = {__builtin_amdgcn_rcpf(o[0]), __builtin_amdgcn_rcpf(o[1])};
    const f32x2 y = z * sg;
    const f32x2 dd = y * (1.f - sg) + sg;
    d[ot & 1][r] = dd[0]; d[ot & 1][r + 1] = dd[1];
    return y;
  }
  __device__ __forceinline__ float apply(int ot, int r, float z) {
    const float sg = sigmoidf_fast(z), y = z * sg;
    d[ot & 1][r] = fmaf(y, 1.f - sg, sg);
    return y;
  }
  __device__ __forceinline__ void flush(int ot0) const {
    bstore(S, v16, (row0 + ot0) * ROW * 4, d[0]);
    bstore(S, v16, (row0 + ot0 + 1) * ROW * 4, d[1]);
  }
};

}  // namespace ahip

using namespace ahip;

__global__ void k_denorm(float *out) {
  const int lane = threadIdx.x;
  // A = all 2^-20 (f16 subnormal), B = all 1.0: exact result 32 * 2^-20 per output; and the mirror image
  const _Float16 sub = (_Float16)9.5367431640625e-07f, one = (_Float16)1.0f;
  f16x8 a, b;
  for (int k = 0; k < 8; ++k) { a[k] = sub; b[k] = one; }
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  f32x4 d = {0.f, 0.f, 0.f, 0.f};
  d = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, d, 0, 0, 0);
  // bf16 subnormal: 2^-130
  const unsigned short bsub = 0x0010, bone = 0x3f80;          // bf16 bit patterns: 2^-126 * 2^-3 (mantissa bit 4 of 7) = 2^-129; 1.0
  typedef unsigned short us8 __attribute__((ext_vector_type(8)));
  us8 ua, ub;
  for (int k = 0; k < 8; ++k) { ua[k] = bsub; ub[k] = bone; }
  f32x4 e = {0.f, 0.f, 0.f, 0.f};
  e = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ua), __builtin_bit_cast(bf16x8, ub), e, 0, 0, 0);
  if (lane == 0) { out[0] = c[0]; out[1] = d[0]; out[2] = e[0]; }
}

// out[32][N] = in[32][K] @ W: two waves, 16 rows each, through linear_h<1>
template <int KS, int NT>
__global__ void __launch_bounds__(128) k_lin_h(const float *Wf, int wbytes, const float *in, int K, float *out, int N) {
  const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4, row = (threadIdx.x >> 6) * 16 + j;
  f32x4 a[2 * KS], o[1][NT];
  for (int t = 0; t < 2 * KS; ++t)
    for (int r = 0; r < 4; ++r) { const int k = feat16(t, r, g); a[t][r] = k < K ? in[row * K + k] : 0.f; }
  Hop b[1][KS], ob[1][NT / 2];
  for (int ks = 0; ks < KS; ++ks) b[0][ks] = split_pair_h(a[2 * ks], a[2 * ks + 1]);
  __amdgpu_buffer_rsrc_t WB = __builtin_amdgcn_make_buffer_rsrc((void *)Wf, 0, wbytes, 0x00020000);
  u32x4 ring[RINGH];
  int wp = 0;
  ring_prime_b(WB, wp, lane * 16, ring);
  EpiNone ep[1];
  linear_h<1, KS, NT, false, false, 0>(WB, wp, b, o, ob, lane * 16, ring, ep);
  for (int t = 0; t < NT; ++t)
    for (int r = 0; r < 4; ++r) { const int n = feat16(t, r, g); if (n < N) out[row * N + n] = o[0][t][r]; }
}

struct EpiNoneX : EpiNone { EpiNoneX() = default; __device__ EpiNoneX(__amdgpu_buffer_rsrc_t, int, int) {} };
template <int G, int OCC, int BALLAST, class EPI = EpiSiluSaveD>
__global__ void __launch_bounds__(256, OCC) k_rate_h(const float *W, int wbytes, float *scr, long long *out, int iters) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, v16 = lane * 16;
  __shared__ float pad[OCC == 2 ? 20000 : 40000];
  pad[threadIdx.x] = 0.f;
  __amdgpu_buffer_rsrc_t WB = __builtin_amdgcn_make_buffer_rsrc((void *)W, 0, wbytes, 0x00020000);
  __amdgpu_buffer_rsrc_t SB;
  {
    unsigned long long b = (unsigned long long)(scr + ((size_t)blockIdx.x * 4 + wave) * 16 * ROW);
    unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    SB = __builtin_amdgcn_make_buffer_rsrc((void *)(((unsigned long long)hi << 32) | lo), 0, 16 * ROW * 4, 0x00020000);
  }
  f32x4 x[G][4], y[G][4];
  u32x4 ring[RINGH];
  float bal[BALLAST > 0 ? BALLAST : 1];
  for (int i = 0; i < BALLAST; ++i) bal[i] = 0.5f * (float)(lane + i);
  for (int g = 0; g < G; ++g)
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 4; ++r) x[g][t][r] = 0.001f * (float)((lane * 7 + t * 4 + r + 3 * g) % 13);
  Hop xa[G][2], ya[G][2];
  for (int g = 0; g < G; ++g) { xa[g][0] = split_pair_h(x[g][0], x[g][1]); xa[g][1] = split_pair_h(x[g][2], x[g][3]); }
  int wp = 0;
  ring_prime_b(WB, wp, v16, ring);
  for (int it = 0; it < iters; ++it) {
    wp = 0;
#pragma unroll
    for (int i = 0; i < BALLAST; ++i) asm volatile("" : "+v"(bal[i]));
    EPI e0[G], e1[G];
#pragma unroll
    for (int g = 0; g < G; ++g) { e0[g] = EPI{SB, 8 * g, v16}; e1[g] = EPI{SB, 8 * g + 4, v16}; }
    linear_h<G, 2, 4, false, true, 0>(WB, wp, xa, y, ya, v16, ring, e0);
    linear_h<G, 2, 4, false, true, 0>(WB, wp, ya, x, xa, v16, ring, e1);
  }
  float sum = 0.f;
  for (int g = 0; g < G; ++g) for (int t = 0; t < 4; ++t) for (int r = 0; r < 4; ++r) sum += x[g][t][r];
  for (int i = 0; i < BALLAST; ++i) sum += bal[i];
  if (sum == 12345.678f) out[0] = (long long)pad[lane];
}


// ---- round 5, second probe: weight fragments SHARED by the four waves of a workgroup through an LDS ring filled by LDS-DMA ----
// Each wave requests a quarter of every step's four fragments (one global_load_lds_dwordx4 = 1 KiB per wave and step, inline asm: hipcc drains every
// builtin LDS-DMA with vmcnt(0) at the next barrier), three steps ahead; a step = s_waitcnt vmcnt(2) (the wave's own request for this step has
// landed: in-order completion, two younger requests may stay in flight) + s_barrier (everybody's has) + four ds_read_b128 + the MFMAs of linear_h.
// Ring: 4 steps x 4 KiB; the slot refilled at step s was read at step s - 1, i.e. before every wave reached this step's barrier.
__device__ __forceinline__ void hs_dma(const float *wbase, int wo_bytes, unsigned lds_dst, int lane) {
  const char *src = (const char *)wbase + wo_bytes + lane * 16;
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(src), "s"(lds_dst) : "memory");
}
template <int KS, int NT, bool SPLIT, class Epi>
__device__ __forceinline__ void linear_hs(const float *wbase, unsigned ring_lds, int uwave, int lane, int &gs, int &wo, const Hop (&in)[KS], f32x4 (&out)[NT], Hop (&outb)[NT / 2], Epi &epi) {
  constexpr int NP = NT / 2, NSTEP = NP * KS;
  f32x4 ah[2], ac[2], prev[2];
#pragma unroll
  for (int s = 0; s < NSTEP; ++s) {
    const int p = s / KS, ks = s % KS;
    if (ks == 0) { ah[0] = ah[1] = ac[0] = ac[1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    u32x4 a[4];
    {
      const unsigned ad = ring_lds + (unsigned)(gs & 3) * 4096u + (unsigned)lane * 16u;
      asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\tds_read_b128 %3, %4 offset:3072\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(a[0]), "=&v"(a[1]), "=&v"(a[2]), "=&v"(a[3]) : "v"(ad) : "memory");
    }
    hs_dma(wbase, wo + uwave * 1024, ring_lds + (unsigned)((gs + 3) & 3) * 4096u + (unsigned)uwave * 1024u, lane);
    wo += 4096;
    pin_s(wo);
    ++gs;
    pin_s(gs);
#pragma unroll
    for (int m = 0; m < 3; ++m) {
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        if (m == 0) ac[hh] = mfma_h(a[2 + hh], in[ks].hi, ac[hh]);
        else if (m == 1) ac[hh] = mfma_h(a[hh], in[ks].lo, ac[hh]);
        else ah[hh] = mfma_h(a[hh], in[ks].hi, ah[hh]);
        if (p > 0) {
          const int idx = ks * 6 + m * 2 + hh, tot = KS * 6, NE = 4;
          const int e0 = (idx * NE + tot - 1) / tot, e1 = ((idx + 1) * NE + tot - 1) / tot;
#pragma unroll
          for (int e = 0; e < NE; ++e)
            if (e >= e0 && e < e1) {
              const int th = e / 2, r = 2 * (e % 2), ot = 2 * (p - 1) + th;
              const f32x2 y = epi_apply2(epi, ot, r, f32x2{prev[th][r], prev[th][r + 1]}, 0);
              out[ot][r] = y[0]; out[ot][r + 1] = y[1];
              if (e == 3) {
                epi.flush(2 * (p - 1));
                if (SPLIT) outb[p - 1] = split_pair_h(out[2 * (p - 1)], out[2 * (p - 1) + 1]);
              }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (ks == KS - 1) {
      const f32x4 r0 = ac[0] * H_LO_INV + ah[0], r1 = ac[1] * H_LO_INV + ah[1];
      epi.tile_done(2 * p, r0);
      epi.tile_done(2 * p + 1, r1);
      if (p == NP - 1) {
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
          const f32x2 y0 = epi_apply2(epi, 2 * p, r, f32x2{r0[r], r0[r + 1]}, 0), y1 = epi_apply2(epi, 2 * p + 1, r, f32x2{r1[r], r1[r + 1]}, 0);
          out[2 * p][r] = y0[0]; out[2 * p][r + 1] = y0[1]; out[2 * p + 1][r] = y1[0]; out[2 * p + 1][r + 1] = y1[1];
        }
        epi.flush(2 * p);
        if (SPLIT) outb[p] = split_pair_h(out[2 * p], out[2 * p + 1]);
      } else { prev[0] = r0; prev[1] = r1; }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}
template <class EPI>
__global__ void __launch_bounds__(256, 2) k_rate_hs(const float *W, int wbytes, float *scr, long long *out, int iters) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, v16 = lane * 16;
  const int uwave = __builtin_amdgcn_readfirstlane(wave);
  __shared__ float ringl[4 * 1024];               // 4 steps x 4 fragments x 1 KiB
  __shared__ float pad[16000];                    // ~80 KB per workgroup in all: two workgroups per CU
  pad[threadIdx.x] = 0.f;
  __amdgpu_buffer_rsrc_t SB;
  {
    unsigned long long b = (unsigned long long)(scr + ((size_t)blockIdx.x * 4 + wave) * 16 * ROW);
    unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    SB = __builtin_amdgcn_make_buffer_rsrc((void *)(((unsigned long long)hi << 32) | lo), 0, 16 * ROW * 4, 0x00020000);
  }
  const unsigned ring_lds = (unsigned)(unsigned long long)(__attribute__((address_space(3))) const void *)ringl;
  f32x4 x[4], y[4];
  for (int t = 0; t < 4; ++t) for (int r = 0; r < 4; ++r) x[t][r] = 0.001f * (float)((lane * 7 + t * 4 + r) % 13);
  Hop xa[2], ya[2];
  xa[0] = split_pair_h(x[0], x[1]); xa[1] = split_pair_h(x[2], x[3]);
  int gs = 0, wo = 0;
  for (int k = 0; k < 3; ++k) { hs_dma(W, wo + uwave * 1024, ring_lds + (unsigned)k * 4096u + (unsigned)uwave * 1024u, lane); wo += 4096; }
  for (int it = 0; it < iters; ++it) {
    wo = 3 * 4096;                                  // the stream ends with a copy of its first three steps
    EPI e0{SB, 0, v16}, e1{SB, 4, v16};
    linear_hs<2, 4, true>(W, ring_lds, uwave, lane, gs, wo, xa, y, ya, e0);
    linear_hs<2, 4, true>(W, ring_lds, uwave, lane, gs, wo, ya, x, xa, e1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float sum = 0.f;
  for (int t = 0; t < 4; ++t) for (int r = 0; r < 4; ++r) sum += x[t][r];
  if (sum == 12345.678f) out[0] = (long long)pad[lane];
}

int main() {
  {
    float *d; hipMalloc((void **)&d, 64);
    hipLaunchKernelGGL(k_denorm, dim3(1), dim3(64), 0, 0, d);
    float h[3]; hipMemcpy(h, d, 12, hipMemcpyDeviceToHost);
    std::printf("f16 MFMA, A subnormal (2^-20) x B 1.0, K = 32: %.6e (exact %.6e)\n", h[0], 32.0 * 9.5367431640625e-07);
    std::printf("f16 MFMA, A 1.0 x B subnormal:                 %.6e\n", h[1]);
    std::printf("bf16 MFMA, A subnormal (2^-129) x B 1.0:       %.6e (exact %.6e)\n", h[2], 32.0 * std::ldexp(1.0, -129));
  }
  {   // accuracy of one linear
    for (int K : {64, 96}) {
      const int N = 64;
      std::vector<double> W((size_t)K * N);
      std::vector<float> in((size_t)32 * K);
      unsigned rs = 12345u + K;
      auto rnd = [&]() { rs = rs * 1664525u + 1013904223u; return ((double)(rs >> 8) / (1 << 24)) * 2.0 - 1.0; };
      for (auto &w : W) w = (double)(float)(rnd() * 0.3);
      for (auto &x : in) x = (float)(rnd() * 2.0);
      std::vector<float> frag;
      append_frag_h(frag, W.data(), K, N, N);
      frag.resize(frag.size() + (size_t)(RINGH + 2) * 256, 0.f);
      float *dW, *din, *dout;
      hipMalloc((void **)&dW, frag.size() * 4); hipMalloc((void **)&din, in.size() * 4); hipMalloc((void **)&dout, (size_t)32 * N * 4);
      hipMemcpy(dW, frag.data(), frag.size() * 4, hipMemcpyHostToDevice); hipMemcpy(din, in.data(), in.size() * 4, hipMemcpyHostToDevice);
      if (K == 64) hipLaunchKernelGGL((k_lin_h<2, 4>), dim3(1), dim3(128), 0, 0, dW, (int)(frag.size() * 4), din, K, dout, N);
      else hipLaunchKernelGGL((k_lin_h<3, 4>), dim3(1), dim3(128), 0, 0, dW, (int)(frag.size() * 4), din, K, dout, N);
      std::vector<float> o((size_t)32 * N);
      hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost);
      double eh = 0, ef = 0, mx = 0;
      for (int r = 0; r < 32; ++r)
        for (int n = 0; n < N; ++n) {
          double ref = 0; float f32 = 0.f;
          for (int k = 0; k < K; ++k) { ref += (double)in[r * K + k] * W[(size_t)k * N + n]; f32 = fmaf(in[r * K + k], (float)W[(size_t)k * N + n], f32); }
          eh = std::max(eh, std::fabs(o[r * N + n] - ref)); ef = std::max(ef, std::fabs((double)f32 - ref)); mx = std::max(mx, std::fabs(ref));
        }
      std::printf("linear %d x %d: max |err| f16x2 %.3e, float32 fmaf chain %.3e (max |value| %.2f)\n", K, N, eh, ef, mx);
    }
  }
  {
    const int eb = 32 + 12;          // two 64 -> 64 linears of 16 entries + wrap copy (the LDS-shared ring runs three steps ahead)
    float *dWb, *scrb; long long *out;
    hipMalloc((void **)&dWb, (size_t)eb * 1024); hipMemset(dWb, 0, (size_t)eb * 1024);
    hipMalloc((void **)&scrb, (size_t)1024 * 4 * 16 * ROW * 4);
    hipMalloc((void **)&out, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int itb = 2000;
    auto run = [&](const char *name, int grid, int G, auto kern) {
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, dWb, eb * 1024, scrb, out, itb);
        hipEventRecord(e1, 0); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double fl = (double)grid * 4 * itb * 2.0 * 64 * 64 * 16 * G * 2;
        if (rep) std::printf("%-56s %7.3f ms  %6.1f f32-equivalent TFLOP/s, %5.0f cycles per 64x64 linear of %d edges per wave at 2.3 GHz\n", name, ms, fl / (ms * 1e-3) * 1e-12,
                             ms * 1e-3 * 2.3e9 / (itb * 2.0), 16 * G);
      }
    };
    run("f16x2 one group, silu+save+split, 1 wave/SIMD", 256, 1, k_rate_h<1, 1, 0>);
    run("f16x2 one group, silu+save+split, 2 waves/SIMD", 512, 1, k_rate_h<1, 2, 0>);
    run("f16x2 two groups, silu+save+split, 1 wave/SIMD", 256, 2, k_rate_h<2, 1, 0>);
    run("f16x2 two groups, silu+save+split, 2 waves/SIMD", 512, 2, k_rate_h<2, 2, 0>);
    run("f16x2 two groups, + 64 live regs, 2 waves/SIMD", 512, 2, k_rate_h<2, 2, 64>);
    run("f16x2 two groups, + 96 live regs, 2 waves/SIMD", 512, 2, k_rate_h<2, 2, 96>);
    run("f16x2 one group, + 96 live regs, 2 waves/SIMD", 512, 1, k_rate_h<1, 2, 96>);
    run("f16x2 LDS-shared ring (4 waves), silu+save+split, 2 waves/SIMD", 512, 1, k_rate_hs<EpiSiluSaveD>);
    run("f16x2 LDS-shared ring (4 waves), no epilogue (split only)", 512, 1, k_rate_hs<EpiNoneX>);
    run("f16x2 one group, packed silu, 1 wave/SIMD", 256, 1, k_rate_h<1, 1, 0, EpiSiluSaveD2>);
    run("f16x2 one group, packed silu, 2 waves/SIMD", 512, 1, k_rate_h<1, 2, 0, EpiSiluSaveD2>);
    run("f16x2 two groups, packed silu, 1 wave/SIMD", 256, 2, k_rate_h<2, 1, 0, EpiSiluSaveD2>);
    run("f16x2 two groups, packed silu, 2 waves/SIMD", 512, 2, k_rate_h<2, 2, 0, EpiSiluSaveD2>);
    run("f16x2 two groups, packed silu + 96 live regs, 2 waves/SIMD", 512, 2, k_rate_h<2, 2, 96, EpiSiluSaveD2>);
    run("f16x2 two groups, no epilogue (split only), 2 waves/SIMD", 512, 2, k_rate_h<2, 2, 0, EpiNoneX>);
    run("f16x2 one group, no epilogue (split only), 2 waves/SIMD", 512, 1, k_rate_h<1, 2, 0, EpiNoneX>);
  }
  return 0;
}
