// Reproducer for the gfx950 store-data hazard behind the wrong saved rows of rounds 2-3 (csrc/fused_common.h: bstore).
//
//   buffer_store_dwordx4 v[10:13], voff, rsrc, SOFF offen      ; 16 bytes of data per lane
//   v_mov_b32 v10, poison                                      ; next issue slot: overwrites a data register
//
// The store reads its data registers after it has issued, so the overwrite can win and the store writes the poison.  The ISA manuals
// (and LLVM's hazard recognizer, GCNHazardRecognizer::createsVALUHazard) say the hazard does not exist when SOFF is an SGPR; this
// program counts poisoned dwords for SOFF = SGPR / SOFF = 0 and for 0, 1, 2 wait states between the store and the overwrite.
// A few stores are issued first so that the vector-memory queue is not empty (an idle queue takes the data at once).
//
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O2 -o /tmp/store_hazard pair_allegro_amd/tools/store_hazard.hip && /tmp/store_hazard
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef int i32x4 __attribute__((ext_vector_type(4)));
static constexpr int ROWS = 8;            // 1 KiB rows per wave and iteration: ROWS - 1 queue fillers + the probed store
static constexpr float GOOD = 1.0f, POISON = -7.0f;

#define FILL(off) "buffer_store_dwordx4 v[14:17], %[vo], %[rs], 0 offen offset:" #off "\n"
#define BODY(SOFF, PAD) BODY2(SOFF, PAD, "v_mov_b32 v10, %[p]\n v_mov_b32 v11, %[p]\n v_mov_b32 v12, %[p]\n v_mov_b32 v13, %[p]\n")
#define BODY2(SOFF, PAD, OVER)                                                                                                   \
  asm volatile("v_mov_b32 v10, %[g]\n v_mov_b32 v11, %[g]\n v_mov_b32 v12, %[g]\n v_mov_b32 v13, %[g]\n"                       \
               "v_mov_b32 v14, %[g]\n v_mov_b32 v15, %[g]\n v_mov_b32 v16, %[g]\n v_mov_b32 v17, %[g]\n s_nop 4\n"             \
               FILL(0) FILL(1024) FILL(2048) FILL(3072) "s_add_u32 %[so2], %[so], 4096\n"                                    \
               "buffer_store_dwordx4 v[14:17], %[vo], %[rs], %[so2] offen\n"                                                  \
               "buffer_store_dwordx4 v[14:17], %[vo], %[rs], %[so2] offen offset:1024\n"                                      \
               "buffer_store_dwordx4 v[14:17], %[vo], %[rs], %[so2] offen offset:2048\n"                                      \
               "buffer_store_dwordx4 v[10:13], %[vo], %[rs], " SOFF " offen offset:3072\n" /* the probed store */             \
               PAD OVER                                                                                                      \
               "s_waitcnt vmcnt(0)\n"                                                                                        \
               : [so2] "=&s"(so2)                                                                                            \
               : [g] "v"(good), [p] "v"(poison), [vo] "v"(voff), [rs] "s"(rs), [so] "s"(so)                                   \
               : "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "memory")

template <int MODE> __global__ void __launch_bounds__(256) k_probe(float *out, int iters, long long wave_floats) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float *base = out + ((size_t)blockIdx.x * 4 + wave) * wave_floats;
  const unsigned long long b = (unsigned long long)base;
  i32x4 rs;
  rs[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
  rs[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32));
  rs[2] = (int)(wave_floats * 4);
  rs[3] = 0x00020000;
  const int voff = lane * 16;
  const float good = GOOD, poison = POISON;
  for (int it = 0; it < iters; ++it) {
    int so = __builtin_amdgcn_readfirstlane(it * ROWS * 1024), so2;
    // MODE: bit 2 = soffset field is the constant 0 (the row offset then sits in the immediate), bits 0-1 = wait states after the store
    if (MODE == 0) BODY("%[so2]", "");
    if (MODE == 1) BODY("%[so2]", "s_nop 0\n");
    if (MODE == 2) BODY("%[so2]", "s_nop 1\n");
    // the overwrite is an MFMA (the accumulator stores of a linear are followed by the next tile pair's first MFMA): all four registers become poison * 4 * 0 + ... = A x B with A = poison, B = 1
    if (MODE == 8) BODY2("%[so2]", "", "v_mfma_f32_16x16x4_f32 v[10:13], %[p], %[g], 0\n s_nop 7\n");
    if (MODE == 9) BODY2("%[so2]", "s_nop 0\n", "v_mfma_f32_16x16x4_f32 v[10:13], %[p], %[g], 0\n s_nop 7\n");
    if (MODE == 4) { BODY("0", ""); }
    if (MODE == 5) { BODY("0", "s_nop 0\n"); }
    if (MODE == 6) { BODY("0", "s_nop 1\n"); }
  }
}

template <int MODE> static void run(const char *what) {
  const int grid = 1024, iters = 16;
  const long long wave_floats = (long long)iters * ROWS * 256 + 2048;
  const size_t n = (size_t)grid * 4 * wave_floats;
  float *d;
  (void)hipMalloc(&d, n * 4);
  long long bad[4] = {0, 0, 0, 0}, probed = 0;
  for (int rep = 0; rep < 5; ++rep) {
    (void)hipMemset(d, 0, n * 4);
    hipLaunchKernelGGL(k_probe<MODE>, dim3(grid), dim3(256), 0, 0, d, iters, wave_floats);
    (void)hipDeviceSynchronize();
    std::vector<float> h(n);
    (void)hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
    for (size_t w = 0; w < (size_t)grid * 4; ++w)
      for (int it = 0; it < iters; ++it) {
        // the probed row: MODE < 4: soffset so + 4096, immediate 3072 -> row 7 of the iteration; MODE >= 4: soffset 0, immediate 3072 -> row 3 of iteration 0 (rewritten each iteration)
        const size_t row = (MODE < 4) ? (size_t)it * ROWS + 7 : 3;
        const float *p = h.data() + w * wave_floats + row * 256;
        for (int k = 0; k < 256; ++k) { ++probed; if (p[k] != GOOD) ++bad[k & 3]; }
      }
  }
  // data register r is overwritten r issue slots after the one that follows the store (and the pad)
  printf("%-46s poisoned dwords by data register: %lld %lld %lld %lld  of %lld each\n", what, bad[0], bad[1], bad[2], bad[3], probed / 4);
  (void)hipFree(d);
}

int main() {
  run<0>("soffset = SGPR, overwrite in the next slot");
  run<1>("soffset = SGPR, s_nop 0 (1 wait state)");
  run<2>("soffset = SGPR, s_nop 1 (2 wait states)");
  run<8>("soffset = SGPR, MFMA writes the data registers");
  run<9>("soffset = SGPR, s_nop 0, then the MFMA");
  run<4>("soffset = 0,    overwrite in the next slot");
  run<5>("soffset = 0,    s_nop 0 (1 wait state)");
  run<6>("soffset = 0,    s_nop 1 (2 wait states)");
  return 0;
}
