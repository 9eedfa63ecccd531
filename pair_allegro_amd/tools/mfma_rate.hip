// Micro-benchmark of the fused kernel's streamed linear (linear_s): cycles per MFMA per wave with 1 or 2 waves per
// SIMD, with and without the element-wise epilogues.  Not part of the product; build on the GPU box:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics -I pair_allegro_amd/csrc pair_allegro_amd/tools/mfma_rate.hip \
//         pair_allegro_amd/csrc/{allegro_hip,prims,neigh,edges,model_io}.o -o gpurun_out/mfma_rate
#include "../csrc/fused.hip"

template <int EPI, int OCC>
__global__ void __launch_bounds__(256, OCC) k_rate(const float *W, int wbytes, float *scr, long long *out, int iters) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, v16 = lane * 16;
  __shared__ float pad[OCC == 2 ? 20000 : 40000];      // 80 / 160 KB: pins the number of workgroups per CU to OCC
  pad[threadIdx.x] = 0.f;
  __amdgpu_buffer_rsrc_t WB = __builtin_amdgcn_make_buffer_rsrc((void *)W, 0, wbytes, 0x00020000);
  __amdgpu_buffer_rsrc_t SB = __builtin_amdgcn_make_buffer_rsrc((void *)(scr + ((size_t)blockIdx.x * 4 + wave) * 16 * ROW), 0, 16 * ROW * 4, 0x00020000);
  f32x4 x[4], y[4], ring[RING];
  for (int t = 0; t < 4; ++t) for (int r = 0; r < 4; ++r) x[t][r] = 0.001f * (float)((lane * 7 + t * 4 + r) % 13);
  int wp = 0;
  ring_prime(WB, wp, v16, ring);
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    wp = 0;
    if (EPI == 0) {
      linear_s<4, 4, false, 0>(WB, wp, x, y, v16, ring, EpiNone{});
      linear_s<4, 4, false, 0>(WB, wp, y, x, v16, ring, EpiNone{});
    } else if (EPI == 1) {
      linear_s<4, 4, false, 0>(WB, wp, x, y, v16, ring, EpiSiluSaveD{SB, 0, v16});
      linear_s<4, 4, false, 0>(WB, wp, y, x, v16, ring, EpiSiluSaveD{SB, 4, v16});
    } else {
      linear_s<4, 4, false, 0>(WB, wp, x, y, v16, ring, EpiSave{SB, 0, v16});
      linear_s<4, 4, false, 0>(WB, wp, y, x, v16, ring, EpiSave{SB, 4, v16});
    }
  }
  const long long t1 = clock64();
  float sum = 0.f;
  for (int t = 0; t < 4; ++t) for (int r = 0; r < 4; ++r) sum += x[t][r];
  if (sum == 12345.678f) out[0] = (long long)pad[lane];
  if (lane == 0) out[blockIdx.x * 4 + wave] = t1 - t0;
}

int main() {
  const int entries = 32 + RING;
  std::vector<float> w((size_t)entries * 256);
  for (size_t i = 0; i < w.size(); ++i) w[i] = 0.01f * (float)((i * 2654435761u >> 7) % 17) - 0.08f;
  for (size_t i = 0; i < (size_t)RING * 256; ++i) w[(size_t)32 * 256 + i] = w[i];
  float *dW, *scr; long long *out;
  const int maxg = 1024;
  hipMalloc((void **)&dW, w.size() * 4); hipMemcpy(dW, w.data(), w.size() * 4, hipMemcpyHostToDevice);
  hipMalloc((void **)&scr, (size_t)maxg * 4 * 16 * ROW * 4);
  hipMalloc((void **)&out, (size_t)maxg * 4 * 8);
  const int iters = 2000;
  std::vector<long long> h((size_t)maxg * 4);
  hipEvent_t ev0, ev1;
  hipEventCreate(&ev0); hipEventCreate(&ev1);
  auto report = [&](const char *name, int grid) {
    hipEventRecord(ev1, 0);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, ev0, ev1);
    std::printf("%7.3f ms %6.1f TFLOP/s | ", ms, (double)grid * 4 * iters * 128.0 * 2048.0 / (ms * 1e-3) * 1e-12);
    hipMemcpy(h.data(), out, (size_t)grid * 4 * 8, hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < grid * 4; ++i) s += (double)h[i];
    std::printf("%-34s grid %4d: %.2f cycles per MFMA per wave\n", name, grid, s / (grid * 4) / (iters * 128.0));
  };
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(ev0, 0); hipLaunchKernelGGL((k_rate<0, 1>), dim3(256), dim3(256), 0, 0, dW, (int)(w.size() * 4), scr, out, iters); report("no epilogue, 1 wave/SIMD", 256);
    hipEventRecord(ev0, 0); hipLaunchKernelGGL((k_rate<0, 2>), dim3(512), dim3(256), 0, 0, dW, (int)(w.size() * 4), scr, out, iters); report("no epilogue, 2 waves/SIMD", 512);
    hipEventRecord(ev0, 0); hipLaunchKernelGGL((k_rate<2, 1>), dim3(256), dim3(256), 0, 0, dW, (int)(w.size() * 4), scr, out, iters); report("save rows, 1 wave/SIMD", 256);
    hipEventRecord(ev0, 0); hipLaunchKernelGGL((k_rate<2, 2>), dim3(512), dim3(256), 0, 0, dW, (int)(w.size() * 4), scr, out, iters); report("save rows, 2 waves/SIMD", 512);
    hipEventRecord(ev0, 0); hipLaunchKernelGGL((k_rate<1, 1>), dim3(256), dim3(256), 0, 0, dW, (int)(w.size() * 4), scr, out, iters); report("silu + save d rows, 1 wave/SIMD", 256);
    hipEventRecord(ev0, 0); hipLaunchKernelGGL((k_rate<1, 2>), dim3(512), dim3(256), 0, 0, dW, (int)(w.size() * 4), scr, out, iters); report("silu + save d rows, 2 waves/SIMD", 512);
  }
  return 0;
}
