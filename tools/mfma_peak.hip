// Micro-benchmark: sustained v_mfma_f32_32x32x2_f32 rate on this box (practical ceiling for the
// conv kernel's roofline; the datasheet 157.3 TF assumes 2.4 GHz).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((vector_size(64)));
template <int NACC>
__global__ void __launch_bounds__(256) k(float* out, int iters, float a, float b, const float* rnd) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float x = a + threadIdx.x, y = b;
  if (rnd) { x = rnd[threadIdx.x]; y = rnd[256 + threadIdx.x]; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC> void run(int blocks_per_cu, const char* tag, const float* rnd = nullptr) {
  float* out; hipMalloc(&out, 256 * 256 * 8 * 4);
  int iters = 20000, grid = 256 * blocks_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256), 0, 0, out, 100, 1.f, 2.f, rnd);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 2.f, rnd);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flops = (double)grid * 4 /*waves*/ * iters * 4.0 * NACC * (2.0 * 32 * 32 * 2);
  printf("%s: NACC=%d blocks/CU=%d  %.2f ms  %.1f TFLOP/s\n", tag, NACC, blocks_per_cu, ms, flops / ms / 1e9);
  hipFree(out);
}
#include <cstdlib>
int main() {
  float h[512]; for (int i = 0; i < 512; ++i) h[i] = (float)rand() / RAND_MAX * 2.f - 1.f;
  float* rnd; hipMalloc(&rnd, 2048); hipMemcpy(rnd, h, 2048, hipMemcpyHostToDevice);
  run<4>(2, "random operands", rnd); run<4>(1, "random operands", rnd);
  run<4>(1, "mfma_f32_32x32x2"); run<4>(2, "mfma_f32_32x32x2"); run<1>(1, "mfma_f32_32x32x2"); run<2>(2, "mfma_f32_32x32x2");
  return 0;
}
