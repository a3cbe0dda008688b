// Probe: cost of executing N straight-line instructions for the first dispatch wave of a launch
// vs later workgroups (instruction-cache cold start per dispatch?).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int N>
__global__ void __launch_bounds__(256) k(unsigned long long* t, float* out, float a) {
  unsigned long long t0 = wall_clock64();
  float x = a + threadIdx.x;
#pragma unroll
  for (int i = 0; i < N; ++i) x = x * 1.0001f + (float)i;   // 2 VALU per step, fully unrolled
  unsigned long long t1 = wall_clock64();
  if (threadIdx.x == 0) { t[blockIdx.x * 2] = t0; t[blockIdx.x * 2 + 1] = t1; }
  out[blockIdx.x * 256 + threadIdx.x] = x;
}
template <int N> void run() {
  const int grid = 2048;
  unsigned long long* t; float* out;
  hipMalloc(&t, grid * 16); hipMalloc(&out, grid * 256 * 4);
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(k<N>, dim3(grid), dim3(256), 0, 0, t, out, 1.f);
  }
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(grid * 2);
  hipMemcpy(h.data(), t, grid * 16, hipMemcpyDeviceToHost);
  unsigned long long t0 = ~0ull; for (int b = 0; b < grid; ++b) if (h[2 * b] < t0) t0 = h[2 * b];
  double fa = 0, fb = 0; int na = 0, nb = 0;
  for (int b = 0; b < grid; ++b) { double d = (double)(h[2 * b + 1] - h[2 * b]) / 100.0; if (h[2 * b] - t0 < 300) { fa += d; ++na; } else { fb += d; ++nb; } }
  printf("N=%5d unrolled steps: first wave (%d blocks) %.2f us per block, later (%d blocks) %.2f us\n", N, na, fa / na, nb, nb ? fb / nb : 0.0);
  hipFree(t); hipFree(out);
}
int main() { run<256>(); run<1024>(); run<4096>(); return 0; }
