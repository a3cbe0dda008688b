// Micro-benchmark (NOT part of the product path): what does the bf16 matrix pipe of this box sustain
// on the split kernel's instruction mix?  A wave owns 2 x 4 accumulator tiles (32x32) and issues the
// six piece products per tile and k16 step from register operands (no LDS, no global traffic), or
// -- LDSREAD -- re-reads its 18 operand fragments per k16 step from LDS exactly as the conv kernel does
// (no stores, no barriers).  Operands: random bf16 bit patterns (full mantissa toggling), or zeros.
//   hipcc --offload-arch=gfx950 -O3 -o tools/experiments/build/mfma_bf16_peak tools/experiments/mfma_bf16_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef short bf16x8 __attribute__((vector_size(16)));
typedef float f32x16 __attribute__((vector_size(64)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

template <int THREADS, int OCC, bool LDSREAD>
__global__ void __launch_bounds__(THREADS, OCC * THREADS / 256) k(const bf16x8* __restrict__ rnd, float* out, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDSREAD ? 49152 : 16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  bf16x8 fa[3][2], fb[3][4];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
#pragma unroll
    for (int t = 0; t < 2; ++t) fa[q][t] = rnd[(q * 2 + t) * 64 + lane];
#pragma unroll
    for (int j = 0; j < 4; ++j) fb[q][j] = rnd[(6 + q * 4 + j) * 64 + lane];
  }
  if (LDSREAD) {
    for (int i = tid; i < 49152 / 16; i += THREADS) reinterpret_cast<bf16x8*>(lds)[i] = rnd[i % (18 * 64)];
    __syncthreads();
  }
  const int fr = lane & 31, fg = lane >> 5;
  const int wm = wave & 1, wn = (wave >> 1) & 1;
#define MF(qa, qb, j) { acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[qa][0], fb[qb][j], acc[0][j], 0, 0, 0); \
                        acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[qa][1], fb[qb][j], acc[1][j], 0, 0, 0); }
  for (int it = 0; it < iters; ++it) {
    if (LDSREAD) {
      // planes [piece][k-group 2][256 rows][16 B]: 8 KB per piece plane, A at 0, B at 24 KB
#pragma unroll
      for (int q = 0; q < 3; ++q) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
          fa[q][t] = *reinterpret_cast<const bf16x8*>(lds + q * 8192 + fg * 4096 + (wm * 64 + t * 32 + fr) * 16);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          fb[q][j] = *reinterpret_cast<const bf16x8*>(lds + 24576 + q * 8192 + fg * 4096 + (wn * 128 + j * 32 + fr) * 16);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      MF(2, 0, j) MF(1, 0, j) MF(0, 0, j) MF(1, 1, j) MF(0, 1, j) MF(0, 2, j)
    }
    if (LDSREAD) asm volatile("" ::: "memory");
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[(size_t)blockIdx.x * THREADS + tid] = s;
}

template <int THREADS, int OCC, bool LDSREAD>
void run(const bf16x8* rnd, float* out, const char* tag) {
  const int iters = 4000, grid = 256 * OCC;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k<THREADS, OCC, LDSREAD>), dim3(grid), dim3(THREADS), 0, 0, rnd, out, 200);
  CK(hipDeviceSynchronize());
  float best = 1e30f, sum = 0;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<THREADS, OCC, LDSREAD>), dim3(grid), dim3(THREADS), 0, 0, rnd, out, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best; sum += ms;
  }
  const double flops = (double)grid * (THREADS / 64) * iters * 48.0 * (2.0 * 32 * 32 * 16);
  printf("%-8s threads=%d wg/CU=%d ldsread=%d  best %.2f ms %.0f TF bf16 (= %.1f TF of f32 work)  mean %.0f TF\n", tag, THREADS, OCC,
         (int)LDSREAD, best, flops / best / 1e9, flops / best / 1e9 / 6, flops / (sum / 3) / 1e9);
}

int main() {
  std::vector<unsigned short> h(18 * 64 * 8);
  unsigned long long s = 88172645463325252ull;
  for (auto& v : h) {  // random sign / mantissa, exponent in [2^-8, 2^0): products stay finite
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    v = (unsigned short)(((s >> 20) & 0x807f) | ((119 + ((s >> 40) & 7)) << 7));
  }
  bf16x8 *rnd, *zero; float* out;
  CK(hipMalloc(&rnd, h.size() * 2)); CK(hipMalloc(&zero, h.size() * 2)); CK(hipMalloc(&out, 512 * 512 * 4));
  CK(hipMemcpy(rnd, h.data(), h.size() * 2, hipMemcpyHostToDevice)); CK(hipMemset(zero, 0, h.size() * 2));
  run<256, 1, false>(rnd, out, "random"); run<256, 2, false>(rnd, out, "random"); run<512, 1, false>(rnd, out, "random");
  run<256, 2, false>(zero, out, "zeros");
  run<256, 2, true>(rnd, out, "random"); run<512, 1, true>(rnd, out, "random"); run<256, 2, true>(zero, out, "zeros");
  return 0;
}
