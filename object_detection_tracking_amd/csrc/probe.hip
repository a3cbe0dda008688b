// Sustained-rate probe of the bf16 matrix pipe (measurement support for bench.py's roofline object; no model code calls it).
//
// The dominant kernels of the detector (csrc/conv_split.hip) evaluate every f32 product as six bf16 x bf16
// v_mfma_f32_32x32x16_bf16 products, so their ceiling is "what the bf16 matrix pipe sustains" / 6.  The datasheet figure
// (2.5 PFLOP/s dense) assumes the 2.4 GHz boost clock; under a matrix load on non-zero operands the part clocks to its
// power budget (MI355X_MICROARCH.md, "DVFS give-back").  This probe measures what THIS box sustains: a wave owns the split
// kernels' 2 x 4 accumulator tiles and issues their exact MFMA mix (six piece products per tile and k16 step) from
// register operands filled with random bf16 bit patterns -- no LDS, no global traffic: an upper bound for any kernel
// with that instruction mix -- in back-to-back launches for at least `min_ms` of steady state (the first `warm_ms`
// are run but not counted: after an idle period the clock needs ~10 ms to settle).  `lds_reads` = 1 re-reads the 18
// operand fragments of every k16 step from LDS as the conv kernels do (no stores, no barriers); `lds_reads` = 2 is the
// fp16x2 kernels' mix instead (v_mfma_f32_32x32x16_f16, three products per tile and k16 step, 24 fragment reads per BK = 32 stage).
// The shader clock is read inside the kernel: s_memtime ticks (shader cycles) per s_memrealtime tick (100 MHz).
#include <vector>

#include "../../include/odt.h"
#include "odt_common.hpp"

namespace odt {
namespace {

typedef short bf16x8 __attribute__((vector_size(16)));

#ifndef ODT_HIP_EMULATOR
template <bool LDSREAD>
__global__ void __launch_bounds__(512, 2) mfma_mix_probe_kernel(const bf16x8* __restrict__ rnd, float* __restrict__ out,
                                                                unsigned long long* __restrict__ clocks, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDSREAD ? 49152 : 16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned long long t0 = 0, r0 = 0;
  if (blockIdx.x == 0 && tid == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
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
    for (int i = tid; i < 49152 / 16; i += 512) reinterpret_cast<bf16x8*>(lds)[i] = rnd[i % (18 * 64)];
    __syncthreads();
  }
  const int fr = lane & 31, fg = lane >> 5;
  const int wm = wave & 3, wn = wave >> 2;
#define ODT_PMF(qa, qb, j) { acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[qa][0], fb[qb][j], acc[0][j], 0, 0, 0); \
                             acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[qa][1], fb[qb][j], acc[1][j], 0, 0, 0); }
  for (int it = 0; it < iters; ++it) {
    if (LDSREAD) {
      // the conv kernels' stage image: planes [piece][k-group 2][256 rows][16 B], A at 0, B at 24 KB
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
      ODT_PMF(2, 0, j) ODT_PMF(1, 0, j) ODT_PMF(0, 0, j) ODT_PMF(1, 1, j) ODT_PMF(0, 1, j) ODT_PMF(0, 2, j)
    }
    if (LDSREAD) asm volatile("" ::: "memory");
  }
#undef ODT_PMF
  float s = 0;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[(size_t)blockIdx.x * 512 + tid] = s;
  if (blockIdx.x == 0 && tid == 0) {
    clocks[0] = __builtin_amdgcn_s_memtime() - t0;
    clocks[1] = __builtin_amdgcn_s_memrealtime() - r0;
  }
}

// the fp16x2 kernels' mix (csrc/conv_h2.hip): three f16 piece products per tile and k16 step, a stage of two k-steps with its
// 24 operand fragments (2 pieces x (2 + 4) x 2 k-steps) re-read from LDS
typedef _Float16 f16x8p __attribute__((ext_vector_type(8)));
__global__ void __launch_bounds__(512, 2) mfma_h2_mix_probe_kernel(const f16x8p* __restrict__ rnd, float* __restrict__ out,
                                                                   unsigned long long* __restrict__ clocks, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned long long t0 = 0, r0 = 0;
  if (blockIdx.x == 0 && tid == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  for (int i = tid; i < 65536 / 16; i += 512) reinterpret_cast<f16x8p*>(lds)[i] = rnd[i % (18 * 64)];
  __syncthreads();
  const int fr = lane & 31, fg = lane >> 5;
  const int wm = wave & 3, wn = wave >> 2;
  f16x8p fa[2][2], fb[2][4];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      // the conv kernels' stage image: planes [piece][k-group 4][256 rows][16 B], A at 0, B at 32 KB
#pragma unroll
      for (int q = 0; q < 2; ++q) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
          fa[q][t] = *reinterpret_cast<const f16x8p*>(lds + q * 16384 + (2 * ks + fg) * 4096 + (wm * 64 + t * 32 + fr) * 16);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          fb[q][j] = *reinterpret_cast<const f16x8p*>(lds + 32768 + q * 16384 + (2 * ks + fg) * 4096 + (wn * 128 + j * 32 + fr) * 16);
      }
#define ODT_PMH(qa, qb, j) { acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[qa][0], fb[qb][j], acc[0][j], 0, 0, 0); \
                             acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[qa][1], fb[qb][j], acc[1][j], 0, 0, 0); }
#pragma unroll
      for (int j = 0; j < 4; ++j) { ODT_PMH(1, 0, j) ODT_PMH(0, 1, j) ODT_PMH(0, 0, j) }
#undef ODT_PMH
      asm volatile("" ::: "memory");
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[(size_t)blockIdx.x * 512 + tid] = s;
  if (blockIdx.x == 0 && tid == 0) {
    clocks[0] = __builtin_amdgcn_s_memtime() - t0;
    clocks[1] = __builtin_amdgcn_s_memrealtime() - r0;
  }
}
#endif

}  // namespace
}  // namespace odt

using namespace odt;

extern "C" int odt_probe_mfma_bf16(int device, double warm_ms, double min_ms, int lds_reads, double* tflops_bf16,
                                   double* clock_ghz, double* measured_ms, int* launches) {
#ifdef ODT_HIP_EMULATOR
  (void)device; (void)warm_ms; (void)min_ms; (void)lds_reads; (void)tflops_bf16; (void)clock_ghz; (void)measured_ms; (void)launches;
  set_error("odt_probe_mfma_bf16: a hardware measurement (not available in the simulator build)");
  return 1;
#else
  ODT_CHECK(tflops_bf16 != nullptr && min_ms > 0 && warm_ms >= 0, "odt_probe_mfma_bf16: bad argument");
  int n = 0;
  ODT_HIP(hipGetDeviceCount(&n));
  ODT_CHECK(device >= 0 && device < n, "odt_probe_mfma_bf16: no such device");
  ODT_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  ODT_HIP(hipGetDeviceProperties(&prop, device));
  const int grid = prop.multiProcessorCount;            // one 8-wave workgroup per CU, as the conv kernels run
  std::vector<unsigned short> h(18 * 64 * 8);
  unsigned long long s = 88172645463325252ull;
  for (auto& v : h) {      // random sign / mantissa, exponent in [2^-8, 2^0): every mantissa bit toggles, sums stay finite
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    if (lds_reads == 2) v = (unsigned short)(((s >> 20) & 0x83ff) | ((7 + ((s >> 40) & 7)) << 10));      // f16 bit patterns
    else v = (unsigned short)(((s >> 20) & 0x807f) | ((119 + ((s >> 40) & 7)) << 7));
  }
  bf16x8* rnd = nullptr; float* out = nullptr; unsigned long long* clk = nullptr;
  ODT_HIP(hipMalloc((void**)&rnd, h.size() * 2));
  ODT_HIP(hipMalloc((void**)&out, (size_t)grid * 512 * 4));
  ODT_HIP(hipMalloc((void**)&clk, 16));
  ODT_HIP(hipMemcpy(rnd, h.data(), h.size() * 2, hipMemcpyHostToDevice));
  hipStream_t st; ODT_HIP(hipStreamCreate(&st));
  const int iters = 4000;                                // ~7-8 ms per launch at the sustained rate
  const double flop_launch = (double)grid * 8 * iters * 48.0 * (2.0 * 32 * 32 * 16);      // (both mixes: 48 MFMAs per wave and iteration)
  auto launch = [&]() {
    if (lds_reads == 2) hipLaunchKernelGGL(mfma_h2_mix_probe_kernel, dim3(grid), dim3(512), 0, st, (const f16x8p*)rnd, out, clk, iters);
    else if (lds_reads) hipLaunchKernelGGL((mfma_mix_probe_kernel<true>), dim3(grid), dim3(512), 0, st, rnd, out, clk, iters);
    else hipLaunchKernelGGL((mfma_mix_probe_kernel<false>), dim3(grid), dim3(512), 0, st, rnd, out, clk, iters);
  };
  hipEvent_t e0, e1, e2;
  ODT_HIP(hipEventCreate(&e0)); ODT_HIP(hipEventCreate(&e1)); ODT_HIP(hipEventCreate(&e2));
  // warm-up: launches until `warm_ms` have passed (not counted)
  ODT_HIP(hipEventRecord(e0, st));
  float ms = 0;
  int nwarm = 0;
  do {
    for (int i = 0; i < 4; ++i) launch();
    nwarm += 4;
    ODT_HIP(hipEventRecord(e1, st));
    ODT_HIP(hipEventSynchronize(e1));
    ODT_HIP(hipEventElapsedTime(&ms, e0, e1));
  } while (ms < warm_ms && nwarm < 4096);
  // steady state: back-to-back launches, timed as one region, until `min_ms`
  int cnt = 0;
  float total = 0;
  ODT_HIP(hipEventRecord(e1, st));
  do {
    for (int i = 0; i < 8; ++i) launch();
    cnt += 8;
    ODT_HIP(hipEventRecord(e2, st));
    ODT_HIP(hipEventSynchronize(e2));
    ODT_HIP(hipEventElapsedTime(&total, e1, e2));
  } while (total < min_ms && cnt < 8192);
  ODT_HIP(hipGetLastError());
  unsigned long long c[2] = {0, 0};
  ODT_HIP(hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost));
  *tflops_bf16 = flop_launch * cnt / (total * 1e-3) / 1e12;
  if (clock_ghz) *clock_ghz = c[1] ? (double)c[0] / (double)c[1] * 0.1 : 0.0;       // shader cycles per 10 ns
  if (measured_ms) *measured_ms = total;
  if (launches) *launches = cnt;
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipEventDestroy(e2);
  (void)hipStreamDestroy(st);
  (void)hipFree(rnd); (void)hipFree(out); (void)hipFree(clk);
  return 0;
#endif
}
