// Structure probe (NOT part of the product path) for the round-2 split kernel: f32 GEMM on the bf16
// matrix pipe (bf16x3 split, six piece products), 8 waves per workgroup, ONE workgroup per CU,
// BK = 16 per LDS stage, three-stage LDS ring, weights (pre-split image) copied global -> LDS by
// LDS-DMA (buffer_load ... lds), activations f32 -> registers -> split -> LDS, one barrier per stage.
//   C[M,N] = A[M,K] (f32, k contiguous) x B (pre-imaged bf16 pieces)
// Tile <WM, WN, TN>: BM = 64 WM, BN = 32 TN WN, wave tile 64 x 32 TN.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/experiments/build/split_gemm3 tools/experiments/split_gemm3.hip
// Run:   split_gemm3 M N K [flags]      flags: 1 = no split arithmetic, 2 = no A loads after the prologue,
//                                              4 = no B DMA after the prologue, 8 = no A stores
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef short bf16x8 __attribute__((vector_size(16)));
typedef float f32x16 __attribute__((vector_size(64)));
typedef float f32x4 __attribute__((vector_size(16)));
typedef unsigned int u32x4 __attribute__((vector_size(16)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

namespace {

__device__ __forceinline__ unsigned cvt_pk_bf16(float a0, float a1) {
  const f32x2_t v = {a0, a1};
  const bf16x2_t r = __builtin_convertvector(v, bf16x2_t);
  return *reinterpret_cast<const unsigned*>(&r);
}
__device__ __forceinline__ void split2(float a0, float a1, unsigned& hi, unsigned& mid, unsigned& lo) {
  hi = cvt_pk_bf16(a0, a1);
  const float r0 = a0 - __uint_as_float(hi << 16);
  const float r1 = a1 - __uint_as_float(hi & 0xffff0000u);
  mid = cvt_pk_bf16(r0, r1);
  const float s0 = r0 - __uint_as_float(mid << 16);
  const float s1 = r1 - __uint_as_float(mid & 0xffff0000u);
  lo = cvt_pk_bf16(s0, s1);
}

template <int WM, int WN, int TN>
struct Cfg {
  static constexpr int BM = 64 * WM, BN = 32 * TN * WN;
  static constexpr int AKG = BM * 16 + 64, APL = 2 * AKG;   // A: [piece][k-group 2][row][8 bf16], 64-B pad per k-group
  static constexpr int BKG = BN * 16, BPL = 2 * BKG;        // B: linear image of the DMA
  static constexpr int STAGE_B = 3 * BPL;                   // bytes of weight image per stage
  static constexpr int STAGE = 3 * APL + STAGE_B;
  static constexpr int NDMA = STAGE_B / (512 * 16);         // LDS-DMA instructions per thread and stage
  static_assert(STAGE_B % (512 * 16) == 0, "whole DMA instructions");
  static_assert(3 * STAGE <= 160 * 1024, "LDS ring");
};

template <int WM, int WN, int TN, int FLAGS, int ALEAD>
__global__ void __launch_bounds__(512, 2) split_gemm3_kernel(const float* __restrict__ A, const unsigned char* __restrict__ Bimg,
                                                            float* __restrict__ C, int M, int N, int K, unsigned long long* __restrict__ clk) {
  using G = Cfg<WM, WN, TN>;
  unsigned long long t0 = 0, r0 = 0;
  if (clk != nullptr) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
  constexpr int BM = G::BM, BN = G::BN, AKG = G::AKG, APL = G::APL, BKG = G::BKG, BPL = G::BPL;
  constexpr int STAGE = G::STAGE, STAGE_B = G::STAGE_B, NDMA = G::NDMA;
  static_assert(WM * WN == 8, "8 waves");
  __shared__ __attribute__((aligned(16))) unsigned char lds[3 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int ntn = N / BN;
  int wg = (int)blockIdx.x;
  {
    const int nwg = (int)gridDim.x, xcd = wg & 7, q = nwg >> 3, r = nwg & 7;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (wg >> 3);
  }
  const int mt = wg / ntn, nt = wg - mt * ntn;
  const int m0 = mt * BM, n0 = nt * BN;
  const int nsteps = K >> 4;

  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)((unsigned)M * K * 4u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_b =
      __builtin_amdgcn_make_buffer_rsrc((void*)Bimg, 0, (int)((unsigned)ntn * nsteps * (unsigned)STAGE_B), 0x00020000);

  // A: thread -> (row (t >> 2) + 128 j, 16-byte column t & 3): 4 lanes cover the 64 contiguous bytes of a row's stage
  constexpr int RA = BM / 128;                    // rows per thread and stage
  typedef unsigned int u32x2 __attribute__((vector_size(8)));
  const int a_c = tid & 3, a_r = tid >> 2;
  unsigned a_off[RA];
#pragma unroll
  for (int j = 0; j < RA; ++j) a_off[j] = ((unsigned)(m0 + a_r + 128 * j) * K + a_c * 4) * 4u;
  f32x4 ga[ALEAD][RA];
  int l_a = 0;                                     // byte offset of the A load stream along k
  auto load_a = [&](int set) {
#pragma unroll
    for (int j = 0; j < RA; ++j) ga[set][j] = (f32x4)__builtin_amdgcn_raw_buffer_load_b128(rs_a, (int)a_off[j], l_a, 0);
    l_a += 64;
  };
  auto store_a = [&](int st, int set) {
#pragma unroll
    for (int j = 0; j < RA; ++j) {
      unsigned h0, m0_, l0, h1, m1, l1;
      if (FLAGS & 1) {
        h0 = m0_ = l0 = __float_as_uint(ga[set][j][0]); h1 = m1 = l1 = __float_as_uint(ga[set][j][2]);
      } else {
        split2(ga[set][j][0], ga[set][j][1], h0, m0_, l0);
        split2(ga[set][j][2], ga[set][j][3], h1, m1, l1);
      }
      if (!(FLAGS & 8)) {
        unsigned char* d = lds + st + (a_c >> 1) * AKG + (a_r + 128 * j) * 16 + (a_c & 1) * 8;
        *reinterpret_cast<u32x2*>(d + 0 * APL) = u32x2{h0, h1};
        *reinterpret_cast<u32x2*>(d + 1 * APL) = u32x2{m0_, m1};
        *reinterpret_cast<u32x2*>(d + 2 * APL) = u32x2{l0, l1};
      } else {
        asm volatile("" :: "v"(h0), "v"(m1), "v"(l0), "v"(h1), "v"(m0_), "v"(l1));
      }
    }
  };
  // B: wave w, instruction i copies the 1-KB chunk i * 8 + w of the stage image
  unsigned l_b = (unsigned)nt * (unsigned)nsteps * (unsigned)STAGE_B;
  auto dma_b = [&](int st) {
#pragma unroll
    for (int i = 0; i < NDMA; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b, (__attribute__((address_space(3))) void*)(lds + st + 3 * APL + (i * 8 + wave) * 1024),
                                               16, lane * 16 + (i * 8 + wave) * 1024, (int)l_b, 0, 0);
    l_b += (unsigned)STAGE_B;
  };

  f32x16 acc[2][TN];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int fr = lane & 31, fg = lane >> 5;
  // ---- prologue: stage 0 complete, stage 1's weights in flight, A of step 1 in registers
  // register set of step c: c % ALEAD; step 1 (and 2 with ALEAD = 2) are in flight when the loop starts
  load_a(0);
  dma_b(0);
  if (nsteps > 1) dma_b(STAGE);
  store_a(0, 0);
  if (nsteps > 1) load_a(ALEAD == 2 ? 1 : 0);
  if (ALEAD == 2 && nsteps > 2) load_a(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // simple prologue: everything landed
  __syncthreads();

  bf16x8 fa[3][2], fb[3];
  const int a_rd = fg * AKG + (wm * 64 + fr) * 16;
  const int b_rd = 3 * APL + fg * BKG + (wn * TN * 32 + fr) * 16;
  auto rdA = [&](int st, int q) {
#pragma unroll
    for (int t = 0; t < 2; ++t) fa[q][t] = *reinterpret_cast<const bf16x8*>(lds + st + q * APL + a_rd + t * 512);
  };
  auto rdB = [&](int st, int q, int j) {
    fb[q] = *reinterpret_cast<const bf16x8*>(lds + st + q * BPL + b_rd + j * 512);
  };
#pragma unroll
  for (int q = 0; q < 3; ++q) rdA(0, q);
#pragma unroll
  for (int q = 0; q < 3; ++q) rdB(0, q, 0);

#define MF(qa, qb, j) { acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[qa][0], fb[qb], acc[0][j], 0, 0, 0); \
                        acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[qa][1], fb[qb], acc[1][j], 0, 0, 0); }
#define FENCE() __builtin_amdgcn_sched_barrier(0)
  int st_cur = 0, st_nxt = STAGE, st_nn = 2 * STAGE;
  // One step.  NEXT: step c+1 exists (its A: registers -> LDS; read its first fragments);
  // PRE: step c+2 exists (fetch its A, start its weight DMA).
  // SET: register set holding A of step c+1 (ALEAD = 2: the other set holds step c+2, and step c+3 is fetched into SET)
  auto step = [&](auto NEXT, auto PRE, auto PRE3, auto SETC) {
    constexpr bool next = decltype(NEXT)::value, pre = decltype(PRE)::value, pre3 = decltype(PRE3)::value;
    constexpr int set = decltype(SETC)::value;
    FENCE();
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const bool last = j == TN - 1;
      if (last) {
        // stage nxt must be complete before its first fragment reads below: own LDS stores and own DMA
        // (issued one step ago; the DMA of step c+2 issued in this step may stay in flight), then the barrier
        if constexpr (pre) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(NDMA) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        FENCE();
      }
      MF(2, 0, j); FENCE();
      if (last) { if constexpr (next) rdA(st_nxt, 2); }
      else if (j == 0) { if constexpr (next) store_a(st_nxt, set); }      // A of step c+1: registers -> LDS
      FENCE();
      MF(1, 0, j); MF(0, 0, j); FENCE();
      if (!last) rdB(st_cur, 0, j + 1); else if constexpr (next) rdB(st_nxt, 0, 0);
      FENCE();
      MF(1, 1, j); FENCE();
      if (last) { if constexpr (next) rdA(st_nxt, 1); }
      else if (j == 0) { if constexpr (ALEAD == 2 ? pre3 : pre) { if (!(FLAGS & 2)) load_a(set); } }   // A of step c+1+ALEAD (registers are free again)
      FENCE();
      MF(0, 1, j); FENCE();
      if (!last) rdB(st_cur, 1, j + 1); else if constexpr (next) rdB(st_nxt, 1, 0);
      if (j == 1) { if constexpr (pre) { if (!(FLAGS & 4)) dma_b(st_nn); } }    // weights of step c+2 -> stage nn
      FENCE();
      MF(0, 2, j); FENCE();
      if (!last) rdB(st_cur, 2, j + 1); else if constexpr (next) { rdA(st_nxt, 0); rdB(st_nxt, 2, 0); }
      FENCE();
    }
    const int t = st_cur; st_cur = st_nxt; st_nxt = st_nn; st_nn = t;
  };
  {
    using T = std::true_type; using F = std::false_type;
    using S0 = std::integral_constant<int, 0>; using S1 = std::integral_constant<int, ALEAD == 2 ? 1 : 0>;
    int c = 0;
    // A of step c+1 sits in set (c+1) % ALEAD: even c -> set 1, odd c -> set 0 (ALEAD = 2).  nsteps is even and >= 4
    // here (K % 32 == 0, K >= 64); straight-line tail so that no MFMA sits in a conditional arm.
    for (; c + 6 <= nsteps; c += 2) { step(T{}, T{}, T{}, S1{}); step(T{}, T{}, T{}, S0{}); }
    step(T{}, T{}, T{}, S1{});
    step(T{}, T{}, F{}, S0{});
    step(T{}, F{}, F{}, S1{});
    step(F{}, F{}, F{}, S0{});
  }
#undef MF
#undef FENCE
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + i * 32 + (r >> 2) * 8 + fg * 4 + (r & 3);
        const int col = n0 + wn * TN * 32 + j * 32 + fr;
        C[(size_t)row * N + col] = acc[i][j][r];
      }
  if (clk != nullptr && tid == 0) {     // shader clock over this workgroup's life: s_memtime ticks per 100 MHz s_memrealtime tick
    clk[blockIdx.x * 2] = __builtin_amdgcn_s_memtime() - t0;
    clk[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime() - r0;
  }
}

void host_split(float x, uint16_t* p) {   // round-to-nearest-even pieces, as v_cvt_pk_bf16_f32
  auto rne = [](float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16); };
  auto up = [](uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; };
  p[0] = rne(x); float r = x - up(p[0]);
  p[1] = rne(r); float s = r - up(p[1]);
  p[2] = rne(s);
}

}  // namespace

template <int WM, int WN, int TN, int ALEAD>
int run(int M, int N, int K, int flags, int reps) {
  using G = Cfg<WM, WN, TN>;
  constexpr int BM = G::BM, BN = G::BN;
  if (M % BM || N % BN || K % 16) { fprintf(stderr, "shape not tileable by %dx%d\n", BM, BN); return 2; }
  std::vector<float> a((size_t)M * K), b((size_t)K * N);
  uint64_t s = 88172645463325252ull;
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (float)((s >> 40) / 16777216.0 * 2.0 - 1.0); };
  for (auto& v : a) v = rnd();
  for (auto& v : b) v = rnd() * 0.05f;
  const int nsteps = K / 16, ntn = N / BN;
  std::vector<uint16_t> img((size_t)3 * N * K);
  for (int tn = 0; tn < ntn; ++tn)
    for (int st = 0; st < nsteps; ++st)
      for (int kg = 0; kg < 2; ++kg)
        for (int n = 0; n < BN; ++n)
          for (int e = 0; e < 8; ++e) {
            uint16_t p[3];
            host_split(b[(size_t)(st * 16 + kg * 8 + e) * N + tn * BN + n], p);
            for (int q = 0; q < 3; ++q)
              img[((((size_t)(tn * nsteps + st) * 3 + q) * 2 + kg) * BN + n) * 8 + e] = p[q];
          }
  float *dA, *dC; unsigned char* dB;
  CHECK(hipMalloc(&dA, a.size() * 4)); CHECK(hipMalloc(&dC, (size_t)M * N * 4)); CHECK(hipMalloc(&dB, img.size() * 2));
  CHECK(hipMemcpy(dA, a.data(), a.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dB, img.data(), img.size() * 2, hipMemcpyHostToDevice));
  CHECK(hipMemset(dC, 0xff, (size_t)M * N * 4));
  dim3 grid((M / BM) * ntn), block(512);
  unsigned long long* dClk; CHECK(hipMalloc(&dClk, (size_t)grid.x * 16));
  auto launch = [&]() {
#define L(F) if (flags == F) { hipLaunchKernelGGL((split_gemm3_kernel<WM, WN, TN, F, ALEAD>), grid, block, 0, 0, dA, dB, dC, M, N, K, dClk); return; }
    L(0) L(1) L(2) L(4) L(6) L(8) L(15)
#undef L
    fprintf(stderr, "flags variant not built\n"); exit(2);
  };
  for (int i = 0; i < 3; ++i) launch();
  CHECK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  CHECK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) launch();
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
  std::vector<unsigned long long> hclk((size_t)grid.x * 2);
  CHECK(hipMemcpy(hclk.data(), dClk, hclk.size() * 8, hipMemcpyDeviceToHost));
  double cyc = 0, real = 0;
  for (unsigned i = 0; i < grid.x; ++i) { cyc += (double)hclk[2 * i]; real += (double)hclk[2 * i + 1]; }
  const double ghz_sustained = cyc / real * 0.1;       // last back-to-back launch
  // the same launch after an idle gap (what a profiler-serialised dispatch sees)
  float ms_gap = 0; double ghz_gap = 0;
  for (int i = 0; i < 3; ++i) {
    CHECK(hipDeviceSynchronize());
    hipEvent_t g0, g1; CHECK(hipEventCreate(&g0)); CHECK(hipEventCreate(&g1));
    for (volatile int spin = 0; spin < 20000000; ++spin) {}
    CHECK(hipEventRecord(g0)); launch(); CHECK(hipEventRecord(g1)); CHECK(hipEventSynchronize(g1));
    CHECK(hipEventElapsedTime(&ms_gap, g0, g1));
    CHECK(hipMemcpy(hclk.data(), dClk, hclk.size() * 8, hipMemcpyDeviceToHost));
    cyc = real = 0;
    for (unsigned k = 0; k < grid.x; ++k) { cyc += (double)hclk[2 * k]; real += (double)hclk[2 * k + 1]; }
    ghz_gap = cyc / real * 0.1;
  }
  std::vector<float> c((size_t)M * N);
  CHECK(hipMemcpy(c.data(), dC, c.size() * 4, hipMemcpyDeviceToHost));
  double worst = 0, worst32 = 0;
  for (int t = 0; t < 64; ++t) {
    const int row = t < 8 ? (t < 4 ? t * 37 % BM : M - 1 - t * 11) : (int)(((uint64_t)t * 2654435761ull) % (uint64_t)M);
    for (int n = 0; n < N; n += 5) {
      double ref = 0, mag = 0; float f = 0.f;
      for (int k = 0; k < K; ++k) {
        const double p = (double)a[(size_t)row * K + k] * (double)b[(size_t)k * N + n];
        ref += p; mag += fabs(p);
        f += a[(size_t)row * K + k] * b[(size_t)k * N + n];
      }
      worst = fmax(worst, fabs(c[(size_t)row * N + n] - ref) / mag);
      worst32 = fmax(worst32, fabs((double)f - ref) / mag);
    }
  }
  const double tf = 2.0 * M * N * K / (ms * 1e-3) / 1e12;
  printf("{\"kernel\": \"split_gemm3<%d,%d,%d,alead%d>\", \"M\": %d, \"N\": %d, \"K\": %d, \"flags\": %d, \"ms\": %.4f, \"effective_f32_TFLOPs\": %.1f, "
         "\"ghz_back_to_back\": %.3f, \"ms_after_idle\": %.4f, \"ghz_after_idle\": %.3f, \"bf16_mfma_TFLOPs\": %.1f, \"max_err_over_sum_abs\": %.3e, \"host_f32_sequential_err\": %.3e}\n",
         WM, WN, TN, ALEAD, M, N, K, flags, ms, tf, ghz_sustained, ms_gap, ghz_gap, tf * 6, worst, worst32);
  hipFree(dA); hipFree(dB); hipFree(dC);
  return 0;
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 65280, N = argc > 2 ? atoi(argv[2]) : 256, K = argc > 3 ? atoi(argv[3]) : 2304;
  const int flags = argc > 4 ? atoi(argv[4]) : 0;
  const int alead = argc > 5 ? atoi(argv[5]) : 1, reps = argc > 6 ? atoi(argv[6]) : 20;
  return alead == 2 ? run<4, 2, 4, 2>(M, N, K, flags, reps) : run<4, 2, 4, 1>(M, N, K, flags, reps);
}
