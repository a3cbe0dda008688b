// Shared pieces of the split convolution kernels (conv_split1.hip: one-stage 4-wave bf16x3 loop; conv_split3.hip: 8-wave
// LDS-DMA bf16x3 kernels; conv_h2.hip / conv_h2k.hip: the fp16x2 kernels; conv_split.hip: weight images, policy, dispatch).
//
// bf16x3 arithmetic.  Every f32 operand is cut into three bf16 pieces by round-to-nearest,
//     x = hi + mid + lo   exactly   (3 x 8 significand bits = the 24 bits of an f32),
// |mid| <= 2^-8 |x|, |lo| <= 2^-16 |x|, so a*b is the sum of nine piece products.  The six largest
// are evaluated on v_mfma_f32_32x32x16_bf16 (hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid); the
// three dropped ones (mid*lo, lo*mid, lo*lo) are bounded by (2^-23 + 2^-32) |a||b|: two f32
// roundings of the product, unbiased (the pieces carry either sign).  Each piece product is exact
// in f32 (8 x 8 bits) and the accumulation is f32 inside the MFMA unit: the result carries the
// error of an f32 dot product with a different summation order (measured ~1e-7 of sum|a||b| on
// K = 2304, the same as a sequential f32 loop; tools/experiments/split_gemm.hip; the bound is
// asserted in tests/test_ops.py).  |x| above 3.39e38 (bf16 rounds to inf) is outside the domain;
// below ~1e-33 the lo piece is a bf16 subnormal (absolute effect < 1e-38 per product).
// bf16 MFMA runs at 16x the f32 MFMA rate, so six products cost 6/16 of the f32 instruction time:
// the ceiling is 2.67x the f32 MFMA peak.
#pragma once
#include <cstdlib>
#include <type_traits>

#include "odt_common.hpp"

namespace odt {
namespace {

typedef unsigned int u32x4 __attribute__((vector_size(16)));
typedef unsigned int u32x2 __attribute__((vector_size(8)));
typedef short bf16x8 __attribute__((vector_size(16)));

constexpr unsigned kOOB = 0x80000000u;   // buffer offset that is out of range for every tensor (< 2 GiB)

// Cout rounded up to the 64-wide n-tile granule: layers whose channel count is not a multiple of 64 (EfficientNet's
// 240, 432, 864 ...) run with zero weight rows and a zero bias in the padding; their tensors' pixel stride covers it
__host__ __device__ __forceinline__ int cout_padded(int cout) { return (cout + 63) & ~63; }
__device__ __forceinline__ int sfast_div(int n, unsigned mul, unsigned sh) {
  return mul ? (int)(__umulhi((unsigned)n, mul) >> sh) : n;
}
// Two f32 -> two bf16 (round to nearest even) in one dword: v_cvt_pk_bf16_f32.  (The CPU simulator
// of the test suite supplies its own ODT_CVT_PK_BF16.)
#ifndef ODT_CVT_PK_BF16
typedef __bf16 odt_bf16x2 __attribute__((ext_vector_type(2)));
typedef float odt_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt_pk_bf16(float a0, float a1) {
  const odt_f32x2 v = {a0, a1};
  const odt_bf16x2 r = __builtin_convertvector(v, odt_bf16x2);
  return *reinterpret_cast<const unsigned*>(&r);
}
#define ODT_CVT_PK_BF16(a0, a1) cvt_pk_bf16(a0, a1)
#endif
// x = hi + mid + lo exactly: hi = RN8(x); x - hi has <= 16 significant bits and is exact in f32;
// mid = RN8(x - hi); the rest has <= 8 bits, so lo is exact.  |mid| <= 2^-8 |x|, |lo| <= 2^-16 |x|.
__device__ __forceinline__ void split2(float a0, float a1, unsigned& hi, unsigned& mid, unsigned& lo) {
  hi = ODT_CVT_PK_BF16(a0, a1);
  const float r0 = a0 - __uint_as_float(hi << 16);
  const float r1 = a1 - __uint_as_float(hi & 0xffff0000u);
  mid = ODT_CVT_PK_BF16(r0, r1);
  const float s0 = r0 - __uint_as_float(mid << 16);
  const float s1 = r1 - __uint_as_float(mid & 0xffff0000u);
  lo = ODT_CVT_PK_BF16(s0, s1);
}

// ---- fp16x2 pieces (conv_h2.hip) ------------------------------------------------------------------------------------
// x 2^s = hi + lo + e with hi = RN11(x 2^s), lo = RN11(x 2^s - hi): |lo| <= 2^-11 |x 2^s|, |e| <= 2^-22 |x 2^s| (two f16
// significands of 11 bits and a sign), provided hi is finite (|x| 2^s < 65520) and lo is not lost to the f16 exponent
// range.  The power of two comes from the tensor's recorded |max| (h2_scale_exp): max |x| 2^s lies in [2^14, 2^15), so
// hi never overflows, lo is a NORMAL f16 for every element down to 2^-17 of the tensor maximum and a subnormal below
// (absolute error <= 2^-25 in scaled units = 2^-39 of the maximum -- the gfx950 matrix pipe takes f16 subnormals at full
// precision, tools/experiments/mfma_f16_denorm_probe.hip).  a*b is evaluated as hi*hi + hi*lo + lo*hi on
// v_mfma_f32_32x32x16_f16 (each piece product is exact in f32: 11 x 11 bits), f32 accumulation; the dropped lo*lo and the
// two representation errors are each <= 2^-22 |a||b|.
#ifdef ODT_HIP_EMULATOR
typedef short f16x8 __attribute__((vector_size(16)));
#else
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#endif
#ifndef ODT_CVT_PK_F16
typedef _Float16 odt_f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt_pk_f16(float a0, float a1) {      // v_cvt_pk_f16_f32 (round to nearest even)
  const odt_f32x2 v = {a0, a1};
  const odt_f16x2 r = __builtin_convertvector(v, odt_f16x2);
  return *reinterpret_cast<const unsigned*>(&r);
}
__device__ __forceinline__ float f16_lo_f32(unsigned u) { return (float)(*reinterpret_cast<const odt_f16x2*>(&u))[0]; }
__device__ __forceinline__ float f16_hi_f32(unsigned u) { return (float)(*reinterpret_cast<const odt_f16x2*>(&u))[1]; }
#define ODT_CVT_PK_F16(a0, a1) cvt_pk_f16(a0, a1)
#define ODT_F16_LO_F32(u) f16_lo_f32(u)
#define ODT_F16_HI_F32(u) f16_hi_f32(u)
#endif
__device__ __forceinline__ void split2h(float a0, float a1, float s, unsigned& hi, unsigned& lo) {
  const float x0 = a0 * s, x1 = a1 * s;
  hi = ODT_CVT_PK_F16(x0, x1);
  const float r0 = x0 - ODT_F16_LO_F32(hi), r1 = x1 - ODT_F16_HI_F32(hi);
  lo = ODT_CVT_PK_F16(r0, r1);
}
// 2^e as a float (e in [-126, 127])
__host__ __device__ __forceinline__ float pow2f(int e) {
  const unsigned u = (unsigned)(e + 127) << 23;
  float f;
  __builtin_memcpy(&f, &u, 4);
  return f;
}
// the power of two that takes a tensor whose |max| has the f32 bit pattern `amax_bits` into [2^14, 2^15): 14 - exponent,
// kept inside the exponents a float scale (and its inverse) can carry; 0 for an all-zero / non-finite maximum
__host__ __device__ __forceinline__ int h2_scale_exp(unsigned amax_bits) {
  const int be = (int)((amax_bits >> 23) & 0xffu);
  if (be == 0 || be == 255) return 0;
  const int s = 14 - (be - 127);
  return s > 100 ? 100 : (s < -100 ? -100 : s);
}
// A tensor's range slot is kAmaxWays words (odt_common.hpp; 1 in the product): a producer workgroup folds its |max| into word
// (workgroup index mod kAmaxWays), a consumer takes the maximum of all of them.  Same value whatever the number of ways.
__device__ __forceinline__ unsigned amax_read(const unsigned* slot) {
  unsigned a = 0u;
#pragma unroll
  for (int w = 0; w < kAmaxWays; ++w) { const unsigned b = slot[w]; a = b > a ? b : a; }
  return a;
}
__device__ __forceinline__ unsigned* amax_way(unsigned* slot) { return slot + (blockIdx.x & (unsigned)(kAmaxWays - 1)); }
// the A-side scale exponent of a conv: from the recorded |max| of its source tensor(s)
__device__ __forceinline__ int h2_in_scale_exp(const ConvParams& p) {
  unsigned a = p.in_amax != nullptr ? amax_read(p.in_amax) : 0u;
  if (p.in2_amax != nullptr) { const unsigned b = amax_read(p.in2_amax); a = b > a ? b : a; }
  return h2_scale_exp(a);
}

// counted waits / LDS-only barrier / LDS pointer type of the LDS-DMA kernels (the simulator runs the DMA synchronously)
#ifdef ODT_HIP_EMULATOR
#define ODT_LANE_ID() (hipemu::lane_id())
#define ODT_PIN2(a, b) do { } while (0)
#define ODT_WAIT_VM_LGKM0(n) do { } while (0)
#define ODT_BARRIER_LDS() __syncthreads()
#define ODT_LDS_PTR(p) ((void*)(p))
#else
// the lane id from the execution mask (v_mbcnt): recomputed where needed instead of held in a register
#define ODT_LANE_ID() ((int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)))
// an opaque use + redefinition of two values: everything they depend on is computed before this point, nothing that uses
// them moves above it (keeps IR-level code sinking from stretching live ranges across a register-tight region)
#define ODT_PIN2(a, b) asm volatile("" : "+v"(a), "+v"(b))
// counted wait: at most n vector-memory operations (A fetches / DMA of younger stages) stay in flight; all LDS done
#define ODT_WAIT_VM_LGKM0(n) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(n) : "memory")
// workgroup barrier that orders LDS traffic only (__syncthreads() would also drain the global stores in flight)
#define ODT_BARRIER_LDS() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); } while (0)
#define ODT_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#endif

#define ODT_STAMP(i) do { if constexpr (TRACE) { if (tid == 0) p.trace[(size_t)blockIdx.x * 16 + (i)] = wall_clock64(); } } while (0)

}  // namespace

// per-family launchers (conv_split.hip dispatches on ConvParams::wt_split_kind)
int launch_conv_split1(const ConvParams& p, const ConvParams* dev, hipStream_t stream);
int launch_conv_split3(const ConvParams& p, const ConvParams* dev, hipStream_t stream);
int launch_conv_h2(const ConvParams& p, const ConvParams* dev, hipStream_t stream);
void launch_conv_h2k(const ConvParams& p, const ConvParams* dev, unsigned grid, hipStream_t stream);   // conv_h2k.hip
bool conv_h2d_fits(const ConvParams& p);                                                               // conv_h2d.hip: double-stage two-wave tiles
void launch_conv_h2d(const ConvParams& p, const ConvParams* dev, unsigned grid, hipStream_t stream);
void launch_split_reduce(const ConvParams& p, const ConvParams* dev, hipStream_t stream);   // split-K combine (conv_split3.hip)

}  // namespace odt
