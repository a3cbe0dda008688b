// TEST INFRASTRUCTURE ONLY -- a minimal HIP-on-CPU *simulator*.
//
// There is no GPU in the build container, so the kernel sources under
// object_detection_tracking_amd/csrc/ are additionally compiled with g++
// against THIS header (it shadows <hip/hip_runtime.h>) into
// tests/emu/libodt_emu.so, and the `-m "not gpu"` suite runs the very same
// kernel code (same indexing, same LDS tiling, same MFMA fragment maps, same
// shuffles) block by block on the host.  It is a simulator for tests, not a
// backend: nothing under object_detection_tracking_amd/ can load it (the
// product loader only opens libodt_hip.so and raises if it is missing), and
// no performance number is ever taken from it.
//
// Model: one workgroup at a time per host worker thread; every HIP thread is a
// fiber (own stack, hand-written context switch).  __syncthreads() and the wave-level primitives (shuffles,
// ballots, MFMA) are fiber switch points.  Wave = 64 lanes.  Wave-level ops
// must be reached by all 64 lanes of a wave in the same order (true of the
// real hardware too).  MFMA fragment maps follow the gfx950 layout:
//   v_mfma_f32_32x32x2_f32 : A[i=l&31][k=l>>5], B[k=l>>5][j=l&31],
//                            D reg r -> row (r&3)+8*(r>>2)+4*(l>>5), col l&31
//   v_mfma_f32_16x16x4_f32 : A[i=l&15][k=l>>4], B[k=l>>4][j=l&15],
//                            D reg r -> row 4*(l>>4)+r, col l&15
#pragma once

#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <thread>
#include <vector>

#define ODT_HIP_EMULATOR 1

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static thread_local

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3_emu { unsigned x, y, z; };

struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct int4 { int x, y, z, w; };
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
static inline float2 make_float2(float a, float b) { return {a, b}; }
static inline float4 make_float4(float a, float b, float c, float d) { return {a, b, c, d}; }
static inline int2 make_int2(int a, int b) { return {a, b}; }
static inline int4 make_int4(int a, int b, int c, int d) { return {a, b, c, d}; }
static inline uint2 make_uint2(unsigned a, unsigned b) { return {a, b}; }
static inline uint4 make_uint4(unsigned a, unsigned b, unsigned c, unsigned d) { return {a, b, c, d}; }

typedef int hipError_t;
typedef void* hipStream_t;
struct hipEvent_s { double t; };
typedef hipEvent_s* hipEvent_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorUnknown = 999 };
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost,
                     hipMemcpyDeviceToDevice, hipMemcpyDefault };
struct hipDeviceProp_t { char name[256]; int multiProcessorCount; char gcnArchName[256]; };

namespace hipemu {

constexpr int kWave = 64;
constexpr size_t kStack = 96 * 1024;

struct WaveX {           // per-wave exchange area (double buffered)
  uint64_t u[2][kWave];
  float a[2][kWave], b[2][kWave];
  unsigned short wa[2][kWave][8], wb[2][kWave][8];   // bf16x8 MFMA operands
  int src[2][kWave];
  unsigned long long seq[kWave];
};

struct Fiber {
  void* sp;              // saved stack pointer while the fiber is not running (hipemu_switch)
  uint3_emu tid;
  int linear;            // linear thread id in block
  int state;             // 0 runnable, 1 at barrier, 2 done
  unsigned long long wseq;  // number of wave ops issued
  char* stack;
};

struct Worker {
  void* sched_sp;        // the scheduler's saved stack pointer while a fiber runs
  std::vector<Fiber> fibers;
  std::vector<WaveX> waves;
  Fiber* cur = nullptr;
  dim3 block_idx, block_dim, grid_dim;
  int at_barrier = 0, live = 0;
  const std::function<void()>* body = nullptr;
};

extern thread_local Worker* tl_worker;

inline Worker& W() { return *tl_worker; }
}  // namespace hipemu
// minimal x86-64 context switch (callee-saved registers + stack pointer): glibc's swapcontext
// makes a sigprocmask system call per switch, which dominated the simulator's run time
extern "C" void hipemu_switch(void** save_sp, void* const* load_sp);
namespace hipemu {
inline void yield_to_sched() { Worker& w = W(); hipemu_switch(&w.cur->sp, &w.sched_sp); }

void launch(const std::function<void()>& body, dim3 grid, dim3 block);
double now_ms();

// ---- wave primitives ------------------------------------------------------
inline int lane_id() { return W().cur->linear & (kWave - 1); }
inline WaveX& wavex() { Worker& w = W(); return w.waves[w.cur->linear / kWave]; }

// deposit + sync; returns parity slot to read from
inline int wave_sync_begin() {
  Fiber* f = W().cur;
  return (int)(f->wseq & 1);
}
inline void wave_sync_end() {
  Worker& w = W();
  Fiber* f = w.cur;
  WaveX& x = w.waves[f->linear / kWave];
  x.seq[f->linear & (kWave - 1)] = ++f->wseq;
  yield_to_sched();
  // all lanes of this wave must have issued the same op
  int base = (f->linear / kWave) * kWave;
  int n = (int)w.fibers.size();
  for (int l = 0; l < kWave && base + l < n; ++l) {
    if (x.seq[l] < f->wseq) {
      fprintf(stderr, "hipemu: divergent wave-level op (block %u,%u lane %d)\n",
              w.block_idx.x, w.block_idx.y, f->linear);
      abort();
    }
  }
}

}  // namespace hipemu

#define threadIdx (hipemu::W().cur->tid)
#define blockIdx (hipemu::W().block_idx)
#define blockDim (hipemu::W().block_dim)
#define gridDim (hipemu::W().grid_dim)
static const int warpSize = 64;

inline void __syncthreads() {
  hipemu::Worker& w = hipemu::W();
  w.cur->state = 1;
  w.at_barrier++;
  hipemu::yield_to_sched();
}
inline void __threadfence() {}
inline void __threadfence_block() {}

// ---- shuffles / votes -------------------------------------------------------
template <typename T>
inline T __shfl(T v, int src, int width = 64) {
  static_assert(sizeof(T) <= 8, "shfl");
  int s = hipemu::wave_sync_begin();
  hipemu::WaveX& x = hipemu::wavex();
  int l = hipemu::lane_id();
  uint64_t raw = 0; memcpy(&raw, &v, sizeof(T));
  x.u[s][l] = raw;
  hipemu::wave_sync_end();
  int base = l & ~(width - 1);
  int sl = base + (src & (width - 1));
  T out; memcpy(&out, &x.u[s][sl], sizeof(T));
  return out;
}
template <typename T> inline T __shfl_xor(T v, int mask, int width = 64) {
  int l = hipemu::lane_id(); return __shfl(v, (l ^ mask), width); }
template <typename T> inline T __shfl_down(T v, unsigned d, int width = 64) {
  int l = hipemu::lane_id(); int t = (l & (width - 1)) + (int)d;
  T o = __shfl(v, t < width ? (l + (int)d) : l, 64); return o; }
template <typename T> inline T __shfl_up(T v, unsigned d, int width = 64) {
  int l = hipemu::lane_id(); int t = (l & (width - 1)) - (int)d;
  T o = __shfl(v, t >= 0 ? (l - (int)d) : l, 64); return o; }
inline unsigned long long __ballot(int pred) {
  int s = hipemu::wave_sync_begin();
  hipemu::WaveX& x = hipemu::wavex();
  x.u[s][hipemu::lane_id()] = pred ? 1u : 0u;
  hipemu::wave_sync_end();
  unsigned long long m = 0;
  for (int l = 0; l < 64; ++l) m |= (unsigned long long)(x.u[s][l] & 1u) << l;
  return m;
}
inline int __any(int p) { return __ballot(p) != 0ull; }
inline int __all(int p) { return __ballot(p) == ~0ull; }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __popc(unsigned v) { return __builtin_popcount(v); }
inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }
inline int __ffs(unsigned v) { return __builtin_ffs((int)v); }
inline int __clzll(unsigned long long v) { return v ? __builtin_clzll(v) : 64; }
inline int __clz(unsigned v) { return v ? __builtin_clz(v) : 32; }

// ---- MFMA -------------------------------------------------------------------
typedef float hipemu_f32x16 __attribute__((vector_size(64)));
typedef float hipemu_f32x4 __attribute__((vector_size(16)));

inline hipemu_f32x16 __builtin_amdgcn_mfma_f32_32x32x2f32(float a, float b, hipemu_f32x16 c,
                                                          int, int, int) {
  int s = hipemu::wave_sync_begin();
  hipemu::WaveX& x = hipemu::wavex();
  int l = hipemu::lane_id();
  x.a[s][l] = a; x.b[s][l] = b;
  hipemu::wave_sync_end();
  int col = l & 31;
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    float acc = c[r];
    for (int k = 0; k < 2; ++k)   // k-ordered fmaf chain (exact f32 semantics)
      acc = fmaf(x.a[s][row + 32 * k], x.b[s][col + 32 * k], acc);
    c[r] = acc;
  }
  return c;
}
inline hipemu_f32x4 __builtin_amdgcn_mfma_f32_16x16x4f32(float a, float b, hipemu_f32x4 c,
                                                         int, int, int) {
  int s = hipemu::wave_sync_begin();
  hipemu::WaveX& x = hipemu::wavex();
  int l = hipemu::lane_id();
  x.a[s][l] = a; x.b[s][l] = b;
  hipemu::wave_sync_end();
  int col = l & 15;
  for (int r = 0; r < 4; ++r) {
    int row = 4 * (l >> 4) + r;
    float acc = c[r];
    for (int k = 0; k < 4; ++k)
      acc = fmaf(x.a[s][row + 16 * k], x.b[s][col + 16 * k], acc);
    c[r] = acc;
  }
  return c;
}
// v_mfma_f32_32x32x16_bf16: lane l holds A[i=l&31][k=8*(l>>5)+e], B[k=8*(l>>5)+e][j=l&31], e=0..7;
// C/D as the f32 32x32 form.  bf16 x bf16 products are exact in f32; the sum over k is modelled
// in double and rounded once (the hardware's internal order is not specified).
typedef short hipemu_bf16x8 __attribute__((vector_size(16)));
inline float hipemu_bf16_to_f32(unsigned short h) {
  unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f;
}
inline hipemu_f32x16 __builtin_amdgcn_mfma_f32_32x32x16_bf16(hipemu_bf16x8 a, hipemu_bf16x8 b, hipemu_f32x16 c,
                                                             int, int, int) {
  int s = hipemu::wave_sync_begin();
  hipemu::WaveX& x = hipemu::wavex();
  int l = hipemu::lane_id();
  for (int e = 0; e < 8; ++e) { x.wa[s][l][e] = (unsigned short)a[e]; x.wb[s][l][e] = (unsigned short)b[e]; }
  hipemu::wave_sync_end();
  int col = l & 31;
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    double acc = c[r];
    for (int g = 0; g < 2; ++g)
      for (int e = 0; e < 8; ++e)
        acc += (double)hipemu_bf16_to_f32(x.wa[s][row + 32 * g][e]) * (double)hipemu_bf16_to_f32(x.wb[s][col + 32 * g][e]);
    c[r] = (float)acc;
  }
  return c;
}
// v_cvt_pk_bf16_f32 (round to nearest even); conv_split.hip takes this instead of the clang
// __bf16 vector conversion
inline unsigned hipemu_bf16_rne(float f) {
  unsigned u; memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;   // NaN stays NaN
  return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
#define ODT_CVT_PK_BF16(a0, a1) (hipemu_bf16_rne(a0) | (hipemu_bf16_rne(a1) << 16))
// v_mfma_f32_32x32x16_f16: the bf16 form's layout with IEEE half operands (subnormals at full precision, as gfx950
// does: tools/experiments/mfma_f16_denorm_probe.hip); f16 x f16 products are exact in f32.
typedef short hipemu_f16x8 __attribute__((vector_size(16)));
inline float hipemu_f16_to_f32(unsigned short h) {
  const unsigned s = (unsigned)(h & 0x8000u) << 16, e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
  float f;
  if (e == 0) { f = (float)m * 5.9604644775390625e-08f; return s ? -f : f; }      // m * 2^-24
  const unsigned u = e == 31 ? (s | 0x7f800000u | (m << 13)) : (s | ((e + 112u) << 23) | (m << 13));
  memcpy(&f, &u, 4);
  return f;
}
inline unsigned hipemu_f16_rne(float f) {      // v_cvt_f16_f32: round to nearest even, subnormals kept, >= 65520 -> inf
  unsigned u; memcpy(&u, &f, 4);
  const unsigned sign = (u >> 16) & 0x8000u;
  u &= 0x7fffffffu;
  if (u > 0x7f800000u) return sign | 0x7e00u;
  if (u >= 0x477ff000u) return sign | 0x7c00u;
  if (u < 0x38800000u) {                         // below 2^-14: a multiple of 2^-24
    float a; memcpy(&a, &u, 4);
    return sign | (unsigned)nearbyint((double)a * 16777216.0);
  }
  const unsigned v = u + 0xfffu + ((u >> 13) & 1u);
  return sign | ((v - 0x38000000u) >> 13);
}
inline hipemu_f32x16 __builtin_amdgcn_mfma_f32_32x32x16_f16(hipemu_f16x8 a, hipemu_f16x8 b, hipemu_f32x16 c,
                                                            int, int, int) {
  int s = hipemu::wave_sync_begin();
  hipemu::WaveX& x = hipemu::wavex();
  int l = hipemu::lane_id();
  for (int e = 0; e < 8; ++e) { x.wa[s][l][e] = (unsigned short)a[e]; x.wb[s][l][e] = (unsigned short)b[e]; }
  hipemu::wave_sync_end();
  int col = l & 31;
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    double acc = c[r];
    for (int g = 0; g < 2; ++g)
      for (int e = 0; e < 8; ++e)
        acc += (double)hipemu_f16_to_f32(x.wa[s][row + 32 * g][e]) * (double)hipemu_f16_to_f32(x.wb[s][col + 32 * g][e]);
    c[r] = (float)acc;
  }
  return c;
}
#define ODT_CVT_PK_F16(a0, a1) (hipemu_f16_rne(a0) | (hipemu_f16_rne(a1) << 16))
#define ODT_F16_LO_F32(u) hipemu_f16_to_f32((unsigned short)((u) & 0xffffu))
#define ODT_F16_HI_F32(u) hipemu_f16_to_f32((unsigned short)((u) >> 16))
// v_perm_b32: byte select from {s0 (bytes 4..7), s1 (bytes 0..3)}; selectors 0..7 only
inline unsigned __builtin_amdgcn_perm(unsigned s0, unsigned s1, unsigned sel) {
  const unsigned long long v = ((unsigned long long)s0 << 32) | s1;
  unsigned r = 0;
  for (int i = 0; i < 4; ++i) r |= (unsigned)((v >> (8 * ((sel >> (8 * i)) & 7))) & 0xff) << (8 * i);
  return r;
}
// ---- raw buffer loads (SRSRC descriptor; out-of-range dwords read as 0) ------------------------
struct __amdgpu_buffer_rsrc_t { const char* base; unsigned num_records; };
typedef unsigned int hipemu_u32x4 __attribute__((vector_size(16)));
inline __amdgpu_buffer_rsrc_t __builtin_amdgcn_make_buffer_rsrc(const void* p, short /*stride*/,
                                                                int num_records, int /*flags*/) {
  return __amdgpu_buffer_rsrc_t{(const char*)p, (unsigned)num_records};
}
inline hipemu_u32x4 __builtin_amdgcn_raw_buffer_load_b128(__amdgpu_buffer_rsrc_t r, int voffset,
                                                          int soffset, int /*aux*/) {
  hipemu_u32x4 v = {0u, 0u, 0u, 0u};
  const unsigned long long off = (unsigned long long)(unsigned)voffset + (unsigned)soffset;
  for (int d = 0; d < 4; ++d) {
    const unsigned long long o = off + 4ull * d;
    if (o + 4 <= r.num_records) { unsigned t; memcpy(&t, r.base + o, 4); v[d] = t; }
  }
  return v;
}
inline unsigned __builtin_amdgcn_raw_buffer_load_b32(__amdgpu_buffer_rsrc_t r, int voffset, int soffset,
                                                    int /*aux*/) {
  const unsigned long long o = (unsigned long long)(unsigned)voffset + (unsigned)soffset;
  unsigned t = 0;
  if (o + 4 <= r.num_records) memcpy(&t, r.base + o, 4);
  return t;
}
inline void __builtin_amdgcn_raw_buffer_store_b128(hipemu_u32x4 v, __amdgpu_buffer_rsrc_t r, int voffset,
                                                   int soffset, int /*aux*/) {
  const unsigned long long off = (unsigned long long)(unsigned)voffset + (unsigned)soffset;
  for (int d = 0; d < 4; ++d) {
    const unsigned long long o = off + 4ull * d;
    if (o + 4 <= r.num_records) { unsigned t = v[d]; memcpy(const_cast<char*>(r.base) + o, &t, 4); }
  }
}
// LDS-DMA (buffer_load ... lds): every lane's `size` bytes land at lds_base + lane * size (wave-uniform base; the per-lane
// part is the SOURCE offset); out-of-range lanes write zeros.  Synchronous here: the simulator checks addressing, not
// the vmcnt / barrier protocol.
inline void __builtin_amdgcn_raw_ptr_buffer_load_lds(__amdgpu_buffer_rsrc_t r, void* lds_base, int size, int voffset,
                                                     int soffset, int /*offset*/, int /*aux*/) {
  const int l = hipemu::lane_id();
  const unsigned long long off = (unsigned long long)(unsigned)voffset + (unsigned)soffset;
  for (int d = 0; d < size / 4; ++d) {
    const unsigned long long o = off + 4ull * d;
    unsigned t = 0;
    if (o + 4 <= r.num_records) memcpy(&t, r.base + o, 4);
    memcpy((char*)lds_base + (size_t)l * size + 4 * d, &t, 4);
  }
}
inline int __builtin_amdgcn_readfirstlane(int v) { return v; }    // callers pass wave-uniform values
inline void __builtin_amdgcn_s_barrier() { __syncthreads(); }
inline void __builtin_amdgcn_s_setprio(int) {}
inline unsigned __builtin_amdgcn_s_getreg(int) { return 0; }
inline void __builtin_amdgcn_sched_barrier(int) {}
inline void __builtin_amdgcn_s_sleep(int) {}
inline unsigned long long wall_clock64() { return (unsigned long long)(hipemu::now_ms() * 1e5); }

// ---- atomics / bit casts ----------------------------------------------------
template <typename T> inline T atomicAdd(T* p, T v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline float atomicAdd(float* p, float v) {
  float old, nw; uint32_t* ip = (uint32_t*)p; uint32_t o = __atomic_load_n(ip, __ATOMIC_RELAXED), n;
  do { memcpy(&old, &o, 4); nw = old + v; memcpy(&n, &nw, 4);
  } while (!__atomic_compare_exchange_n(ip, &o, n, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
  return old;
}
template <typename T> inline T atomicOr(T* p, T v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
template <typename T> inline T atomicMax(T* p, T v) {
  T o = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (o < v && !__atomic_compare_exchange_n(p, &o, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
  return o; }
template <typename T> inline T atomicMin(T* p, T v) {
  T o = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (o > v && !__atomic_compare_exchange_n(p, &o, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
  return o; }
template <typename T> inline T atomicExch(T* p, T v) { return __atomic_exchange_n(p, v, __ATOMIC_RELAXED); }
template <typename T> inline T atomicCAS(T* p, T expect, T desired) {
  __atomic_compare_exchange_n(p, &expect, desired, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED);
  return expect; }
inline double __longlong_as_double(long long v) { double d; memcpy(&d, &v, 8); return d; }
inline long long __double_as_longlong(double d) { long long v; memcpy(&v, &d, 8); return v; }
inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
inline int __float_as_int(float f) { int u; memcpy(&u, &f, 4); return u; }
inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
inline float __int_as_float(int u) { float f; memcpy(&f, &u, 4); return f; }
inline float __fmaf_rn(float a, float b, float c) { return fmaf(a, b, c); }
inline float __fdividef(float a, float b) { return a / b; }
inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
// hardware transcendental / packed-FMA builtins (v_exp_f32, v_rcp_f32, v_pk_fma_f32) the kernels use directly
inline float __builtin_amdgcn_exp2f(float x) { return exp2f(x); }
inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
template <typename V> inline V __builtin_elementwise_fma(V a, V b, V c) {
  V r;
  for (unsigned e = 0; e < sizeof(V) / sizeof(float); ++e) r[e] = fmaf(a[e], b[e], c[e]);
  return r;
}

// ---- runtime API --------------------------------------------------------------
inline const char* hipGetErrorString(hipError_t e) { return e == 0 ? "hipSuccess(emu)" : "hipError(emu)"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
  memset(p, 0, sizeof(*p)); strcpy(p->name, "hipemu-cpu-simulator"); strcpy(p->gcnArchName, "emu");
  p->multiProcessorCount = 256; return hipSuccess; }
inline hipError_t hipMalloc(void** p, size_t n) {
  *p = aligned_alloc(256, (n + 255) / 256 * 256 + 256); return *p ? hipSuccess : hipErrorOutOfMemory; }
template <typename T> inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc((void**)p, n); }
inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipHostMalloc(void** p, size_t n, unsigned = 0) { return hipMalloc(p, n); }
constexpr unsigned hipHostMallocMapped = 2;
inline hipError_t hipHostGetDevicePointer(void** d, void* h, unsigned = 0) { *d = h; return hipSuccess; }
inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t = 0) {
  memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t = 0) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipStreamCreate(hipStream_t* s) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
enum { hipStreamDefault = 0, hipStreamNonBlocking = 1, hipEventDisableTiming = 2 };
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { *s = nullptr; return hipSuccess; }
inline hipError_t hipDeviceGetStreamPriorityRange(int* least, int* greatest) { *least = 0; *greatest = -1; return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
// ---- graphs: not emulated; capture reports failure and the library falls back to direct launches
typedef struct hipGraph_s* hipGraph_t;
typedef struct hipGraphExec_s* hipGraphExec_t;
typedef void* hipGraphNode_t;
enum hipStreamCaptureMode { hipStreamCaptureModeGlobal = 0, hipStreamCaptureModeThreadLocal = 1 };
inline hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { return hipErrorUnknown; }
inline hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* g) { *g = nullptr; return hipErrorUnknown; }
inline hipError_t hipGraphInstantiate(hipGraphExec_t*, hipGraph_t, hipGraphNode_t*, char*, size_t) { return hipErrorUnknown; }
inline hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return hipErrorUnknown; }
inline hipError_t hipGraphDestroy(hipGraph_t) { return hipSuccess; }
inline hipError_t hipGraphExecDestroy(hipGraphExec_t) { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new hipEvent_s{0}; return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = new hipEvent_s{0}; return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = 0) { e->t = hipemu::now_ms(); return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->t - a->t); return hipSuccess; }

template <typename... KArgs, typename... Args>
inline void hipLaunchKernelGGL(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t /*shmem*/,
                               hipStream_t /*stream*/, Args... args) {
  std::function<void()> body = [=]() { kernel(static_cast<KArgs>(args)...); };
  hipemu::launch(body, grid, block);
}
