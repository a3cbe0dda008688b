#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((vector_size(16)));
// each lane loads 16 B from src at its own offset (some out of range) straight into LDS; then dump LDS
__global__ void k(const float* src, int nbytes, const int* offs, float* out) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[4096];
  const int tid = threadIdx.x;
  for (int i = tid; i < 1024; i += 256) reinterpret_cast<float*>(lds)[i] = -7.0f;
  __syncthreads();
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + wave * 1024), 16, offs[tid], 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = tid; i < 1024; i += 256) out[i] = reinterpret_cast<float*>(lds)[i];
}
int main() {
  const int n = 4096;
  std::vector<float> h(n); for (int i = 0; i < n; ++i) h[i] = i + 1;
  std::vector<int> offs(256);
  for (int t = 0; t < 256; ++t) offs[t] = (t % 5 == 3) ? (int)0x80000000u : ((t * 37) % (n / 4)) * 16;
  float *d, *o; int* dof;
  hipMalloc(&d, n * 4); hipMalloc(&o, 4096); hipMalloc(&dof, 1024);
  hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(dof, offs.data(), 1024, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, d, n * 4, dof, o);
  std::vector<float> r(1024); hipMemcpy(r.data(), o, 4096, hipMemcpyDeviceToHost);
  int bad = 0, zeros = 0;
  for (int t = 0; t < 256; ++t) for (int e = 0; e < 4; ++e) {
    float want = (t % 5 == 3) ? 0.f : h[offs[t] / 4 + e];
    if (r[t * 4 + e] != want) { if (bad < 8) printf("lane %d e %d got %g want %g\n", t, e, r[t*4+e], want); ++bad; }
    if (t % 5 == 3 && r[t*4+e] == 0.f) ++zeros;
  }
  printf("glds_oob_test bad=%d oob_zeros=%d (of %d)\n", bad, zeros, 4 * 51);
  return 0;
}
