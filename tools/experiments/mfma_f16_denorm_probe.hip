// Does v_mfma_f32_32x32x16_f16 on gfx950 honour f16 subnormal inputs, and does the packed f32 -> f16 conversion round to
// nearest and produce subnormals?  (The fp16x2 split of the convolution kernels needs both: the lo piece of an operand
// that is small against the tensor maximum is an f16 subnormal.)
//   hipcc --offload-arch=gfx950 -O2 mfma_f16_denorm_probe.hip -o build/mfma_f16_denorm_probe && build/mfma_f16_denorm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));

__global__ void probe(const float* in, float* out, unsigned* cvt) {
  const int lane = threadIdx.x;
  // case c: A = in[2c], B = in[2c+1] in every k position; C[i][j] = 16 * A * B when nothing is flushed
  for (int c = 0; c < 4; ++c) {
    const f2 ab = {in[2 * c], in[2 * c + 1]};
    const h2 abh = __builtin_convertvector(ab, h2);
    h8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = abh[0]; b[e] = abh[1]; }
    f16v acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    if (lane == 0) out[c] = acc[0];
  }
  if (lane == 0) {
    for (int c = 0; c < 6; ++c) {
      const f2 v = {in[8 + c], 0.f};
      const h2 h = __builtin_convertvector(v, h2);
      cvt[c] = *reinterpret_cast<const unsigned short*>(&h);
    }
  }
}

int main() {
  float h_in[16] = {
      ldexpf(1.f, -20), 1024.f,            // A subnormal, B normal: expect 16 * 2^-10 = 2^-6
      1024.f, ldexpf(1.f, -20),            // B subnormal
      ldexpf(1.f, -20), ldexpf(1.f, -20),  // both subnormal: 16 * 2^-40 = 2^-36
      ldexpf(1.f, -14), 1.f,               // smallest normal: 2^-10
      ldexpf(1.f, -20),                    // -> 0x0010 when subnormals are produced
      1.f + ldexpf(1.f, -11) + ldexpf(1.f, -20),   // just above the tie: RN -> 0x3c01, RTZ -> 0x3c00
      1.f + ldexpf(1.f, -11),              // exact tie: RNE -> 0x3c00
      1.f + 3 * ldexpf(1.f, -11),          // exact tie: RNE -> 0x3c02
      ldexpf(1.f, -25) * 1.5f,             // below half the smallest subnormal step? 2^-24 is the step: 1.5 * 2^-25 -> RN 0x0001
      65519.f,                             // largest value that rounds to 65504 (0x7bff)
      0, 0};
  float *d_in, *d_out; unsigned* d_cvt;
  hipMalloc(&d_in, sizeof(h_in)); hipMalloc(&d_out, 16 * 4); hipMalloc(&d_cvt, 16 * 4);
  hipMemcpy(d_in, h_in, sizeof(h_in), hipMemcpyHostToDevice);
  probe<<<1, 64>>>(d_in, d_out, d_cvt);
  float o[4]; unsigned c[6];
  hipMemcpy(o, d_out, 16, hipMemcpyDeviceToHost); hipMemcpy(c, d_cvt, 24, hipMemcpyDeviceToHost);
  const char* names[4] = {"A subnormal x B normal", "A normal x B subnormal", "both subnormal", "A smallest normal"};
  const float expect[4] = {ldexpf(1.f, -6), ldexpf(1.f, -6), ldexpf(1.f, -36), ldexpf(1.f, -10)};
  for (int i = 0; i < 4; ++i) printf("mfma f16 %-24s got %.6e expect %.6e %s\n", names[i], o[i], expect[i], o[i] == expect[i] ? "OK" : "FLUSHED/DIFFERENT");
  const unsigned ce[6] = {0x0010, 0x3c01, 0x3c00, 0x3c02, 0x0001, 0x7bff};
  for (int i = 0; i < 6; ++i) printf("cvt f32->f16 case %d got 0x%04x expect 0x%04x %s\n", i, c[i], ce[i], c[i] == ce[i] ? "OK" : "DIFFERENT");
  return 0;
}
