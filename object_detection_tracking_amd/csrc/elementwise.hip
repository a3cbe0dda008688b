// HBM-bound elementwise kernels of the detector front end (gfx950).
//   preprocess : reference models.py:340-355 ((x*(1/255) - mean_bgr)/std_bgr) fused with
//                the zero pad of nn.py:871-878 and an HWC3 -> HWC4 repack so conv0 can read
//                its 7x(8 taps x 4 ch) rows as contiguous 128-byte slices.
//   maxpool    : reference nn.py:890-896 (zero pad top/left 1, 3x3 stride-2 VALID max).
#include "odt_common.hpp"

namespace odt {
namespace {

// a wave's maximum into the tensor's range slot (f32 bit pattern of a non-negative value; see ConvParams::out_amax)
__device__ __forceinline__ void record_amax(unsigned* slot, float vmax) {
  if (slot == nullptr) return;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, d));
  if ((threadIdx.x & 63) == 0) {
    const unsigned b = __float_as_uint(vmax);
    unsigned* w = slot + (blockIdx.x & (unsigned)(kAmaxWays - 1));       // (kAmaxWays words per tensor: conv_split_common.hpp amax_read)
    if (b > __atomic_load_n(w, __ATOMIC_RELAXED)) atomicMax(w, b);
  }
}

template <typename T>
__global__ void __launch_bounds__(256) preprocess_kernel(const T* __restrict__ frames, int B, int H,
                                                         int W, int pad_t, int pad_l, int Hp, int Wp,
                                                         float* __restrict__ out, unsigned* __restrict__ amax) {
  const long total = (long)B * Hp * Wp;
  float vmax = 0.f;                      // |max| of what this thread writes (the fp16x2 conv kernels' range record)
  // BGR mean / std (models.py:343-351: RGB constants reversed)
  const float mean0 = 0.406f, mean1 = 0.456f, mean2 = 0.485f;
  const float std0 = 0.225f, std1 = 0.224f, std2 = 0.229f;
  const float inv255 = (float)(1.0 / 255);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % Wp);
    const long t = i / Wp;
    const int y = (int)(t % Hp);
    const int b = (int)(t / Hp);
    const int sy = y - pad_t, sx = x - pad_l;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if ((unsigned)sy < (unsigned)H && (unsigned)sx < (unsigned)W) {
      const T* src = frames + (((long)b * H + sy) * W + sx) * 3;
      v[0] = ((float)src[0] * inv255 - mean0) / std0;
      v[1] = ((float)src[1] * inv255 - mean1) / std1;
      v[2] = ((float)src[2] * inv255 - mean2) / std2;
    }
    *reinterpret_cast<f32x4*>(out + i * 4) = v;
    vmax = fmaxf(vmax, fmaxf(fabsf(v[0]), fmaxf(fabsf(v[1]), fabsf(v[2]))));
  }
  record_amax(amax, vmax);
}

// Same, with the reference's host-side frame resize (nn.py:1540-1546: cv2.resize(frame.astype
// (float32), (neww, newh), INTER_LINEAR)) moved onto the device: the source frame [Hs, Ws] is
// sampled at pixel centres (i + 0.5) * (src / dst) - 0.5 (double, like the host restatement in
// nn.py of this package), indices clamped to the image, weights (1 - f, f) in fp32, rows blended
// horizontally first -- the operand order of object_detection_tracking_amd.nn.resizeImage, so
// the result is bit-identical to resizing on the host and feeding the float32 image.
template <typename T>
__global__ void __launch_bounds__(256) preprocess_resize_kernel(const T* __restrict__ frames, int B, int Hs,
                                                                int Ws, int H, int W, int pad_t, int pad_l,
                                                                int Hp, int Wp, float* __restrict__ out,
                                                                unsigned* __restrict__ amax) {
  const long total = (long)B * Hp * Wp;
  float vmax = 0.f;
  const float mean0 = 0.406f, mean1 = 0.456f, mean2 = 0.485f;
  const float std0 = 0.225f, std1 = 0.224f, std2 = 0.229f;
  const float inv255 = (float)(1.0 / 255);
  const double ry = (double)Hs / (double)H, rx = (double)Ws / (double)W;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % Wp);
    const long t = i / Wp;
    const int y = (int)(t % Hp);
    const int b = (int)(t / Hp);
    const int dy = y - pad_t, dx = x - pad_l;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if ((unsigned)dy < (unsigned)H && (unsigned)dx < (unsigned)W) {
      // axis taps (nn._axis_taps)
      double fy = ((double)dy + 0.5) * ry - 0.5, fx = ((double)dx + 0.5) * rx - 0.5;
      long y0 = (long)floor(fy), x0 = (long)floor(fx);
      double wy1 = fy - (double)y0, wx1 = fx - (double)x0;
      if (y0 < 0) { wy1 = 0.0; y0 = 0; }
      if (x0 < 0) { wx1 = 0.0; x0 = 0; }
      if (y0 >= Hs - 1) { y0 = Hs - 1; wy1 = 0.0; }
      if (x0 >= Ws - 1) { x0 = Ws - 1; wx1 = 0.0; }
      const long y1 = y0 + 1 < Hs ? y0 + 1 : Hs - 1, x1 = x0 + 1 < Ws ? x0 + 1 : Ws - 1;
      const float wy = (float)wy1, wx = (float)wx1;
      const float omy = 1.0f - wy, omx = 1.0f - wx;
      const T* r0 = frames + ((long)b * Hs + y0) * Ws * 3;
      const T* r1 = frames + ((long)b * Hs + y1) * Ws * 3;
      float px[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float top = (float)r0[x0 * 3 + c] * omx + (float)r0[x1 * 3 + c] * wx;
        const float bot = (float)r1[x0 * 3 + c] * omx + (float)r1[x1 * 3 + c] * wx;
        px[c] = top * omy + bot * wy;
      }
      v[0] = (px[0] * inv255 - mean0) / std0;
      v[1] = (px[1] * inv255 - mean1) / std1;
      v[2] = (px[2] * inv255 - mean2) / std2;
    }
    *reinterpret_cast<f32x4*>(out + i * 4) = v;
    vmax = fmaxf(vmax, fmaxf(fabsf(v[0]), fmaxf(fabsf(v[1]), fabsf(v[2]))));
  }
  record_amax(amax, vmax);
}

// one thread per (pixel, 4-channel group); channels contiguous -> 16-byte accesses
__global__ void __launch_bounds__(256) maxpool3x3s2_kernel(const float* __restrict__ in, int B, int H,
                                                           int W, int C, float* __restrict__ out,
                                                           int Ho, int Wo) {
  const int c4n = C >> 2;
  const long total = (long)B * Ho * Wo * c4n;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % c4n);
    long t = i / c4n;
    const int xo = (int)(t % Wo);
    t /= Wo;
    const int yo = (int)(t % Ho);
    const int b = (int)(t / Ho);
    bool any_pad = false;
    f32x4 m = {-3.402823466e38f, -3.402823466e38f, -3.402823466e38f, -3.402823466e38f};
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int y = 2 * yo + dy - 1;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int x = 2 * xo + dx - 1;
        if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) {
          const f32x4 v =
              *reinterpret_cast<const f32x4*>(in + (((long)b * H + y) * W + x) * C + c4 * 4);
#pragma unroll
          for (int k = 0; k < 4; ++k) m[k] = fmaxf(m[k], v[k]);
        } else {
          any_pad = true;   // tf.pad zeros take part in the max
        }
      }
    }
    if (any_pad) {
#pragma unroll
      for (int k = 0; k < 4; ++k) m[k] = fmaxf(m[k], 0.f);
    }
    *reinterpret_cast<f32x4*>(out + i * 4) = m;
  }
}

// P6 = 1x1 max-pool stride 2 of P5 (nn.py:1011-1013) == [::2, ::2]
__global__ void __launch_bounds__(256) subsample2_kernel(const float* __restrict__ in, int B, int H,
                                                         int W, int C, float* __restrict__ out, int Ho,
                                                         int Wo) {
  const int c4n = C >> 2;
  const long total = (long)B * Ho * Wo * c4n;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % c4n);
    long t = i / c4n;
    const int xo = (int)(t % Wo);
    t /= Wo;
    const int yo = (int)(t % Ho);
    const int b = (int)(t / Ho);
    *reinterpret_cast<f32x4*>(out + i * 4) =
        *reinterpret_cast<const f32x4*>(in + (((long)b * H + 2 * yo) * W + 2 * xo) * C + c4 * 4);
  }
}

inline unsigned grid_for(long total) {
  long g = (total + 255) / 256;
  if (g > 256 * 8) g = 256 * 8;
  if (g < 1) g = 1;
  return (unsigned)g;
}

}  // namespace

int launch_preprocess_resize(const void* frames, int dtype, int B, int Hs, int Ws, int H, int W, int pad_t,
                             int pad_l, int Hp, int Wp, float* out, hipStream_t stream, unsigned* amax) {
  const long total = (long)B * Hp * Wp;
  ODT_CHECK(Hs > 0 && Ws > 0, "preprocess: empty source frame");
  if (dtype == 0) {
    hipLaunchKernelGGL(preprocess_resize_kernel<unsigned char>, dim3(grid_for(total)), dim3(256), 0, stream,
                       (const unsigned char*)frames, B, Hs, Ws, H, W, pad_t, pad_l, Hp, Wp, out, amax);
  } else if (dtype == 1) {
    hipLaunchKernelGGL(preprocess_resize_kernel<float>, dim3(grid_for(total)), dim3(256), 0, stream,
                       (const float*)frames, B, Hs, Ws, H, W, pad_t, pad_l, Hp, Wp, out, amax);
  } else {
    set_error("preprocess: dtype must be ODT_DTYPE_U8 or ODT_DTYPE_F32");
    return 1;
  }
  ODT_HIP(hipGetLastError());
  return 0;
}

int launch_preprocess(const void* frames, int dtype, int B, int H, int W, int pad_t, int pad_l,
                      int Hp, int Wp, float* out, hipStream_t stream, unsigned* amax) {
  const long total = (long)B * Hp * Wp;
  if (dtype == 0) {
    hipLaunchKernelGGL(preprocess_kernel<unsigned char>, dim3(grid_for(total)), dim3(256), 0, stream,
                       (const unsigned char*)frames, B, H, W, pad_t, pad_l, Hp, Wp, out, amax);
  } else if (dtype == 1) {
    hipLaunchKernelGGL(preprocess_kernel<float>, dim3(grid_for(total)), dim3(256), 0, stream,
                       (const float*)frames, B, H, W, pad_t, pad_l, Hp, Wp, out, amax);
  } else {
    set_error("preprocess: dtype must be ODT_DTYPE_U8 or ODT_DTYPE_F32");
    return 1;
  }
  ODT_HIP(hipGetLastError());
  return 0;
}

int launch_subsample2(const float* in, int B, int H, int W, int C, float* out, int Ho, int Wo,
                      hipStream_t stream) {
  ODT_CHECK(C % 4 == 0, "subsample2: C must be a multiple of 4");
  const long total = (long)B * Ho * Wo * (C / 4);
  hipLaunchKernelGGL(subsample2_kernel, dim3(grid_for(total)), dim3(256), 0, stream, in, B, H, W, C, out,
                     Ho, Wo);
  ODT_HIP(hipGetLastError());
  return 0;
}

int launch_maxpool3x3s2(const float* in, int B, int H, int W, int C, float* out, int Ho, int Wo,
                        hipStream_t stream) {
  ODT_CHECK(C % 4 == 0, "maxpool: C must be a multiple of 4");
  const long total = (long)B * Ho * Wo * (C / 4);
  hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3(grid_for(total)), dim3(256), 0, stream, in, B, H, W, C,
                     out, Ho, Wo);
  ODT_HIP(hipGetLastError());
  return 0;
}

}  // namespace odt
