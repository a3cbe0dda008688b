// EfficientNet building blocks on gfx950 (HBM-bound elementwise / stencil kernels; the 1x1 convs of
// the MBConv blocks run on conv_igemm_kernel).  NHWC fp32, channel counts padded to a multiple of
// 32 with zero channels (ldc) so that every tensor feeds the implicit-GEMM kernel directly.
//
// Restates reference efficientdet/backbone/efficientnet_model.py:162-330 (MBConvBlock: depthwise
// kxk 'same' conv + BN + swish, squeeze-excite = spatial mean -> 1x1 reduce + swish -> 1x1 expand ->
// sigmoid -> channel scale) and efficientdet_wrapper.py:45-60 + dataloader.normalize_image (BGR ->
// RGB, [0,1], ImageNet mean / std).  Built with -ffp-contract=off; summation orders are fixed
// (no atomics): bit-deterministic run to run.
#include <algorithm>

#include "odt_common.hpp"

namespace odt {
namespace {

inline unsigned grid_for(long total) {
  long g = (total + 255) / 256;
  if (g > 256 * 16) g = 256 * 16;
  if (g < 1) g = 1;
  return (unsigned)g;
}

__device__ __forceinline__ float swishf(float x) { return x * (1.0f / (1.0f + expf(-x))); }

// uint8 / float32 BGR frames -> normalised RGB, HWC4, zero padded (pad_t/pad_l = TF 'SAME' pads of
// the stride-2 stem; the 4th channel and the padding are zero)
template <typename T>
__global__ void __launch_bounds__(256) preprocess_rgb_kernel(const T* __restrict__ frames, int B, int H, int W,
                                                             int pad_t, int pad_l, int Hp, int Wp,
                                                             float* __restrict__ out) {
  const long total = (long)B * Hp * Wp;
  const float mean_r = 0.485f, mean_g = 0.456f, mean_b = 0.406f;
  const float std_r = 0.229f, std_g = 0.224f, std_b = 0.225f;
  const float inv255 = (float)(1.0 / 255);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % Wp);
    const long t = i / Wp;
    const int y = (int)(t % Hp), b = (int)(t / Hp);
    const int sy = y - pad_t, sx = x - pad_l;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if ((unsigned)sy < (unsigned)H && (unsigned)sx < (unsigned)W) {
      const T* src = frames + (((long)b * H + sy) * W + sx) * 3;       // B, G, R
      v[0] = ((float)src[2] * inv255 - mean_r) / std_r;
      v[1] = ((float)src[1] * inv255 - mean_g) / std_g;
      v[2] = ((float)src[0] * inv255 - mean_b) / std_b;
    }
    *reinterpret_cast<f32x4*>(out + i * 4) = v;
  }
}

// Same with the reference's input resize (dataloader.py:100-123 set_scale_factors_to_output_size +
// resize_and_crop_image on the NORMALISED image): the source [Hs,Ws] is scaled to [Hr,Wr] with
// TF-1.x tf.image.resize_images(BILINEAR) -- legacy coordinates in = out * (in_size / out_size),
// lower = floor, upper = min(lower + 1, size - 1), value = top + (bottom - top) * y_lerp with
// top = tl + (tr - tl) * x_lerp -- and zero padded to [H,W] at the bottom / right.
template <typename T>
__global__ void __launch_bounds__(256) preprocess_rgb_resize_kernel(const T* __restrict__ frames, int B, int Hs, int Ws,
                                                                    int Hr, int Wr, int pad_t, int pad_l, int Hp,
                                                                    int Wp, float* __restrict__ out) {
  const long total = (long)B * Hp * Wp;
  const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
  const float inv255 = (float)(1.0 / 255);
  const float sy = (float)Hs / (float)Hr, sx = (float)Ws / (float)Wr;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % Wp);
    const long t = i / Wp;
    const int y = (int)(t % Hp), b = (int)(t / Hp);
    const int dy = y - pad_t, dx = x - pad_l;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if ((unsigned)dy < (unsigned)Hr && (unsigned)dx < (unsigned)Wr) {
      const float fy = (float)dy * sy, fx = (float)dx * sx;
      const int y0 = (int)floorf(fy), x0 = (int)floorf(fx);
      const int y1 = y0 + 1 < Hs ? y0 + 1 : Hs - 1, x1 = x0 + 1 < Ws ? x0 + 1 : Ws - 1;
      const float ly = fy - (float)y0, lx = fx - (float)x0;
      const T* r0 = frames + ((long)b * Hs + y0) * Ws * 3;
      const T* r1 = frames + ((long)b * Hs + y1) * Ws * 3;
#pragma unroll
      for (int c = 0; c < 3; ++c) {          // output channel c = R,G,B <- source channel 2 - c
        const int sc = 2 - c;
        const float tl = ((float)r0[x0 * 3 + sc] * inv255 - mean[c]) / stdv[c];
        const float tr = ((float)r0[x1 * 3 + sc] * inv255 - mean[c]) / stdv[c];
        const float bl = ((float)r1[x0 * 3 + sc] * inv255 - mean[c]) / stdv[c];
        const float br = ((float)r1[x1 * 3 + sc] * inv255 - mean[c]) / stdv[c];
        const float top = tl + (tr - tl) * lx;
        const float bot = bl + (br - bl) * lx;
        v[c] = top + (bot - top) * ly;
      }
    }
    *reinterpret_cast<f32x4*>(out + i * 4) = v;
  }
}

// depthwise k x k conv (k = 3 or 5), TF 'SAME' padding, stride 1 or 2, folded BN, optional swish, optional fused
// squeeze (spatial mean of the output for the squeeze-excite gate).
// Workgroup = (block of 16 channel quads, pixel split, image); thread = one channel quad (16-byte accesses, channels
// innermost: 16 lanes read 256 contiguous bytes of a pixel) x one of 16 pixel groups, taking every 16th block of PX
// horizontally adjacent outputs of the split (one or two blocks per thread: the parallelism is in the grid).  The
// (PX-1) S + K input columns of a kernel row are loaded once and reused by all PX outputs.  Taps outside the image
// contribute nothing; every output sums ky-major, kx inner.  The squeeze is deterministic: per-thread sums in walk
// order, a fixed tree over the pixel groups -> sum_part[b][split][c]; channel_mean_fold_kernel adds the splits.
// (Folding in here by "the last workgroup to finish" was tried: the device-scope release / acquire fences it needs
// write back and invalidate the XCD's L2 per workgroup on this multi-die part -- 4x slower kernels.)
template <int K, int S, int PX>
__global__ void __launch_bounds__(256) dwconv_kernel(DwConvParams p) {
  constexpr int NC = (PX - 1) * S + K;
  __shared__ f32x4 red[256];
  const int tid = threadIdx.x;
  const int cq = tid & 15, pg = tid >> 4;
  // workgroup -> (channel block, pixel split, image).  Launch order goes round the 8 XCDs (each with its own L2), and
  // vertically adjacent splits share their halo rows: every XCD gets one contiguous band of the (image, split, channel
  // block) sequence, so that a halo row is fetched from HBM once per band instead of once per neighbour
  // (profiles/r02_pmc_summary_effdet_d7.json: the depthwise kernels fetched 1.6x their input before this).
  int cb = (int)blockIdx.x, sp = (int)blockIdx.y, b = (int)blockIdx.z;
  if (p.xcd_bands) {
    const unsigned nb = gridDim.x * gridDim.y, total = nb * gridDim.z;
    const unsigned lg = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    const unsigned xcd = lg & 7u, q = total >> 3, r = total & 7u;
    const unsigned nl = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (lg >> 3);
    b = (int)(nl / nb);
    const unsigned rem = nl - (unsigned)b * nb;
    sp = (int)(rem / gridDim.x); cb = (int)(rem - (unsigned)sp * gridDim.x);
  }
  // several maps in one launch (nlvl > 0, batch 1): the grid's image index selects the map
  const float* pin = p.in;
  float* pout = p.out;
  int pH = p.H, pW = p.W, pHo = p.Ho, pWo = p.Wo;
  if (p.nlvl > 0) {
    // (selected with constant indices: a runtime-indexed kernel-argument field would be copied to scratch memory)
#pragma unroll
    for (int i = 0; i < 5; ++i)
      if (i == b) { pin = p.lin[i]; pout = p.lout[i]; pH = pHo = p.lH[i]; pW = pWo = p.lW[i]; }
    b = 0;
  }
  const int c4 = cb * 16 + cq, c4n = p.ldc >> 2;
  const bool cok = c4 < c4n;
  const int nxb = (pWo + PX - 1) / PX, units = nxb * pHo;
  const int per = (units + p.nsplit - 1) / p.nsplit;
  const int lo = sp * per, hi = lo + per < units ? lo + per : units;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  f32x4 sum = zero;
  if (cok) {
    const f32x4 bias = *reinterpret_cast<const f32x4*>(p.bias + c4 * 4);
    for (int u = lo + pg; u < hi; u += 16) {
      const int yo = u / nxb, xb = u - yo * nxb;
      const int xo0 = xb * PX, x0 = xo0 * S - p.pad_l;
      f32x4 acc[PX];
#pragma unroll
      for (int q = 0; q < PX; ++q) acc[q] = zero;
#pragma unroll
      for (int ky = 0; ky < K; ++ky) {
        const int y = yo * S + ky - p.pad_t;
        if ((unsigned)y >= (unsigned)pH) continue;
        const float* row = pin + (((long)b * pH + y) * pW) * p.ldc + c4 * 4;
        f32x4 col[NC];
#pragma unroll
        for (int cidx = 0; cidx < NC; ++cidx) {
          const int x = x0 + cidx;
          col[cidx] = (unsigned)x < (unsigned)pW ? *reinterpret_cast<const f32x4*>(row + (long)x * p.ldc) : zero;
        }
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
          const f32x4 w = *reinterpret_cast<const f32x4*>(p.wt + (long)(ky * K + kx) * p.ldc + c4 * 4);
#pragma unroll
          for (int q = 0; q < PX; ++q) acc[q] += col[q * S + kx] * w;
        }
      }
#pragma unroll
      for (int q = 0; q < PX; ++q) {
        const int xo = xo0 + q;
        if (xo >= pWo) break;
        f32x4 v = acc[q] + bias;
        if (p.act == 2) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = swishf(v[e]);
        }
        *reinterpret_cast<f32x4*>(pout + (((long)b * pHo + yo) * pWo + xo) * p.ldc + c4 * 4) = v;
        sum += v;
      }
    }
  }
  if (p.sum_part == nullptr) return;
  red[tid] = sum;
  __syncthreads();
  if (pg == 0 && cok) {
    f32x4 t = red[cq];
#pragma unroll
    for (int g = 1; g < 16; ++g) t += red[g * 16 + cq];               // fixed order
    *reinterpret_cast<f32x4*>(p.sum_part + ((long)b * p.nsplit + sp) * p.ldc + c4 * 4) = t;
  }
}

// spatial mean per (image, channel), two deterministic stages: (1) every workgroup (image, 64-channel
// group, pixel split) sums its pixel range with 4 phases x 64 channels and a fixed tree over the
// phases; (2) the splits are added in index order and divided by HW.
__global__ void __launch_bounds__(256) channel_sum_kernel(const float* __restrict__ in, int HW, int ldc, int nsplit,
                                                          float* __restrict__ part_out) {
  // thread = (channel quad cq of the 64-channel group, pixel phase ph of 16): 16-byte loads
  __shared__ f32x4 part[16][16];
  const int b = blockIdx.y, cq = threadIdx.x & 15, ph = threadIdx.x >> 4, sp = blockIdx.z;
  const int c = blockIdx.x * 64 + cq * 4;
  const int per = (HW + nsplit - 1) / nsplit;
  const int lo = sp * per, hi = lo + per < HW ? lo + per : HW;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (c < ldc) {
    const float* src = in + (long)b * HW * ldc + c;
    for (int i = lo + ph; i < hi; i += 16) s += *reinterpret_cast<const f32x4*>(src + (long)i * ldc);
  }
  part[ph][cq] = s;
  __syncthreads();
  if (ph == 0 && c < ldc) {
    f32x4 t = part[0][cq];
#pragma unroll
    for (int q = 1; q < 16; ++q) t += part[q][cq];        // fixed order
    *reinterpret_cast<f32x4*>(part_out + ((long)b * nsplit + sp) * ldc + c) = t;
  }
}

__global__ void __launch_bounds__(256) channel_mean_final_kernel(const float* __restrict__ part, int B, int ldc,
                                                                 int nsplit, int HW, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * ldc) return;
  const int b = i / ldc, c = i - b * ldc;
  float s = 0.f;
  for (int sp = 0; sp < nsplit; ++sp) s += part[((long)b * nsplit + sp) * ldc + c];
  out[i] = s / (float)HW;
}

// Squeeze-excite gate in two small launches per block: mean (sum of the pixel-split partials / HW) -> 1x1
// reduce + bias + swish -> 1x1 expand + bias + sigmoid.  One workgroup per image; fixed summation
// orders (splits in index order, 64-lane strided partial sums + a fixed shuffle tree, reduced
// channels in index order).
constexpr int kSeMaxC = 4096, kSeMaxR = 256;     // EfficientNet-B7: 3840 expanded channels, 160 reduced
// stage A, grid (ceil(se / 4), B): one reduced channel per wave: r[j] = swish(b1[j] + <mean, w1[j]>)
// fold: mean[b][c] = (sum of the pixel-split partials, 4 phases in fixed order) / HW; grid (ldc/64, B)
__global__ void __launch_bounds__(256) channel_mean_fold_kernel(SeGateParams p) {
  __shared__ float ph4[4][64];
  const int b = blockIdx.y, cl = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  float s = 0.f;
  if (c < p.ldc) {
    const float* src = p.part + (long)b * p.nsplit * p.ldc + c;
#pragma unroll 8
    for (int sp = q; sp < p.nsplit; sp += 4) s += src[(long)sp * p.ldc];
  }
  ph4[q][cl] = s;
  __syncthreads();
  if (q == 0 && c < p.ldc)
    p.mean[(long)b * p.ldc + c] = ((ph4[0][cl] + ph4[1][cl]) + (ph4[2][cl] + ph4[3][cl])) / (float)p.HW;
}

__global__ void __launch_bounds__(256) se_reduce_kernel(SeGateParams p) {
  __shared__ float mean[kSeMaxC];
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int c = tid; c < p.ldc; c += blockDim.x) mean[c] = p.mean[(long)b * p.ldc + c];
  __syncthreads();
  const int j = blockIdx.x * 4 + wave;
  if (j < p.se) {
    const float* w = p.w1 + (long)j * p.ldc;
    float s = 0.f;
#pragma unroll 4
    for (int c = lane; c < p.ldc; c += 64) s += mean[c] * w[c];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (lane == 0) { const float v = s + p.b1[j]; p.r[(long)b * kSeMaxR + j] = v * (1.0f / (1.0f + expf(-v))); }
  }
}

// stage B, grid (ceil(mid / 256), B): gate[c] = sigmoid(b2[c] + sum_j r[j] * w2t[j][c]), j in index order
__global__ void __launch_bounds__(256) se_expand_kernel(SeGateParams p) {
  __shared__ float r[kSeMaxR];
  const int b = blockIdx.y, tid = threadIdx.x;
  for (int j = tid; j < p.se; j += blockDim.x) r[j] = p.r[(long)b * kSeMaxR + j];
  __syncthreads();
  const int c = blockIdx.x * 256 + tid;
  if (c >= p.mid) return;
  float s = p.b2[c];
#pragma unroll 8
  for (int j = 0; j < p.se; ++j) s += r[j] * p.w2t[(long)j * p.ldc + c];
  p.gate[(long)b * p.ldc + c] = 1.0f / (1.0f + expf(-s));
}

// (Round 5 measured the three stages as ONE launch -- a single 1024-thread workgroup per image running fold -> reduce ->
// expand through LDS, bit-identical gate: 37 us per block on average against 3 x 6.4 us for the three launches, D7 68.4 ->
// 64.5 FPS same box, profiles/r05_effdet_se_one_launch_ab.txt: for the wide late blocks one workgroup is too little
// parallelism for 2 x 500 K MACs.  Not kept.)
// x[b, :, :, c] *= s[b, c]   (squeeze-excite gate, already passed through the sigmoid)
__global__ void __launch_bounds__(256) channel_scale_kernel(float* __restrict__ x, const float* __restrict__ s,
                                                            int B, int HW, int ldc) {
  const int c4n = ldc >> 2;
  const long total = (long)B * HW * c4n;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % c4n);
    const int b = (int)(i / ((long)HW * c4n));
    f32x4* q = reinterpret_cast<f32x4*>(x + i * 4);
    const f32x4 g = *reinterpret_cast<const f32x4*>(s + (long)b * ldc + c4 * 4);
    *q = *q * g;
  }
}

// BiFPN node input combination (reference efficientdet_arch.py:105-200 resample_feature_map +
// :615-650 build_bifpn_layer): out = act( sum_i  w_i * resample_i(in_i) ), up to three inputs, each
// either at the node's resolution, nearest-neighbour upsampled (TF1 rule: src = min(floor(dst *
// in/out), in-1)) or 3x3 / stride-2 / 'SAME' max-pooled on the fly.  'fastattn' weights follow the
// graph's operand order ((x * w) / (sum_w + 1e-4)); 'sum' adds left to right.
// the node's fused value for channel quad c4 at pixel (y, x) of image b
__device__ __forceinline__ f32x4 bifpn_fuse_at(const FuseParams& p, int b, int y, int x, int c4) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    if (k >= p.n) break;
    const float* src = p.in[k] + (long)b * p.ih[k] * p.iw[k] * p.ldc + c4 * 4;
    f32x4 v;
    if (p.mode[k] == 0) {
      v = *reinterpret_cast<const f32x4*>(src + ((long)y * p.iw[k] + x) * p.ldc);
    } else if (p.mode[k] == 1) {
      int sy = (int)floorf((float)y * p.sy[k]), sx = (int)floorf((float)x * p.sx[k]);
      sy = sy < p.ih[k] - 1 ? sy : p.ih[k] - 1; sx = sx < p.iw[k] - 1 ? sx : p.iw[k] - 1;
      v = *reinterpret_cast<const f32x4*>(src + ((long)sy * p.iw[k] + sx) * p.ldc);
    } else {
      const float ninf = -3.402823466e38f;
      v = f32x4{ninf, ninf, ninf, ninf};
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        const int yy = 2 * y + dy - p.pt[k];
        if ((unsigned)yy >= (unsigned)p.ih[k]) continue;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const int xx = 2 * x + dx - p.pl[k];
          if ((unsigned)xx >= (unsigned)p.iw[k]) continue;
          const f32x4 q = *reinterpret_cast<const f32x4*>(src + ((long)yy * p.iw[k] + xx) * p.ldc);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], q[e]);
        }
      }
    }
    if (p.weighted) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = v[e] * p.wgt[k] / p.denom;
    }
    if (k == 0) acc = v; else acc += v;
  }
  if (p.act == 2) {
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] = swishf(acc[e]);
  }
  return acc;
}

__global__ void __launch_bounds__(256) bifpn_fuse_kernel(FuseParams p) {
  const int c4n = p.ldc >> 2;
  const long total = (long)p.B * p.h * p.w * c4n;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % c4n);
    long t = i / c4n;
    const int x = (int)(t % p.w); t /= p.w;
    const int y = (int)(t % p.h);
    const int b = (int)(t / p.h);
    *reinterpret_cast<f32x4*>(p.out + (((long)b * p.h + y) * p.w + x) * p.ldc + c4 * 4) = bifpn_fuse_at(p, b, y, x, c4);
  }
}


}  // namespace

int launch_preprocess_rgb(const void* frames, int dtype, int B, int H, int W, int pad_t, int pad_l, int Hp, int Wp,
                          float* out, hipStream_t stream) {
  const long total = (long)B * Hp * Wp;
  if (dtype == 0)
    hipLaunchKernelGGL(preprocess_rgb_kernel<unsigned char>, dim3(grid_for(total)), dim3(256), 0, stream,
                       (const unsigned char*)frames, B, H, W, pad_t, pad_l, Hp, Wp, out);
  else if (dtype == 1)
    hipLaunchKernelGGL(preprocess_rgb_kernel<float>, dim3(grid_for(total)), dim3(256), 0, stream,
                       (const float*)frames, B, H, W, pad_t, pad_l, Hp, Wp, out);
  else { set_error("preprocess: dtype must be ODT_DTYPE_U8 or ODT_DTYPE_F32"); return 1; }
  ODT_HIP(hipGetLastError());
  return 0;
}

int launch_bifpn_fuse(const FuseParams& p, hipStream_t stream) {
  ODT_CHECK(p.n >= 1 && p.n <= 3 && p.ldc % 4 == 0, "bifpn_fuse: bad arguments");
  const long total = (long)p.B * p.h * p.w * (p.ldc >> 2);
  hipLaunchKernelGGL(bifpn_fuse_kernel, dim3(grid_for(total)), dim3(256), 0, stream, p);
  ODT_HIP(hipGetLastError());
  return 0;
}


int launch_preprocess_rgb_resize(const void* frames, int dtype, int B, int Hs, int Ws, int Hr, int Wr, int pad_t,
                                 int pad_l, int Hp, int Wp, float* out, hipStream_t stream) {
  const long total = (long)B * Hp * Wp;
  if (dtype == 0)
    hipLaunchKernelGGL(preprocess_rgb_resize_kernel<unsigned char>, dim3(grid_for(total)), dim3(256), 0, stream,
                       (const unsigned char*)frames, B, Hs, Ws, Hr, Wr, pad_t, pad_l, Hp, Wp, out);
  else if (dtype == 1)
    hipLaunchKernelGGL(preprocess_rgb_resize_kernel<float>, dim3(grid_for(total)), dim3(256), 0, stream,
                       (const float*)frames, B, Hs, Ws, Hr, Wr, pad_t, pad_l, Hp, Wp, out);
  else { set_error("preprocess: dtype must be ODT_DTYPE_U8 or ODT_DTYPE_F32"); return 1; }
  ODT_HIP(hipGetLastError());
  return 0;
}

namespace {
constexpr int kDwPX2 = 2;
// outputs per thread along x at stride 1: 8 (k = 5: 10.6 loads per output against 16.3 at 4; D7 same-box A/B 66.7 -> 67.6
// FPS); ODT_DW_PX=4 is the A/B knob (knobs.hpp)
int dw_px1() {
  return env_knob_long(K_DW_PX, 8) == 4 ? 4 : 8;
}
}  // namespace

static int dwconv_cblocks(const DwConvParams& p) { return ((p.ldc >> 2) + 15) / 16; }

// pixel splits: one or two output blocks per thread (16 pixel groups per workgroup), at most 4096 workgroups in all
int dwconv_splits(const DwConvParams& p) {
  const int px = p.stride == 1 ? dw_px1() : kDwPX2;
  const long units = (long)((p.Wo + px - 1) / px) * p.Ho;
  // (with the fused squeeze every split is a partial sum the fold kernel has to add up: at most 1024 of them)
  const long sumcap = env_knob_long(K_DW_SUMCAP, 2048L);   // A/B knob (1024 ... 8192: +-0.5 %)
  const long cap = std::max<long>(1, (p.sum_part != nullptr ? sumcap : 4096) / ((long)dwconv_cblocks(p) * p.B));
  return (int)std::max<long>(1, std::min(std::min<long>(cap, 1024), (units + 15) / 16));
}

int launch_dwconv(const DwConvParams& p0, hipStream_t stream) {
  ODT_CHECK(p0.ldc % 4 == 0 && (p0.k == 3 || p0.k == 5) && (p0.stride == 1 || p0.stride == 2), "dwconv: bad geometry");
  DwConvParams p = p0;
  if (p.nlvl > 0) {
    // several maps in one launch: splits sized for the largest map (the others' surplus workgroups find nothing to do)
    ODT_CHECK(p.nlvl <= 5 && p.B == 1 && p.stride == 1 && p.sum_part == nullptr, "dwconv: multi-map launches are batch 1, stride 1");
    int big = 0;
    for (int i = 1; i < p.nlvl; ++i) if ((long)p.lH[i] * p.lW[i] > (long)p.lH[big] * p.lW[big]) big = i;
    p.H = p.Ho = p.lH[big]; p.W = p.Wo = p.lW[big];
  }
  p.cqn = 16; p.nsplit = dwconv_splits(p);
  const bool bands = !env_knob_off(K_DW_XCD);     // A/B knob
  p.xcd_bands = bands ? 1 : 0;
  const dim3 g(dwconv_cblocks(p), p.nsplit, p.nlvl > 0 ? p.nlvl : p.B), t(256);
  const bool wide = dw_px1() == 8;
  if (p.k == 3 && p.stride == 1) {
    if (wide) hipLaunchKernelGGL((dwconv_kernel<3, 1, 8>), g, t, 0, stream, p);
    else hipLaunchKernelGGL((dwconv_kernel<3, 1, 4>), g, t, 0, stream, p);
  } else if (p.k == 3) {
    hipLaunchKernelGGL((dwconv_kernel<3, 2, kDwPX2>), g, t, 0, stream, p);
  } else if (p.stride == 1) {
    if (wide) hipLaunchKernelGGL((dwconv_kernel<5, 1, 8>), g, t, 0, stream, p);
    else hipLaunchKernelGGL((dwconv_kernel<5, 1, 4>), g, t, 0, stream, p);
  } else {
    hipLaunchKernelGGL((dwconv_kernel<5, 2, kDwPX2>), g, t, 0, stream, p);
  }
  ODT_HIP(hipGetLastError());
  return 0;
}

// pixel splits so that (channel groups x images x splits) is about 512 workgroups, >= 64 pixels each
int channel_mean_splits(int HW, int ldc, int B) {
  const int groups = std::max(1, (ldc + 63) / 64 * B);
  int n = (512 + groups - 1) / groups;
  n = std::min(n, std::max(1, HW / 64));
  return std::max(1, std::min(n, 512));
}

// scratch: [B, channel_mean_splits(HW), ldc] floats
int launch_channel_mean(const float* in, int B, int HW, int ldc, float* scratch, float* out, hipStream_t stream) {
  const int ns = channel_mean_splits(HW, ldc, B);
  hipLaunchKernelGGL(channel_sum_kernel, dim3((ldc + 63) / 64, B, ns), dim3(256), 0, stream, in, HW, ldc, ns, scratch);
  hipLaunchKernelGGL(channel_mean_final_kernel, dim3((B * ldc + 255) / 256), dim3(256), 0, stream,
                     (const float*)scratch, B, ldc, ns, HW, out);
  ODT_HIP(hipGetLastError());
  return 0;
}

// channel_sum_kernel partials (scratch) + the gate, replacing channel_mean_final + two tiny GEMMs
int launch_se_gate(const float* in, const SeGateParams& p0, int B, float* scratch, hipStream_t stream) {
  SeGateParams p = p0;
  p.nsplit = channel_mean_splits(p.HW, p.ldc, B); p.part = scratch;
  hipLaunchKernelGGL(channel_sum_kernel, dim3((p.ldc + 63) / 64, B, p.nsplit), dim3(256), 0, stream, in, p.HW, p.ldc,
                     p.nsplit, scratch);
  ODT_CHECK(p.ldc <= kSeMaxC && p.se <= kSeMaxR, "se_gate: channel count too large for the LDS staging");
  hipLaunchKernelGGL(channel_mean_fold_kernel, dim3((p.ldc + 63) / 64, B), dim3(256), 0, stream, p);
  hipLaunchKernelGGL(se_reduce_kernel, dim3((p.se + 3) / 4, B), dim3(256), 0, stream, p);
  hipLaunchKernelGGL(se_expand_kernel, dim3((p.mid + 255) / 256, B), dim3(256), 0, stream, p);
  ODT_HIP(hipGetLastError());
  return 0;
}

int launch_se_gate_from_parts(const SeGateParams& p, int B, hipStream_t stream) {
  ODT_CHECK(p.ldc <= kSeMaxC && p.se <= kSeMaxR, "se_gate: channel count too large for the LDS staging");
  ODT_CHECK(p.part != nullptr && p.nsplit >= 1, "se_gate: partial sums missing");
  hipLaunchKernelGGL(channel_mean_fold_kernel, dim3((p.ldc + 63) / 64, B), dim3(256), 0, stream, p);
  hipLaunchKernelGGL(se_reduce_kernel, dim3((p.se + 3) / 4, B), dim3(256), 0, stream, p);
  hipLaunchKernelGGL(se_expand_kernel, dim3((p.mid + 255) / 256, B), dim3(256), 0, stream, p);
  ODT_HIP(hipGetLastError());
  return 0;
}

int launch_channel_scale(float* x, const float* s, int B, int HW, int ldc, hipStream_t stream) {
  const long total = (long)B * HW * (ldc >> 2);
  hipLaunchKernelGGL(channel_scale_kernel, dim3(grid_for(total)), dim3(256), 0, stream, x, s, B, HW, ldc);
  ODT_HIP(hipGetLastError());
  return 0;
}

}  // namespace odt
