// Multilevel ROIAlign on gfx950 (gather / bilinear, L2-bound): FPN level assignment,
// tf.image.crop_and_resize(14x14, bilinear, extrapolation 0) with the reference's box
// transform, 2x2 average pool -> 7x7, plus the 7x7 mean ("DeepSORT appearance feature").
//
// Restates reference models.py:439-485 (fpn_map_rois_to_levels, multilevel_roi_align),
// nn.py:1229-1335 (crop_and_resize, roi_align[_multi]) and deep_sort/utils.py:27-28 (mean).
// Feature maps are NHWC, so each bilinear tap is one coalesced 256-byte row read; one thread
// per channel, one workgroup per (RoI, 64 channels), the NCHW output and the mean through an LDS
// tile.  fp32, operand order of TF-1.15 crop_and_resize_op.cc (see oracle/tfops.py); built with
// -ffp-contract=off.
#include <cstdint>

#include "odt_common.hpp"

namespace odt {
namespace {

__device__ __forceinline__ int fpn_level_of(float x0, float y0, float x1, float y1) {
  // models.py:441-447: floor(4 + log(sqrt(area) * (1/224) + 1e-6) * (1/ln 2))
  const float area = (y1 - y0) * (x1 - x0);
  const float sq = sqrtf(area);
  const float lv = floorf(4.0f + logf(sq * (float)(1. / 224) + 1e-6f) * (float)(1.0 / 0.6931471805599453));
  if (!(lv > 2.0f)) return 0;     // also NaN (negative area)
  if (lv >= 5.0f) return 3;
  return (int)lv - 2;
}

// One workgroup per (RoI, block of 64 channels): 16 channel quads x 16 output slots.  A thread owns four adjacent
// channels (every bilinear tap is ONE 16-byte load; 16 lanes cover the 256 contiguous bytes of a pixel's 64 channels --
// a quarter of the load instructions of one channel per lane, which is what bounds this gather) and takes every 16th of
// the OUT x OUT outputs.  Every output value also goes to an LDS tile [64][OUT*OUT (+1)], so that
//   * fpn_box_feat (NCHW: this RoI's [64][OUT][OUT] block is contiguous) leaves as whole coalesced rows instead of one
//     4-byte store per lane at a stride of 49 floats, and
//   * the 7x7 mean (deep_sort/utils.py:27-28) is summed from the tile in the order of np.mean over (h, w) of the row-major
//     window -- the order the former second pass (roi_pool_mean_kernel over the NCHW tensor) used: bit-identical `pooled`,
//     without re-reading the tensor this kernel has just written and without its launch.
constexpr int kRoiCB = 64;                              // channels per workgroup
template <int OUT>
__global__ void __launch_bounds__(256) roi_align_kernel(RoiAlignParams p, int B) {
  constexpr int OO = OUT * OUT;
  constexpr int LDP = (OO & 1) ? OO : OO + 1;           // odd row pitch: channels fall on distinct banks
  __shared__ float tile[kRoiCB * LDP];
  const int r = blockIdx.x, cb = blockIdx.y * kRoiCB;
  int b, out_row = r;
  if (p.box_ind != nullptr) {
    b = p.box_ind[r];
  } else {
    b = r / p.per_image;
    const int j = r - b * p.per_image;
    if (p.count != nullptr) {
      if (j >= p.count[b]) {
        if (p.amax != nullptr && p.out_nhwc != nullptr && !p.pack_rows) {      // (see RoiAlignParams::amax)
          const int nchz = p.C - cb < kRoiCB ? p.C - cb : kRoiCB;
          for (int i = threadIdx.x; i < OO * nchz; i += 256) p.out_nhwc[((size_t)r * OO + i / nchz) * p.C + cb + i % nchz] = 0.f;
        }
        return;
      }
      if (p.pack_rows) {           // outputs packed over the valid rows (reference [M,...]: fpn_box_feat, masks)
        int base = 0;
        for (int q = 0; q < b; ++q) base += p.count[q];
        out_row = base + j;
      }                            // else row r stays row r: the box head's consumers index b * per_image + j
    }
  }
  (void)B;
  const float* bx = p.boxes + (size_t)r * 4;
  const float X0 = bx[0], Y0 = bx[1], X1 = bx[2], Y1 = bx[3];
  const int L = p.levels != nullptr ? p.levels[r] - p.level0 : fpn_level_of(X0, Y0, X1, Y1);
  // select this level's parameters with compares (a runtime-indexed kernel-argument array would
  // be spilled to scratch / waterfall SGPR reads)
  float is = p.inv_stride[0];
  int H = p.h[0], W = p.w[0], aw = p.alloc_w[0], ah = p.alloc_h[0], ldc = p.ldc[0];
  const float* fbase = p.feat[0];
#pragma unroll
  for (int q = 1; q < 5; ++q) {
    if (L == q) {
      is = p.inv_stride[q]; H = p.h[q]; W = p.w[q]; aw = p.alloc_w[q]; ah = p.alloc_h[q];
      ldc = p.ldc[q]; fbase = p.feat[q];
    }
  }
  const float x0 = X0 * is, y0 = Y0 * is, x1 = X1 * is, y1 = Y1 * is;
  const float fH1 = (float)(H - 1), fW1 = (float)(W - 1);
  constexpr int CS = 2 * OUT;                           // 14 (box head, features) or 28 (mask head)
  // transform_fpcoor_for_tf (nn.py:1238-1271)
  const float sw = (x1 - x0) / (float)CS, sh = (y1 - y0) / (float)CS;
  const float nx0 = (x0 + sw / 2.0f - 0.5f) / fW1, ny0 = (y0 + sh / 2.0f - 0.5f) / fH1;
  const float nw = sw * (float)(CS - 1) / fW1, nh = sh * (float)(CS - 1) / fH1;
  const float by1 = ny0, bx1 = nx0, by2 = ny0 + nh, bx2 = nx0 + nw;
  // crop_and_resize_op.cc
  const float hs = (by2 - by1) * fH1 / (float)(CS - 1);
  const float ws = (bx2 - bx1) * fW1 / (float)(CS - 1);
  const float* feat = fbase + (size_t)b * ah * aw * ldc;
  const int cq = threadIdx.x & 15, c = cb + cq * 4;      // first of this thread's four channels
  const int nch = p.C - cb < kRoiCB ? p.C - cb : kRoiCB;        // channels of this block
  const bool stage = p.out_nchw != nullptr || p.pooled != nullptr;
  // whole quads only where the pixel row holds them (the row pitch ldc is a multiple of 4 and pad channels, if any, are
  // readable: feature maps are allocated with ldc >= C rounded up to 4); lanes past the block's channels idle
  const int nq = (nch + 3) >> 2;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  float vmax = 0.f;                                      // |max| of what goes to out_nhwc (p.amax)
  if (cq < nq) {
    for (int q = threadIdx.x >> 4; q < OO; q += 16) {
      const int oy = q / OUT, ox = q - oy * OUT;
      float yl[2];
      int top[2], bot[2];
      bool vy[2];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const float in_y = by1 * fH1 + (float)(2 * oy + k) * hs;
        vy[k] = !(in_y < 0.f || in_y > fH1);
        const float iy = vy[k] ? in_y : 0.f;
        top[k] = (int)floorf(iy);
        bot[k] = (int)ceilf(iy);
        yl[k] = iy - (float)top[k];
      }
      f32x4 v[2][2];
#pragma unroll
      for (int qx = 0; qx < 2; ++qx) {
        const float in_x = bx1 * fW1 + (float)(2 * ox + qx) * ws;
        const bool vx = !(in_x < 0.f || in_x > fW1);
        const float ix = vx ? in_x : 0.f;
        const int lef = (int)floorf(ix), rig = (int)ceilf(ix);
        const float xl = ix - (float)lef;
#pragma unroll
        for (int qy = 0; qy < 2; ++qy) {
          f32x4 val = zero;
          if (vx && vy[qy]) {
            const f32x4 tl = *reinterpret_cast<const f32x4*>(feat + ((size_t)top[qy] * aw + lef) * ldc + c);
            const f32x4 tr = *reinterpret_cast<const f32x4*>(feat + ((size_t)top[qy] * aw + rig) * ldc + c);
            const f32x4 bl = *reinterpret_cast<const f32x4*>(feat + ((size_t)bot[qy] * aw + lef) * ldc + c);
            const f32x4 br = *reinterpret_cast<const f32x4*>(feat + ((size_t)bot[qy] * aw + rig) * ldc + c);
            const f32x4 t = tl + (tr - tl) * xl;
            const f32x4 bm = bl + (br - bl) * xl;
            val = t + (bm - t) * yl[qy];
          }
          v[qy][qx] = val;
        }
      }
      // 2x2 average pool (nn.py:1332): ((v00 + v01) + v10) + v11, * 0.25
      const f32x4 o = (((v[0][0] + v[0][1]) + v[1][0]) + v[1][1]) * 0.25f;
      if (p.out_nhwc) {
        vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(o[0]), fabsf(o[1]))), fmaxf(fabsf(o[2]), fabsf(o[3])));     // (pad channels read as stored: finite)
        float* dst = p.out_nhwc + (((size_t)out_row * OUT + oy) * OUT + ox) * p.C + c;
        if (c + 4 <= p.C && (p.C & 3) == 0) {
          *reinterpret_cast<f32x4*>(dst) = o;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (c + e < p.C) dst[e] = o[e];
        }
      }
      if (stage) {
#pragma unroll
        for (int e = 0; e < 4; ++e) tile[(cq * 4 + e) * LDP + q] = o[e];
      }
    }
  }
  if (p.amax != nullptr) {
    // one conditional atomic per wave (f32 bit patterns of non-negative values order like unsigned integers)
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, d));
    if ((threadIdx.x & 63) == 0) {
      const unsigned bits = __float_as_uint(vmax);
      if (bits > __atomic_load_n(p.amax, __ATOMIC_RELAXED)) atomicMax(p.amax, bits);
    }
  }
  if (!stage) return;
  __syncthreads();
  if (p.out_nchw) {
    float* dst = p.out_nchw + ((size_t)out_row * p.C + cb) * OO;          // [nch][OUT][OUT], contiguous
    for (int i = threadIdx.x; i < nch * OO; i += 256) dst[i] = tile[(i / OO) * LDP + i % OO];
  }
  if (p.pooled && threadIdx.x < nch) {
    // np.mean(feat, axis=(1, 2)) of the row-major window: sequential sum, then / 49
    float s = 0.f;
    for (int q = 0; q < OO; ++q) s += tile[threadIdx.x * LDP + q];
    p.pooled[(size_t)out_row * p.C + cb + threadIdx.x] = s / (float)OO;
  }
}

}  // namespace

int launch_roi_align(const RoiAlignParams& p, hipStream_t stream) {
  ODT_CHECK(p.R_cap > 0, "roi_align: no rows");
  const int B = (p.box_ind == nullptr && p.per_image > 0) ? p.R_cap / p.per_image : 0;
  const int out = p.out_size == 0 ? kRoiOut : p.out_size;
  ODT_CHECK(out == kRoiOut || out == 2 * kRoiOut, "roi_align: output side must be 7 or 14");
  ODT_CHECK(out == kRoiOut || p.pooled == nullptr, "roi_align: pooled features are 7x7 only");
  for (int l = 0; l < 5; ++l)          // 16-byte taps: channel quads of a pixel row
    ODT_CHECK(p.feat[l] == nullptr || ((p.ldc[l] & 3) == 0 && ((uintptr_t)p.feat[l] & 15) == 0), "roi_align: feature rows must be 16-byte aligned (ldc % 4 == 0)");
  const dim3 grid(p.R_cap, (p.C + kRoiCB - 1) / kRoiCB);
  if (out == kRoiOut)
    hipLaunchKernelGGL(roi_align_kernel<kRoiOut>, grid, dim3(256), 0, stream, p, B);
  else
    hipLaunchKernelGGL(roi_align_kernel<2 * kRoiOut>, grid, dim3(256), 0, stream, p, B);
  ODT_HIP(hipGetLastError());
  return 0;
}

}  // namespace odt
