// EfficientDet detection tail on gfx950.
//
// Restates reference efficientdet_wrapper.py:363-480 (add_metric_fn_inputs: concatenate the levels,
// tf.nn.top_k over ALL (anchor, class) logits with k = max_detection_topk, gather class / box /
// level), efficientdet/anchors.py:369-489 (decode_box_outputs_tf, sigmoid,
// tf.image.non_max_suppression_with_scores: hard NMS, IoU 0.5, score threshold, max_boxes_to_draw;
// boxes * image_scale as x1,y1,x2,y2; class + 1) -- the per-level ROIAlign mean of
// efficientdet_wrapper.py:244-361 runs on roi_align_kernel.
//
//   pack    : per level, class logits [h,w,ldc] -> contiguous sortable u32 keys [N * classes]
//   select  : exact global radix select of the k largest 64-bit keys (score key << 32 | ~index, i.e.
//             value desc, index asc like tf.nn.top_k): 8 passes of 8 bits, per-workgroup LDS
//             histograms merged with integer atomics (order independent => deterministic)
//   compact : keys >= the k-th key -> list (atomic append; the order is fixed by the sort below)
//   sort    : one workgroup, bitonic sort of the k keys in LDS (<= 8192), then decode each winner
//   nms     : one workgroup walks the sorted candidates; every kept box suppresses the rest in
//             parallel (<= max_boxes kept => <= 100 x k IoUs, no k x k mask)
#include "odt_common.hpp"
#include "select_device.hpp"

namespace odt {
namespace {

__global__ void __launch_bounds__(256) eff_pack_kernel(const float* __restrict__ cls, int npix, int ldc, int nch,
                                                       unsigned* __restrict__ keys) {
  const long total = (long)npix * nch;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long pix = i / nch;
    const int ch = (int)(i - pix * nch);
    keys[i] = sortable_key(cls[pix * ldc + ch]);
  }
}

// state[0] = prefix of the score key (high word), state[1] = prefix of ~index (low word),
// state[2] = k still to find among the elements matching the prefix, state[3] = compaction counter,
// state[4] = 1 once the threshold is final: after the four score passes, when exactly the elements still to find carry
// the threshold score (no tie to break by index -- the usual case), the four index passes have nothing to do and leave
// at once (the low word stays 0: every index qualifies)
__global__ void __launch_bounds__(256) eff_hist_kernel(const unsigned* __restrict__ keys, long n, int pass,
                                                       const unsigned* __restrict__ state,
                                                       unsigned* __restrict__ hist) {
  __shared__ unsigned h[256];
  if (pass >= 4 && state[4] != 0u) return;
  h[threadIdx.x] = 0;
  __syncthreads();
  const unsigned phi = state[0], plo = state[1];
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const unsigned k = keys[i];
    if (pass < 4) {
      const int sh = 24 - 8 * pass;
      if (pass == 0 || (k >> (sh + 8)) == (phi >> (sh + 8))) atomicAdd(&h[(k >> sh) & 255u], 1u);
    } else if (k == phi) {
      const unsigned inv = 0xFFFFFFFFu - (unsigned)i;
      const int sh = 24 - 8 * (pass - 4);
      if (pass == 4 || (inv >> (sh + 8)) == (plo >> (sh + 8))) atomicAdd(&h[(inv >> sh) & 255u], 1u);
    }
  }
  __syncthreads();
  if (h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}

__global__ void eff_init_kernel(unsigned* __restrict__ state, unsigned* __restrict__ hist, int k) {
  hist[threadIdx.x] = 0;
  if (threadIdx.x < 5) state[threadIdx.x] = threadIdx.x == 2 ? (unsigned)k : 0u;
}

__global__ void eff_scan_kernel(unsigned* __restrict__ hist, int pass, unsigned* __restrict__ state) {
  if (threadIdx.x != 0) return;
  if (pass >= 4 && state[4] != 0u) return;
  unsigned need = state[2], acc = 0;
  int d = 255;
  for (; d > 0; --d) {
    if (acc + hist[d] >= need) break;
    acc += hist[d];
  }
  need -= acc;
  const int sh = pass < 4 ? 24 - 8 * pass : 24 - 8 * (pass - 4);
  if (pass < 4) state[0] = (pass == 0 ? 0u : (state[0] & ~(0xFFFFFFFFu >> (8 * pass)))) | ((unsigned)d << sh);
  else state[1] = (pass == 4 ? 0u : (state[1] & ~(0xFFFFFFFFu >> (8 * (pass - 4))))) | ((unsigned)d << sh);
  state[2] = need;
  if (pass == 3 && hist[d] == need) state[4] = 1u;       // the whole threshold bucket is wanted: no index passes
  for (int i = 0; i < 256; ++i) hist[i] = 0;
}

__global__ void __launch_bounds__(256) eff_compact_kernel(const unsigned* __restrict__ keys, long n,
                                                          unsigned* __restrict__ state, int cap,
                                                          unsigned long long* __restrict__ out) {
  const unsigned long long T = ((unsigned long long)state[0] << 32) | state[1];
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const unsigned long long k = ((unsigned long long)keys[i] << 32) | (0xFFFFFFFFu - (unsigned)i);
    if (k >= T) {
      const unsigned pos = atomicAdd(&state[3], 1u);
      if ((int)pos < cap) out[pos] = k;
    }
  }
}

constexpr int kEffSortCap = 8192;

// one workgroup per image: sort the k selected keys (descending), decode every winner
__global__ void __launch_bounds__(1024) eff_sort_decode_kernel(EffPostParams p, int b) {
  __shared__ unsigned long long s[kEffSortCap];
  const int tid = threadIdx.x, nthr = blockDim.x, k = p.k;
  int npow = 1;
  while (npow < k) npow <<= 1;
  for (int i = tid; i < npow; i += nthr) s[i] = i < k ? p.sel[(size_t)b * p.k + i] : 0ull;
  __syncthreads();
  for (int size = 2; size <= npow; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int i = tid; i < npow; i += nthr) {
        const int j = i ^ stride;
        if (j > i) {
          const bool desc = (i & size) == 0;
          const unsigned long long a = s[i], c = s[j];
          if (desc ? a < c : a > c) { s[i] = c; s[j] = a; }
        }
      }
      __syncthreads();
    }
  for (int i = tid; i < k; i += nthr) {
    const unsigned long long key = s[i];
    const unsigned e = key64_index(key);
    const float logit = key_to_float((unsigned)(key >> 32));
    const unsigned a = e / (unsigned)p.ncls, cls = e - a * (unsigned)p.ncls;
    int l = 0;
#pragma unroll
    for (int q = 1; q < 5; ++q) if (a >= (unsigned)p.anchor_off[q]) l = q;
    const unsigned la = a - (unsigned)p.anchor_off[l];
    const unsigned pix = la / 9u, an = la - pix * 9u;
    const float* bx = p.box[l] + ((size_t)b * p.npix[l] + pix) * p.ldc_box + an * 4;
    const float* A = p.anchors + (size_t)a * 4;
    // decode_box_outputs_tf (anchors.py:369-396)
    const float yca = (A[0] + A[2]) / 2.f, xca = (A[1] + A[3]) / 2.f;
    const float ha = A[2] - A[0], wa = A[3] - A[1];
    const float w = expf(bx[3]) * wa, h = expf(bx[2]) * ha;
    const float yc = bx[0] * ha + yca, xc = bx[1] * wa + xca;
    float* o = p.cand_boxes + ((size_t)b * p.k + i) * 4;
    o[0] = yc - h / 2.f; o[1] = xc - w / 2.f; o[2] = yc + h / 2.f; o[3] = xc + w / 2.f;
    p.cand_scores[(size_t)b * p.k + i] = 1.0f / (1.0f + expf(-logit));
    p.cand_cls[(size_t)b * p.k + i] = (int)cls;
    p.cand_lvl[(size_t)b * p.k + i] = l + 3;
  }
}

// tf.image.non_max_suppression_with_scores (hard NMS) over the score-sorted candidates + outputs
__global__ void __launch_bounds__(1024) eff_nms_kernel(EffPostParams p) {
  __shared__ unsigned char removed[kEffSortCap];
  __shared__ int kept[1024];
  __shared__ int s_n;
  const int b = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x, k = p.k;
  const float* boxes = p.cand_boxes + (size_t)b * k * 4;
  const float* scores = p.cand_scores + (size_t)b * k;
  // candidates are sorted by score: the valid ones (score > threshold) are a prefix
  for (int i = tid; i < k; i += nthr) removed[i] = scores[i] > p.score_thresh ? 0 : 1;
  if (tid == 0) s_n = 0;
  __syncthreads();
  for (int i = 0; i < k; ++i) {
    if (removed[i]) continue;                 // uniform: LDS value read by every thread
    if (s_n >= p.max_out) break;
    float bi[4];
    bi[0] = fminf(boxes[i * 4], boxes[i * 4 + 2]); bi[1] = fminf(boxes[i * 4 + 1], boxes[i * 4 + 3]);
    bi[2] = fmaxf(boxes[i * 4], boxes[i * 4 + 2]); bi[3] = fmaxf(boxes[i * 4 + 1], boxes[i * 4 + 3]);
    const float ai = (bi[2] - bi[0]) * (bi[3] - bi[1]);
    __syncthreads();
    if (tid == 0) { kept[s_n] = i; s_n = s_n + 1; }
    for (int j = i + 1 + tid; j < k; j += nthr) {
      if (removed[j]) continue;
      float bj[4];
      bj[0] = fminf(boxes[j * 4], boxes[j * 4 + 2]); bj[1] = fminf(boxes[j * 4 + 1], boxes[j * 4 + 3]);
      bj[2] = fmaxf(boxes[j * 4], boxes[j * 4 + 2]); bj[3] = fmaxf(boxes[j * 4 + 1], boxes[j * 4 + 3]);
      const float aj = (bj[2] - bj[0]) * (bj[3] - bj[1]);
      if (iou_gt(bi, ai, bj, aj, p.iou_thresh)) removed[j] = 1;
    }
    __syncthreads();
  }
  __syncthreads();
  const int n = s_n;
  if (tid == 0) p.out_valid[b] = n;
  for (int r = tid; r < p.max_out; r += nthr) {
    float* ob = p.out_boxes + ((size_t)b * p.max_out + r) * 4;
    if (r < n) {
      const int i = kept[r];
      // boxes * image_scale, then [x1, y1, x2, y2] (anchors.py:462-468)
      ob[0] = boxes[i * 4 + 1] * p.image_scale; ob[1] = boxes[i * 4] * p.image_scale;
      ob[2] = boxes[i * 4 + 3] * p.image_scale; ob[3] = boxes[i * 4 + 2] * p.image_scale;
      p.out_scores[(size_t)b * p.max_out + r] = scores[i];
      p.out_labels[(size_t)b * p.max_out + r] = p.cand_cls[(size_t)b * k + i] + 1;
      p.out_levels[(size_t)b * p.max_out + r] = p.cand_lvl[(size_t)b * k + i];
    } else {
      ob[0] = ob[1] = ob[2] = ob[3] = 0.f;
      p.out_scores[(size_t)b * p.max_out + r] = 0.f;
      p.out_labels[(size_t)b * p.max_out + r] = 0;
      p.out_levels[(size_t)b * p.max_out + r] = 3;
    }
  }
}

inline unsigned grid_for(long total) {
  long g = (total + 255) / 256;
  if (g > 256 * 16) g = 256 * 16;
  return (unsigned)(g < 1 ? 1 : g);
}

}  // namespace

int launch_effdet_post(const EffPostParams& p, hipStream_t stream) {
  ODT_CHECK(p.k >= 1 && p.k <= kEffSortCap && p.max_out >= 1 && p.max_out <= 1024, "effdet: top-k must be <= 8192, detections <= 1024");
  const long ntot = (long)p.anchor_off[5] * p.ncls;
  ODT_CHECK(ntot < 0x7fffffffL && p.k <= ntot, "effdet: too many / too few logits for the 32-bit index keys");
  for (int b = 0; b < p.B; ++b) {
    unsigned* keys = p.keys;                       // reused per image (stream order)
    for (int l = 0; l < 5; ++l) {
      const long n = (long)p.npix[l] * 9 * p.ncls;
      hipLaunchKernelGGL(eff_pack_kernel, dim3(grid_for(n)), dim3(256), 0, stream,
                         p.cls[l] + (size_t)b * p.npix[l] * p.ldc_cls, p.npix[l], p.ldc_cls, 9 * p.ncls,
                         keys + (size_t)p.anchor_off[l] * p.ncls);
    }
    hipLaunchKernelGGL(eff_init_kernel, dim3(1), dim3(256), 0, stream, p.state, p.hist, p.k);
    for (int pass = 0; pass < 8; ++pass) {
      hipLaunchKernelGGL(eff_hist_kernel, dim3(grid_for(ntot / 4)), dim3(256), 0, stream, (const unsigned*)keys, ntot,
                         pass, (const unsigned*)p.state, p.hist);
      hipLaunchKernelGGL(eff_scan_kernel, dim3(1), dim3(64), 0, stream, p.hist, pass, p.state);
    }
    hipLaunchKernelGGL(eff_compact_kernel, dim3(grid_for(ntot / 4)), dim3(256), 0, stream, (const unsigned*)keys, ntot,
                       p.state, p.k, p.sel + (size_t)b * p.k);
    hipLaunchKernelGGL(eff_sort_decode_kernel, dim3(1), dim3(1024), 0, stream, p, b);
  }
  hipLaunchKernelGGL(eff_nms_kernel, dim3(p.B), dim3(1024), 0, stream, p);
  ODT_HIP(hipGetLastError());
  return 0;
}

}  // namespace odt
