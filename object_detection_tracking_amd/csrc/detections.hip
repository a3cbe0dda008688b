// Detection tail on gfx950: box-head decode + clip + softmax, per-class NMS, final top-k.
//
//   ODT_GRAPH_SINGLE: reference models.py:828-843 (decode /[10,10,5,5], default clip
//                     log(1333/16), clip_boxes, softmax), nms_return_masks :1202-1223
//                     (prob > result_score_thres, NMS IoU .5, <= result_per_im),
//                     fastrcnn_predictions :1258-1304 (union, top result_per_im).
//   ODT_GRAPH_MULTI : fastrcnn_predictions_multibatch :2924-2976
//                     (combined_non_max_suppression, score_threshold = -inf, zero-padded
//                     slots of the other images are legal candidates).
// Canonical tie orders are those of oracle/tfops.py.  One 1024-thread workgroup per
// (class, image) for the NMS, one per image for the final selection.
#include "odt_common.hpp"
#include "select_device.hpp"

namespace odt {
namespace {

// ---- K11: per-RoI softmax + decode + clip ---------------------------------------------------
__global__ void __launch_bounds__(256) head_post_kernel(DetectParams p) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = row / p.K, j = row - b * p.K;
  if (b >= p.B || j >= p.nprops[b]) return;
  const float* ho = p.head_out + (size_t)row * p.ld;
  const int C = p.C;
  // tf.nn.softmax: exp(x - max) * (1 / sum)
  float mx = ho[0];
  for (int c = 1; c < C; ++c) mx = fmaxf(mx, ho[c]);
  float sum = 0.f;
  float* pr = p.probs + (size_t)row * C;
  for (int c = 0; c < C; ++c) {
    const float e = expf(ho[c] - mx);
    pr[c] = e;
    sum += e;
  }
  const float inv = 1.0f / sum;
  for (int c = 0; c < C; ++c) pr[c] = pr[c] * inv;
  // decode_bbox_target(box_logits / reg_weights, proposal) + clip_boxes
  const float* an = p.props + (size_t)row * 4;
  const float wa = an[2] - an[0], ha = an[3] - an[1];
  const float xa = (an[2] + an[0]) * 0.5f, ya = (an[3] + an[1]) * 0.5f;
  const float fw = (float)p.img_w, fh = (float)p.img_h;
  for (int c = 1; c < C; ++c) {
    const float* d = ho + C + c * 4;
    const float tx = d[0] / p.reg_w[0], ty = d[1] / p.reg_w[1];
    const float tw = d[2] / p.reg_w[2], th = d[3] / p.reg_w[3];
    const float wb = expf(fminf(tw, p.decode_clip)) * wa;
    const float hb = expf(fminf(th, p.decode_clip)) * ha;
    const float xb = tx * wa + xa, yb = ty * ha + ya;
    float* o = p.dec_boxes + ((size_t)row * (C - 1) + (c - 1)) * 4;
    o[0] = fminf(fmaxf(xb - wb * 0.5f, 0.f), fw);
    o[1] = fminf(fmaxf(yb - hb * 0.5f, 0.f), fh);
    o[2] = fminf(fmaxf(xb + wb * 0.5f, 0.f), fw);
    o[3] = fminf(fmaxf(yb + hb * 0.5f, 0.f), fh);
  }
}

// ---- K12/K13: per-(class, image) NMS ---------------------------------------------------------
__global__ void __launch_bounds__(kSelThreads) class_nms_kernel(DetectParams p) {
  __shared__ __attribute__((aligned(16))) char raw[sizeof(NmsScratch)];
  __shared__ int s_roi[kMaxTopK];
  __shared__ int s_wave[kSelWaves + 1];
  NmsScratch& s = *reinterpret_cast<NmsScratch*>(raw);
  unsigned long long* keys = s.mask;                              // alias: dead before the bitmask
  const int c = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int n = p.nprops[b];
  const int Cm1 = p.C - 1;
  float prob = 0.f;
  int cand = 0;
  if (tid < n) {
    prob = p.probs[((size_t)b * p.K + tid) * p.C + c + 1];
    cand = (p.graph == 0) ? (prob > p.score_thresh) : 1;
  }
  // candidates first, by (prob desc, roi asc); non-candidates get small unique keys
  keys[tid] = cand ? make_key64(prob, (unsigned)tid) : (unsigned long long)(kMaxTopK - tid);
  int ncand;
  block_scan_excl(cand, s_wave, &ncand);     // contains barriers -> keys[] visible afterwards
  {
    const unsigned long long my = keys[tid];
    int rank = 0;
    for (int j = 0; j < kSelThreads; ++j) rank += keys[j] > my ? 1 : 0;
    if (rank < ncand) s_roi[rank] = tid;
  }
  __syncthreads();
  if (tid < ncand) {
    const float* src = p.dec_boxes + (((size_t)b * p.K + s_roi[tid]) * Cm1 + c) * 4;
    s.box[tid * 4 + 0] = src[0]; s.box[tid * 4 + 1] = src[1];
    s.box[tid * 4 + 2] = src[2]; s.box[tid * 4 + 3] = src[3];
  }
  __syncthreads();
  block_nms(ncand, p.per_im, p.nms_thresh, s);
  const int nk = s.nkeep;
  int* out = p.cls_keep + ((size_t)b * Cm1 + c) * p.per_im;
  for (int i = tid; i < nk; i += blockDim.x) out[i] = s_roi[s.keep[i]];
  if (tid == 0) p.cls_count[b * Cm1 + c] = nk;
}

// K > 1024 (select_device.hpp: paneled walk): the same candidates in the same order, any number of RoIs up to kMaxTopKBig
struct ClassBoxAt {
  const float* dec;      // dec_boxes of this image, class c: row stride Cm1 * 4
  const int* roi;        // LDS: candidate position -> RoI
  int stride;
  __device__ __forceinline__ void operator()(int i, float* out) const {
    const float* src = dec + (size_t)roi[i] * stride;
    out[0] = src[0]; out[1] = src[1]; out[2] = src[2]; out[3] = src[3];
  }
};
__global__ void __launch_bounds__(kSelThreads) class_nms_big_kernel(DetectParams p) {
  __shared__ __attribute__((aligned(16))) char raw[sizeof(NmsBigScratch)];
  __shared__ int s_roi[kMaxTopKBig];
  __shared__ int s_ncand;
  NmsBigScratch& s = *reinterpret_cast<NmsBigScratch*>(raw);
  unsigned long long* keys = reinterpret_cast<unsigned long long*>(s.kbox);     // dead before the walk keeps its first box
  const int c = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int n = p.nprops[b];
  const int Cm1 = p.C - 1;
  if (tid == 0) s_ncand = 0;
  __syncthreads();
  for (int t = tid; t < n; t += blockDim.x) {
    const float prob = p.probs[((size_t)b * p.K + t) * p.C + c + 1];
    const int cand = (p.graph == 0) ? (prob > p.score_thresh) : 1;
    keys[t] = cand ? make_key64(prob, (unsigned)t) : 0ull;       // real keys are > 0 (inverted index in the low word)
    if (cand) atomicAdd(&s_ncand, 1);
  }
  __syncthreads();
  // candidates by (prob desc, roi asc)
  for (int t = tid; t < n; t += blockDim.x) {
    const unsigned long long my = keys[t];
    if (my == 0ull) continue;
    int rank = 0;
    for (int j = 0; j < n; ++j) rank += keys[j] > my ? 1 : 0;
    s_roi[rank] = t;
  }
  __syncthreads();
  const int ncand = s_ncand;
  block_nms_paneled(ncand, p.per_im, p.nms_thresh,
                    ClassBoxAt{p.dec_boxes + ((size_t)b * p.K * Cm1 + c) * 4, s_roi, Cm1 * 4}, s);
  const int nk = s.nkeep;
  int* out = p.cls_keep + ((size_t)b * Cm1 + c) * p.per_im;
  for (int i = tid; i < nk; i += blockDim.x) out[i] = s_roi[s.keep[i]];
  if (tid == 0) p.cls_count[b * Cm1 + c] = nk;
}

// ---- final selection: top result_per_im over the union (models.py:1288-1301 / :2959-2973) ----
constexpr int kMaxFinal = 4096;
__global__ void __launch_bounds__(kSelThreads) final_select_kernel(DetectParams p) {
  __shared__ float s_prob[kMaxFinal];
  __shared__ int s_code[kMaxFinal];        // class * per_im + order
  __shared__ int s_off[64];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int Cm1 = p.C - 1, per = p.per_im;
  if (tid == 0) {
    int run = 0;
    for (int c = 0; c < Cm1; ++c) { s_off[c] = run; run += p.cls_count[b * Cm1 + c]; }
    s_off[Cm1] = run;
  }
  __syncthreads();
  const int total = s_off[Cm1];
  for (int e = tid; e < Cm1 * per; e += blockDim.x) {
    const int c = e / per, i = e - c * per;
    if (i < p.cls_count[b * Cm1 + c]) {
      const int roi = p.cls_keep[((size_t)b * Cm1 + c) * per + i];
      s_prob[s_off[c] + i] = p.probs[((size_t)b * p.K + roi) * p.C + c + 1];
      s_code[s_off[c] + i] = e;
    }
  }
  __syncthreads();
  const int nreal = total < per ? total : per;
  // rank in the merged (prob desc, class asc, order asc) order: every class list is already
  // sorted that way, so the rank is a sum of binary-search counts over the C-1 lists.
  for (int e = tid; e < total; e += blockDim.x) {
    const float my = s_prob[e];
    const int mc = s_code[e];
    int rank = 0;
    for (int c2 = 0; c2 < Cm1; ++c2) {
      int lo = s_off[c2], hi = s_off[c2 + 1];
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        const float o = s_prob[mid];
        if (o > my || (o == my && s_code[mid] < mc)) lo = mid + 1; else hi = mid;
      }
      rank += lo - s_off[c2];
    }
    if (rank < per) {
      const int c = mc / per, i = mc - c * per;
      const int roi = p.cls_keep[((size_t)b * Cm1 + c) * per + i];
      const float* src = p.dec_boxes + (((size_t)b * p.K + roi) * Cm1 + c) * 4;
      float* ob = p.out_boxes + ((size_t)b * per + rank) * 4;
      ob[0] = src[0]; ob[1] = src[1]; ob[2] = src[2]; ob[3] = src[3];
      p.out_probs[(size_t)b * per + rank] = my;
      p.out_labels[(size_t)b * per + rank] = c + 1;
    }
  }
  // Padding.  SINGLE: rows >= nreal are zero and not counted.  MULTI: the zero-score slots of
  // the other images' RoIs are legal picks of combined_non_max_suppression (score_threshold
  // -inf): class c can add min(per - kept_c, slots) of them; they sort after every real
  // detection, by class.
  int slots = 0;
  if (p.graph == 1)
    for (int q = 0; q < p.B; ++q) slots += (q == b) ? 0 : p.nprops[q];
  int valid = nreal;
  for (int r = nreal + tid; r < per; r += blockDim.x) {
    float* ob = p.out_boxes + ((size_t)b * per + r) * 4;
    ob[0] = 0.f; ob[1] = 0.f; ob[2] = 0.f; ob[3] = 0.f;
    p.out_probs[(size_t)b * per + r] = 0.f;
    int label = 0;
    if (p.graph == 1) {
      int run = nreal;
      for (int c = 0; c < Cm1 && label == 0; ++c) {
        int extra = per - p.cls_count[b * Cm1 + c];
        if (extra > slots) extra = slots;
        if (r < run + extra) label = c + 1;
        run += extra;
      }
    }
    p.out_labels[(size_t)b * per + r] = label;
  }
  if (p.graph == 1) {
    int run = nreal;
    for (int c = 0; c < Cm1; ++c) {
      int extra = per - p.cls_count[b * Cm1 + c];
      if (extra > slots) extra = slots;
      run += extra;
    }
    valid = run < per ? run : per;
  }
  if (tid == 0) p.out_valid[b] = valid;
}

}  // namespace

// the selection half alone (per-class NMS + merged top result_per_im) over boxes / scores supplied by the caller:
// the kernels behind tf.image.combined_non_max_suppression / nms_return_masks + fastrcnn_predictions, for tests that
// feed them known-answer vectors (p.probs [B*K, C] with column 0 unused, p.dec_boxes [B*K, C-1, 4] filled)
int launch_class_nms(const DetectParams& p, hipStream_t stream) {
  ODT_CHECK(p.K >= 1 && p.K <= kMaxTopKBig, "class_nms: K must be in [1,4096]");
  ODT_CHECK(p.C >= 2 && p.C <= 64, "class_nms: 2..64 classes");
  ODT_CHECK((p.C - 1) * p.per_im <= kMaxFinal, "class_nms: (C-1)*result_per_im too large");
  if (p.K <= kMaxTopK)
    hipLaunchKernelGGL(class_nms_kernel, dim3(p.C - 1, p.B), dim3(kSelThreads), 0, stream, p);
  else
    hipLaunchKernelGGL(class_nms_big_kernel, dim3(p.C - 1, p.B), dim3(kSelThreads), 0, stream, p);
  hipLaunchKernelGGL(final_select_kernel, dim3(p.B), dim3(kSelThreads), 0, stream, p);
  ODT_HIP(hipGetLastError());
  return 0;
}

int launch_detections(const DetectParams& p, hipStream_t stream) {
  ODT_CHECK(p.K >= 1 && p.K <= kMaxTopKBig, "detections: K must be in [1,4096]");
  ODT_CHECK(p.C >= 2 && p.C <= 64, "detections: 2..64 classes");
  ODT_CHECK((p.C - 1) * p.per_im <= kMaxFinal, "detections: (C-1)*result_per_im too large");
  const int rows = p.B * p.K;
  hipLaunchKernelGGL(head_post_kernel, dim3((rows + 255) / 256), dim3(256), 0, stream, p);
  if (p.K <= kMaxTopK)
    hipLaunchKernelGGL(class_nms_kernel, dim3(p.C - 1, p.B), dim3(kSelThreads), 0, stream, p);
  else
    hipLaunchKernelGGL(class_nms_big_kernel, dim3(p.C - 1, p.B), dim3(kSelThreads), 0, stream, p);
  hipLaunchKernelGGL(final_select_kernel, dim3(p.B), dim3(kSelThreads), 0, stream, p);
  ODT_HIP(hipGetLastError());
  return 0;
}

namespace {
// final_masks = sigmoid(mask_logits[r, label - 1]) on the 28x28 grid (reference models.py:951-962;
// the 2x2-stride-2 transposed conv was computed as one 1x1 conv to (dy,dx,c) sub-pixel channels,
// so the pixel shuffle happens here: (y, x) <- cell (y/2, x/2), sub-pixel (y%2, x%2)).
__global__ void __launch_bounds__(256) mask_select_kernel(MaskSelectParams p) {
  const int r = blockIdx.x;
  const int b = r / p.per_image, j = r - b * p.per_image;
  float* out = p.masks + (size_t)r * 784;
  if (j >= p.valid[b]) {
    for (int i = threadIdx.x; i < 784; i += blockDim.x) out[i] = 0.f;
    return;
  }
  const int cls = p.labels[r] - 1;
  for (int i = threadIdx.x; i < 784; i += blockDim.x) {
    const int y = i / 28, x = i - y * 28;
    const size_t cell = ((size_t)r * 14 + (y >> 1)) * 14 + (x >> 1);
    const float v = p.logits[(cell * 4 + (y & 1) * 2 + (x & 1)) * p.ld + cls];
    out[i] = 1.0f / (1.0f + expf(-v));
  }
}
}  // namespace

int launch_mask_select(const MaskSelectParams& p, hipStream_t stream) {
  hipLaunchKernelGGL(mask_select_kernel, dim3(p.B * p.per_image), dim3(256), 0, stream, p);
  ODT_HIP(hipGetLastError());
  return 0;
}

}  // namespace odt
