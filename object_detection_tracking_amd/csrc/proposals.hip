// RPN proposal generation on gfx950: per-level top-k over the objectness logits, anchor decode
// + clip of the selected anchors only, greedy NMS, cross-level merge.
//
//   ODT_GRAPH_SINGLE: generate_fpn_proposals / generate_rpn_proposals / decode_bbox_target
//                     (reference models.py:402-436, nn.py:1353-1400, nn.py:1518-1538)
//   ODT_GRAPH_MULTI : generate_rpn_proposals_multibatch + the zero-padded merge and the
//                     area > 0 filter (reference nn.py:1406-1482, models.py:2458-2522)
//
// One 1024-thread workgroup per (image, level) for select + decode, one for NMS, one per image
// for the merge; everything stays in LDS, counts stay on the device (no host sync).  K <= 1024 (every
// configuration the reference ships; the script default is 1000) takes the one-candidate-per-thread
// kernels; 1024 < K <= 4096 the same kernels in rounds of 1024 with the paneled NMS walk.
#include "odt_common.hpp"
#include "select_device.hpp"

namespace odt {
namespace {

// ---- stage A1: per-chunk top-k.  A level's logits are split into chunks of kSelChunk so that
// the big levels (P2: 388 800 logits at 1080p) spread over many CUs; the union of the chunk
// winners contains the level's top-k.
struct ChunkMap { int level, first, n; };
__device__ __forceinline__ ChunkMap chunk_of(const ProposalParams& p, int chunk) {
  ChunkMap m{0, 0, 0};
  int c = chunk;
  for (int l = 0; l < p.nlevels; ++l) {
    const int n = p.lvl[l].h * p.lvl[l].w * 3;
    const int nc = (n + kSelChunk - 1) / kSelChunk;
    if (c < nc) {
      m.level = l; m.first = c * kSelChunk;
      m.n = n - m.first < kSelChunk ? n - m.first : kSelChunk;
      return m;
    }
    c -= nc;
  }
  return m;
}
struct RpnChunkKeyAt {
  const float* base;   // rpn of this image
  int first;
  __device__ __forceinline__ unsigned long long operator()(int e) const {
    const int g = first + e;
    const int pix = g / 3, a = g - pix * 3;
    return make_key64(base[(size_t)pix * kRpnCh + a], (unsigned)g);
  }
};
template <class Scratch>
__global__ void __launch_bounds__(kSelThreads) rpn_chunk_topk_kernel(ProposalParams p, int total_chunks) {
  __shared__ Scratch s;
  const int chunk = blockIdx.x, b = blockIdx.y;
  const ChunkMap cm = chunk_of(p, chunk);
  const RpnLevel lv = p.lvl[cm.level];
  const int k = cm.n < p.K ? cm.n : p.K;
  RpnChunkKeyAt ka{lv.rpn + (size_t)b * lv.h * lv.w * kRpnCh, cm.first};
  block_topk_keys(ka, cm.n, k, s);
  unsigned long long* out = p.chunk_keys + ((size_t)b * total_chunks + chunk) * p.K;
  for (int t = threadIdx.x; t < p.K; t += blockDim.x) out[t] = t < k ? s.keys_b[t] : 0ull;
}

struct CandKeyAt {
  const unsigned long long* keys;
  __device__ __forceinline__ unsigned long long operator()(int e) const { return keys[e]; }
};

// ---- stage A2: merge the chunk winners, decode, clip, (min-size filter), ordered compaction ---
// (K <= 1024: one round, one candidate per thread; the *_big instantiation walks k in rounds of 1024)
template <class Scratch>
__global__ void __launch_bounds__(kSelThreads) rpn_select_kernel(ProposalParams p, int total_chunks) {
  __shared__ Scratch s;
  const int l = blockIdx.x, b = blockIdx.y;
  const RpnLevel lv = p.lvl[l];
  const int n = lv.h * lv.w * 3;
  const int k = n < p.K ? n : p.K;
  int first_chunk = 0;
  for (int q = 0; q < l; ++q) first_chunk += (p.lvl[q].h * p.lvl[q].w * 3 + kSelChunk - 1) / kSelChunk;
  const int nc = (n + kSelChunk - 1) / kSelChunk;
  CandKeyAt ka{p.chunk_keys + ((size_t)b * total_chunks + first_chunk) * p.K};
  // zero keys pad short chunks; real keys are > 0 and there are at least k of them
  block_topk_keys(ka, nc * p.K, k, s);
  const float* base = lv.rpn + (size_t)b * lv.h * lv.w * kRpnCh;

  const size_t o = ((size_t)b * p.nlevels + l) * p.K;
  int run = 0;
  for (int r0 = 0; r0 < k; r0 += kSelThreads) {
    const int tid = r0 + (int)threadIdx.x;
    float bx[4] = {0.f, 0.f, 0.f, 0.f};
    float score = 0.f;
    int valid = 0;
    if (tid < k) {
      const unsigned long long key = s.keys_b[tid];
      const int e = (int)key64_index(key);
      const int pix = e / 3, a = e - pix * 3;
      const int y = pix / lv.w, x = pix - y * lv.w;
      const float* d = base + (size_t)pix * kRpnCh + 3 + a * 4;
      const float* an = lv.anchors + ((size_t)(y * lv.field + x) * 3 + a) * 4;
      score = base[(size_t)pix * kRpnCh + a];
      // decode_bbox_target (nn.py:1518-1538), fp32, same operand order as the oracle
      const float wa = an[2] - an[0], ha = an[3] - an[1];
      const float xa = (an[2] + an[0]) * 0.5f, ya = (an[3] + an[1]) * 0.5f;
      const float wb = expf(fminf(d[2], p.decode_clip)) * wa;
      const float hb = expf(fminf(d[3], p.decode_clip)) * ha;
      const float xb = d[0] * wa + xa, yb = d[1] * ha + ya;
      const float fw = (float)p.img_w, fh = (float)p.img_h;
      // clip_boxes (nn.py:1339-1346)
      bx[0] = fminf(fmaxf(xb - wb * 0.5f, 0.f), fw);
      bx[1] = fminf(fmaxf(yb - hb * 0.5f, 0.f), fh);
      bx[2] = fminf(fmaxf(xb + wb * 0.5f, 0.f), fw);
      bx[3] = fminf(fmaxf(yb + hb * 0.5f, 0.f), fh);
      // rpn_min_size = 0 with strict > (nn.py:1377-1378); the multibatch graph has no filter
      valid = (p.graph == 0) ? ((bx[2] - bx[0] > 0.f) && (bx[3] - bx[1] > 0.f)) : 1;
    }
    int total;
    const int pos = run + block_scan_excl(valid, s.wave_tmp, &total);
    if (valid) {
      float* cb = p.cand_boxes + (o + pos) * 4;
      cb[0] = bx[0]; cb[1] = bx[1]; cb[2] = bx[2]; cb[3] = bx[3];
      p.cand_scores[o + pos] = score;
    }
    run += total;
    __syncthreads();               // wave_tmp is reused by the next round's scan
  }
  if (threadIdx.x == 0) p.cand_count[b * p.nlevels + l] = run;
}

// ---- stage B: NMS per (image, level) ----------------------------------------------------
__global__ void __launch_bounds__(kSelThreads) rpn_nms_kernel(ProposalParams p) {
  __shared__ NmsScratch s;
  const int l = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const size_t o = ((size_t)b * p.nlevels + l) * p.K;
  const int n = p.cand_count[b * p.nlevels + l];
  for (int i = tid; i < n * 4; i += blockDim.x) s.box[i] = p.cand_boxes[o * 4 + i];
  __syncthreads();
  block_nms(n, p.K, p.nms_thresh, s);
  const int nk = s.nkeep;
  for (int i = tid; i < nk; i += blockDim.x) {
    const int c = s.keep[i];
    const float* src = p.cand_boxes + (o + c) * 4;   // un-normalised original corners
    float* dst = p.lvl_boxes + (o + i) * 4;
    dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2]; dst[3] = src[3];
    p.lvl_scores[o + i] = p.cand_scores[o + c];
  }
  if (tid == 0) p.lvl_count[b * p.nlevels + l] = nk;
}

// K > 1024: the same selection through the paneled walk (select_device.hpp); boxes are read from the workspace
struct CandBoxAt {
  const float* boxes;
  __device__ __forceinline__ void operator()(int i, float* out) const {
    out[0] = boxes[i * 4 + 0]; out[1] = boxes[i * 4 + 1]; out[2] = boxes[i * 4 + 2]; out[3] = boxes[i * 4 + 3];
  }
};
__global__ void __launch_bounds__(kSelThreads) rpn_nms_big_kernel(ProposalParams p) {
  __shared__ __attribute__((aligned(16))) char raw[sizeof(NmsBigScratch)];
  NmsBigScratch& s = *reinterpret_cast<NmsBigScratch*>(raw);
  const int l = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const size_t o = ((size_t)b * p.nlevels + l) * p.K;
  const int n = p.cand_count[b * p.nlevels + l];
  block_nms_paneled(n, p.K, p.nms_thresh, CandBoxAt{p.cand_boxes + o * 4}, s);
  const int nk = s.nkeep;
  for (int i = tid; i < nk; i += blockDim.x) {
    const int c = s.keep[i];
    const float* src = p.cand_boxes + (o + c) * 4;
    float* dst = p.lvl_boxes + (o + i) * 4;
    dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2]; dst[3] = src[3];
    p.lvl_scores[o + i] = p.cand_scores[o + c];
  }
  if (tid == 0) p.lvl_count[b * p.nlevels + l] = nk;
}

// ---- stage C: cross-level top-k (models.py:429-434 / :2487-2522) ---------------------------
// Entries are 2*L sorted lists per image (per level: survivors, then -- multibatch graph only --
// the zero padding combined_non_max_suppression appends up to K); the rank of an entry in the
// merged (score desc, concat position asc) order is the sum of binary-search counts.
struct MergeLists {
  const float* scores[5];
  int cnt[5];
  int pad[5];
  int K;
};
__device__ __forceinline__ bool ent_greater(float s1, int p1, float s2, int p2) {
  return s1 > s2 || (s1 == s2 && p1 < p2);
}
__device__ __forceinline__ int count_greater(const MergeLists& m, int L, float s, int pos) {
  int total = 0;
  for (int l = 0; l < L; ++l) {
    {  // survivors of level l: sorted by (score desc, position asc)
      int lo = 0, hi = m.cnt[l];
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (ent_greater(m.scores[l][mid], l * m.K + mid, s, pos)) lo = mid + 1; else hi = mid;
      }
      total += lo;
    }
    {  // zero padding of level l: score 0, positions l*K + cnt .. l*K + K - 1
      int lo = 0, hi = m.pad[l];
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (ent_greater(0.f, l * m.K + m.cnt[l] + mid, s, pos)) lo = mid + 1; else hi = mid;
      }
      total += lo;
    }
  }
  return total;
}

template <int CAP>
__global__ void __launch_bounds__(kSelThreads) rpn_merge_kernel(ProposalParams p) {
  __shared__ float s_box[CAP * 4];
  __shared__ int s_wave[kSelWaves + 1];
  const int b = blockIdx.x, tid = threadIdx.x, L = p.nlevels, K = p.K;
  MergeLists m;
  m.K = K;
  int n_total = 0;
  for (int l = 0; l < 5; ++l) {
    m.scores[l] = nullptr; m.cnt[l] = 0; m.pad[l] = 0;
  }
  for (int l = 0; l < L; ++l) {
    m.scores[l] = p.lvl_scores + ((size_t)b * L + l) * K;
    m.cnt[l] = p.lvl_count[b * L + l];
    m.pad[l] = (p.graph == 1) ? K - m.cnt[l] : 0;
    n_total += m.cnt[l] + m.pad[l];
  }
  const int k = n_total < K ? n_total : K;
  for (int i = tid; i < K * 4; i += blockDim.x) s_box[i] = 0.f;
  __syncthreads();
  for (int e = tid; e < L * K; e += blockDim.x) {
    const int l = e / K, j = e - l * K;
    const bool real = j < m.cnt[l];
    if (!real && j >= m.cnt[l] + m.pad[l]) continue;
    const float s = real ? m.scores[l][j] : 0.f;
    const int rank = count_greater(m, L, s, e);
    if (rank < k && real) {
      const float* src = p.lvl_boxes + (((size_t)b * L + l) * K + j) * 4;
      s_box[rank * 4 + 0] = src[0]; s_box[rank * 4 + 1] = src[1];
      s_box[rank * 4 + 2] = src[2]; s_box[rank * 4 + 3] = src[3];
    }
  }
  __syncthreads();
  // multibatch graph: drop zero-area rows (padding) keeping order (models.py:2517-2520)
  float* out = p.props + (size_t)b * K * 4;
  int run = 0;
  for (int r0 = 0; r0 < k; r0 += kSelThreads) {          // (one round for K <= 1024)
    const int t = r0 + tid;
    int keepf = 0;
    float bx[4] = {0.f, 0.f, 0.f, 0.f};
    if (t < k) {
      bx[0] = s_box[t * 4 + 0]; bx[1] = s_box[t * 4 + 1];
      bx[2] = s_box[t * 4 + 2]; bx[3] = s_box[t * 4 + 3];
      keepf = (p.graph == 1) ? (((bx[3] - bx[1]) * (bx[2] - bx[0])) > 0.f) : 1;
    }
    int total;
    const int pos = run + block_scan_excl(keepf, s_wave, &total);
    if (keepf) {
      out[pos * 4 + 0] = bx[0]; out[pos * 4 + 1] = bx[1];
      out[pos * 4 + 2] = bx[2]; out[pos * 4 + 3] = bx[3];
    }
    run += total;
    __syncthreads();
  }
  for (int i = run + tid; i < K; i += blockDim.x) {   // deterministic tail
    out[i * 4 + 0] = 0.f; out[i * 4 + 1] = 0.f; out[i * 4 + 2] = 0.f; out[i * 4 + 3] = 0.f;
  }
  if (tid == 0) p.nprops[b] = run;
}

// ---- stand-alone top-k / NMS (parity entry points) ---------------------------------------
struct PlainScoreAt {
  const float* base;
  __device__ __forceinline__ float operator()(int e) const { return base[e]; }
};
template <class Scratch>
__global__ void __launch_bounds__(kSelThreads) topk_kernel(const float* scores, int n, int k,
                                                           int* idx_out) {
  __shared__ Scratch s;
  PlainScoreAt sc{scores};
  block_topk(sc, n, k, s);
  for (int t = threadIdx.x; t < k; t += blockDim.x) idx_out[t] = (int)key64_index(s.keys_b[t]);
}

__global__ void __launch_bounds__(kSelThreads) nms_kernel(const float* boxes, const float* scores,
                                                          int n, int max_out, float thresh,
                                                          int* idx_out, int* n_out) {
  __shared__ __attribute__((aligned(16))) char raw[sizeof(NmsScratch)];
  NmsScratch& s = *reinterpret_cast<NmsScratch*>(raw);
  // sort by (score desc, index asc): rank sort on 64-bit keys staged in the (not yet used)
  // bitmask area
  unsigned long long* keys = s.mask;
  int* order = reinterpret_cast<int*>(s.mask + kMaxTopK);
  const int tid = threadIdx.x;
  if (tid < n) keys[tid] = make_key64(scores[tid], (unsigned)tid);
  __syncthreads();
  if (tid < n) {
    const unsigned long long my = keys[tid];
    int rank = 0;
    for (int j = 0; j < n; ++j) rank += keys[j] > my ? 1 : 0;
    order[rank] = tid;
  }
  __syncthreads();
  const int src = (tid < n) ? order[tid] : 0;   // sorted position tid -> original index
  __shared__ int s_orig[kMaxTopK];
  if (tid < n) s_orig[tid] = src;
  __syncthreads();                               // keys/order (aliasing the bitmask) are dead now
  if (tid < n)
    for (int q = 0; q < 4; ++q) s.box[tid * 4 + q] = boxes[src * 4 + q];
  __syncthreads();
  block_nms(n, max_out, thresh, s);
  const int nk = s.nkeep;
  for (int i = tid; i < nk; i += blockDim.x) idx_out[i] = s_orig[s.keep[i]];
  if (tid == 0) *n_out = nk;
}


// n > 1024 candidates: rank sort into global-order indices, then the paneled walk
struct OrderedBoxAt {
  const float* boxes;
  const int* order;
  __device__ __forceinline__ void operator()(int i, float* out) const {
    const int src = order[i];
    out[0] = boxes[src * 4 + 0]; out[1] = boxes[src * 4 + 1]; out[2] = boxes[src * 4 + 2]; out[3] = boxes[src * 4 + 3];
  }
};
__global__ void __launch_bounds__(kSelThreads) nms_big_kernel(const float* boxes, const float* scores,
                                                              int n, int max_out, float thresh,
                                                              int* idx_out, int* n_out) {
  __shared__ __attribute__((aligned(16))) char raw[sizeof(NmsBigScratch)];
  __shared__ int s_orig[kMaxTopKBig];
  NmsBigScratch& s = *reinterpret_cast<NmsBigScratch*>(raw);
  unsigned long long* keys = reinterpret_cast<unsigned long long*>(s.kbox);     // 32 KiB of the 64: dead before the walk
  const int tid = threadIdx.x;
  for (int t = tid; t < n; t += blockDim.x) keys[t] = make_key64(scores[t], (unsigned)t);
  __syncthreads();
  for (int t = tid; t < n; t += blockDim.x) {
    const unsigned long long my = keys[t];
    int rank = 0;
    for (int j = 0; j < n; ++j) rank += keys[j] > my ? 1 : 0;
    s_orig[rank] = t;
  }
  __syncthreads();
  block_nms_paneled(n, max_out, thresh, OrderedBoxAt{boxes, s_orig}, s);
  const int nk = s.nkeep;
  for (int i = tid; i < nk; i += blockDim.x) idx_out[i] = s_orig[s.keep[i]];
  if (tid == 0) *n_out = nk;
}

}  // namespace

int proposal_total_chunks(const ProposalParams& p) {
  int t = 0;
  for (int l = 0; l < p.nlevels; ++l) t += (p.lvl[l].h * p.lvl[l].w * 3 + kSelChunk - 1) / kSelChunk;
  return t;
}

size_t proposal_workspace_bytes(int B, int L, int K) {
  const size_t per = (size_t)B * L * K;
  return 2 * (per * 4 * sizeof(float) + per * sizeof(float)) + 2 * (size_t)B * L * sizeof(int) + 256;
}

int launch_proposals(const ProposalParams& p, hipStream_t stream) {
  ODT_CHECK(p.K >= 1 && p.K <= kMaxTopKBig, "proposals: rpn_test_post_nms_topk must be in [1,4096]");
  ODT_CHECK(p.nlevels >= 1 && p.nlevels <= 5, "proposals: 1..5 levels");
  const int tc = proposal_total_chunks(p);
  ODT_CHECK(p.chunk_keys != nullptr, "proposals: chunk_keys workspace missing");
  if (p.K <= kMaxTopK) {
    hipLaunchKernelGGL(rpn_chunk_topk_kernel<TopkScratch>, dim3(tc, p.B), dim3(kSelThreads), 0, stream, p, tc);
    hipLaunchKernelGGL(rpn_select_kernel<TopkScratch>, dim3(p.nlevels, p.B), dim3(kSelThreads), 0, stream, p, tc);
    hipLaunchKernelGGL(rpn_nms_kernel, dim3(p.nlevels, p.B), dim3(kSelThreads), 0, stream, p);
    hipLaunchKernelGGL(rpn_merge_kernel<kMaxTopK>, dim3(p.B), dim3(kSelThreads), 0, stream, p);
  } else {            // the script accepts any --rpn_test_post_nms_topk (reference obj_detect_tracking.py:132)
    hipLaunchKernelGGL(rpn_chunk_topk_kernel<TopkScratchBig>, dim3(tc, p.B), dim3(kSelThreads), 0, stream, p, tc);
    hipLaunchKernelGGL(rpn_select_kernel<TopkScratchBig>, dim3(p.nlevels, p.B), dim3(kSelThreads), 0, stream, p, tc);
    hipLaunchKernelGGL(rpn_nms_big_kernel, dim3(p.nlevels, p.B), dim3(kSelThreads), 0, stream, p);
    hipLaunchKernelGGL(rpn_merge_kernel<kMaxTopKBig>, dim3(p.B), dim3(kSelThreads), 0, stream, p);
  }
  ODT_HIP(hipGetLastError());
  return 0;
}

int launch_topk(const float* scores, int n, int k, int* idx_out, hipStream_t stream) {
  ODT_CHECK(k >= 1 && k <= kMaxTopKBig && k <= n, "topk: need 1 <= k <= min(n,4096)");
  if (k <= kMaxTopK)
    hipLaunchKernelGGL(topk_kernel<TopkScratch>, dim3(1), dim3(kSelThreads), 0, stream, scores, n, k, idx_out);
  else
    hipLaunchKernelGGL(topk_kernel<TopkScratchBig>, dim3(1), dim3(kSelThreads), 0, stream, scores, n, k, idx_out);
  ODT_HIP(hipGetLastError());
  return 0;
}

int launch_nms(const float* boxes, const float* scores, int n, int max_out, float thresh,
               int* idx_out, int* n_out, hipStream_t stream) {
  ODT_CHECK(n >= 0 && n <= kMaxTopKBig, "nms: at most 4096 candidates");
  if (n <= kMaxTopK)
    hipLaunchKernelGGL(nms_kernel, dim3(1), dim3(kSelThreads), 0, stream, boxes, scores, n, max_out,
                       thresh, idx_out, n_out);
  else
    hipLaunchKernelGGL(nms_big_kernel, dim3(1), dim3(kSelThreads), 0, stream, boxes, scores, n, max_out,
                       thresh, idx_out, n_out);
  ODT_HIP(hipGetLastError());
  return 0;
}

}  // namespace odt
