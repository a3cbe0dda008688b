// Workgroup-level selection primitives for gfx950 (wave64, LDS resident):
//   * block_scan_excl    -- exclusive prefix sum over the workgroup (wave shuffles + LDS)
//   * block_topk         -- exact top-k of n floats (64-bit radix select in LDS histograms,
//                           then rank sort), canonical order (score desc, index asc)
//   * block_nms          -- tf.image.non_max_suppression on <= 1024 score-sorted boxes:
//                           upper-triangular 64-bit suppression bitmask in LDS (parallel IoUs),
//                           then one wave walks the candidates in order.
// These restate the semantics pinned in oracle/tfops.py (TF-1.15 top_k / NMS kernels).
// All arithmetic is fp32 with one rounding per operation (build uses -ffp-contract=off) so
// that on identical inputs the selected indices are bit-identical to the oracle's.
#pragma once
#include "odt_common.hpp"

namespace odt {

constexpr int kSelThreads = 1024;               // workgroup size of the selection kernels
constexpr int kSelWaves = kSelThreads / 64;
constexpr int kNmsWords = kMaxTopK / 64;        // 64-bit words per bitmask row

__device__ __forceinline__ unsigned sortable_key(float f) {
  f = f + 0.0f;                                  // -0.0 -> +0.0 (float compare treats them equal)
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_to_float(unsigned k) {
  const unsigned u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  return __uint_as_float(u);
}
__device__ __forceinline__ unsigned long long make_key64(float score, unsigned index) {
  return ((unsigned long long)sortable_key(score) << 32) | (unsigned long long)(0xFFFFFFFFu - index);
}
__device__ __forceinline__ unsigned key64_index(unsigned long long k) {
  return 0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull);
}

// Exclusive scan of one int per thread; returns the exclusive prefix, *total gets the sum.
// s_wave: LDS scratch of kSelWaves + 1 ints.  Contains two barriers.
__device__ __forceinline__ int block_scan_excl(int v, int* s_wave, int* total) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nwaves = (int)(blockDim.x >> 6);
  int inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int o = __shfl_up(inc, (unsigned)d);
    if (lane >= d) inc += o;
  }
  if (lane == 63) s_wave[wave] = inc;
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int w = 0; w < nwaves; ++w) {
      const int t = s_wave[w];
      s_wave[w] = run;
      run += t;
    }
    s_wave[nwaves] = run;
  }
  __syncthreads();
  *total = s_wave[nwaves];
  return s_wave[wave] + inc - v;
}

// LDS scratch block_topk needs (CAP: largest k).
template <int CAP>
struct TopkScratchT {
  static constexpr int cap = CAP;
  int hist[4096];
  unsigned long long keys_a[CAP];
  unsigned long long keys_b[CAP];
  int wave_tmp[kSelWaves + 1];
  int misc[4];
};
using TopkScratch = TopkScratchT<kMaxTopK>;          // K <= 1024: the production configurations
using TopkScratchBig = TopkScratchT<kMaxTopKBig>;    // 1024 < K <= 4096

// Exact top-k (k <= Scratch::cap, k <= n) over n unique 64-bit keys key_at(0..n-1) (see make_key64:
// score in the high word, inverted index in the low word).  On return s.keys_b[0..k) holds the
// selected keys sorted descending == (score desc, index asc).  Workgroup-uniform call.
template <class KeyAt, class Scratch>
__device__ void block_topk_keys(KeyAt key_at, int n, int k, Scratch& s) {
  const int tid = threadIdx.x, nthr = blockDim.x;
  unsigned long long prefix = 0ull, mask = 0ull;
  int need = k;
  const int shifts[6] = {52, 40, 32, 20, 8, 0};
  const int bitsn[6] = {12, 12, 8, 12, 12, 8};
  for (int d = 0; d < 6; ++d) {
    const int shift = shifts[d], nb = 1 << bitsn[d];
    for (int i = tid; i < nb; i += nthr) s.hist[i] = 0;
    __syncthreads();
    for (int e = tid; e < n; e += nthr) {
      const unsigned long long key = key_at(e);
      if ((key & mask) == prefix) atomicAdd(&s.hist[(int)((key >> shift) & (unsigned long long)(nb - 1))], 1);
    }
    __syncthreads();
    // locate the bin holding the need-th largest key: bins per thread, suffix counts
    const int per = nb >= nthr ? nb / nthr : 1;
    const int nact = nb / per;
    int mine = 0;
    if (tid < nact)
      for (int q = 0; q < per; ++q) mine += s.hist[tid * per + q];
    int total;
    const int excl = block_scan_excl(mine, s.wave_tmp, &total);
    const int above = total - excl - mine;      // keys in bins owned by higher threads
    if (tid < nact && above < need && need <= above + mine) {
      int acc = above;
      for (int q = per - 1; q >= 0; --q) {
        const int h = s.hist[tid * per + q];
        if (need <= acc + h) {
          s.misc[0] = tid * per + q;
          s.misc[1] = acc;
          s.misc[2] = h;
          break;
        }
        acc += h;
      }
    }
    __syncthreads();
    const int bin = s.misc[0], cnt_above = s.misc[1], in_bin = s.misc[2];
    need -= cnt_above;
    prefix |= (unsigned long long)bin << shift;
    mask |= (unsigned long long)(nb - 1) << shift;
    __syncthreads();
    if (in_bin == need) break;                  // everything in this bin is selected
  }
  if (tid == 0) s.misc[3] = 0;
  __syncthreads();
  for (int e = tid; e < n; e += nthr) {
    const unsigned long long key = key_at(e);
    if ((key & mask) >= prefix) {
      const int pos = atomicAdd(&s.misc[3], 1);
      if (pos < Scratch::cap) s.keys_a[pos] = key;
    }
  }
  __syncthreads();
  // rank sort (keys are unique): one key per thread
  for (int t = tid; t < k; t += nthr) {
    const unsigned long long my = s.keys_a[t];
    int rank = 0;
    for (int j = 0; j < k; ++j) rank += s.keys_a[j] > my ? 1 : 0;
    s.keys_b[rank] = my;
  }
  __syncthreads();
}

template <class ScoreAt>
struct ScoreKeyAt {
  ScoreAt score_at;
  __device__ __forceinline__ unsigned long long operator()(int e) const {
    return make_key64(score_at(e), (unsigned)e);
  }
};
// top-k of score_at(0..n-1); keys carry the element index.
template <class ScoreAt, class Scratch>
__device__ void block_topk(ScoreAt score_at, int n, int k, Scratch& s) {
  ScoreKeyAt<ScoreAt> ka{score_at};
  block_topk_keys(ka, n, k, s);
}

// ---- NMS --------------------------------------------------------------------------------
struct NmsScratch {
  unsigned long long mask[kMaxTopK * kNmsWords];   // 128 KiB: row i, word w
  float box[kMaxTopK * 4];                         // normalised (min/max) y1,x1,y2,x2 order free
  float area[kMaxTopK];
  int keep[kMaxTopK];
  int nkeep;
};

// TF-1.15 non_max_suppression_op.cc IOU() on pre-normalised boxes (a0<=a2, a1<=a3).
__device__ __forceinline__ bool iou_gt(const float* bi, float area_i, const float* bj, float area_j,
                                       float thresh) {
  if (area_i <= 0.f || area_j <= 0.f) return false;   // TF: IoU = 0
  const float y0 = fmaxf(bi[0], bj[0]), x0 = fmaxf(bi[1], bj[1]);
  const float y1 = fminf(bi[2], bj[2]), x1 = fminf(bi[3], bj[3]);
  const float inter = fmaxf(y1 - y0, 0.f) * fmaxf(x1 - x0, 0.f);
  if (inter == 0.f) return 0.f > thresh;              // IoU = 0 exactly
  const float iou = inter / ((area_i + area_j) - inter);
  return iou > thresh;
}

// Boxes must already be in s.box (any consistent axis order) for candidates 0..n-1, sorted by
// descending score.  Fills s.keep[0..s.nkeep) with the selected candidate positions.
__device__ inline void block_nms(int n, int max_out, float thresh, NmsScratch& s) {
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int nw = (n + 63) >> 6;
  for (int i = tid; i < n; i += nthr) {
    float* b = &s.box[i * 4];
    const float a0 = fminf(b[0], b[2]), a2 = fmaxf(b[0], b[2]);
    const float a1 = fminf(b[1], b[3]), a3 = fmaxf(b[1], b[3]);
    b[0] = a0; b[1] = a1; b[2] = a2; b[3] = a3;
    s.area[i] = (a2 - a0) * (a3 - a1);
  }
  __syncthreads();
  for (int idx = tid; idx < n * nw; idx += nthr) {
    const int i = idx / nw, w = idx - i * nw;
    unsigned long long bits = 0ull;
    if (w >= (i >> 6)) {
      const float* bi = &s.box[i * 4];
      const float ai = s.area[i];
      const int j0 = w << 6;
      int jb = i + 1 - j0;
      if (jb < 0) jb = 0;
      int je = n - j0;
      if (je > 64) je = 64;
      for (int q = jb; q < je; ++q) {
        const int j = j0 + q;
        if (iou_gt(bi, ai, &s.box[j * 4], s.area[j], thresh)) bits |= 1ull << q;
      }
    }
    s.mask[i * kNmsWords + w] = bits;
  }
  __syncthreads();
  if (tid < 64) {   // wave 0: lane w < nw owns word w of the "removed" set
    unsigned long long removed = 0ull;
    int nkeep = 0;
    for (int i = 0; i < n; ++i) {
      if (nkeep >= max_out) break;
      const unsigned long long r = __shfl(removed, i >> 6);
      if (!((r >> (i & 63)) & 1ull)) {
        if (tid == 0) s.keep[nkeep] = i;
        ++nkeep;
        if (tid < nw) removed |= s.mask[i * kNmsWords + tid];
      }
    }
    if (tid == 0) s.nkeep = nkeep;
  }
  __syncthreads();
}

// ---- NMS over more than kMaxTopK candidates (1024 < n <= kMaxTopKBig) ----------------------------------------------
// The K x K / 64 bitmask of block_nms does not fit LDS beyond 1024 candidates, so the score-sorted candidates are walked
// in panels of P: a panel's candidates are first tested against every box kept so far (one thread per candidate, the
// kept boxes live in LDS), then the panel's own P x P / 64 upper-triangular bitmask is built and one wave walks it as
// block_nms does.  Greedy NMS only asks "does SOME earlier kept box overlap by more than the threshold", so the result
// (and every IoU evaluated, same operand order: earlier box first) is that of the one-panel walk: identical picks.
constexpr int kNmsPanel = 512;
struct NmsBigScratch {
  unsigned long long mask[kNmsPanel * (kNmsPanel / 64)];   // 32 KiB
  float pbox[kNmsPanel * 4];                               // the panel's boxes, normalised
  float parea[kNmsPanel];
  float kbox[kMaxTopKBig * 4];                             // kept boxes, normalised (64 KiB)
  float karea[kMaxTopKBig];
  int keep[kMaxTopKBig];
  unsigned long long pre[kNmsPanel / 64];                  // panel candidates suppressed by earlier panels' picks
  int nkeep;
};

// (nms_big_kernel / class_nms_big_kernel put sizeof(int) * kMaxTopKBig of their own next to this: 154 KB of the 160 KB of LDS a
// gfx950 workgroup can have -- a larger kMaxTopKBig or panel must fail here, not at launch)
static_assert(sizeof(NmsBigScratch) + sizeof(int) * kMaxTopKBig + 64 <= 160 * 1024, "NmsBigScratch + the kernels' index arrays must fit 160 KB of LDS");

// box_at(i, out[4]): corners of candidate i (0..n-1, score-descending).  Fills s.keep[0..s.nkeep).
template <class BoxAt>
__device__ inline void block_nms_paneled(int n, int max_out, float thresh, BoxAt box_at, NmsBigScratch& s) {
  const int tid = threadIdx.x, nthr = blockDim.x;
  constexpr int P = kNmsPanel, PW = kNmsPanel / 64;
  if (tid == 0) s.nkeep = 0;
  __syncthreads();
  for (int base = 0; base < n; base += P) {
    const int m = n - base < P ? n - base : P;
    const int nw = (m + 63) >> 6;
    const int kept = s.nkeep;                    // uniform: written before the barrier that ended the last round
    if (kept >= max_out) break;
    if (tid < m) {
      float b[4];
      box_at(base + tid, b);
      const float a0 = fminf(b[0], b[2]), a2 = fmaxf(b[0], b[2]);
      const float a1 = fminf(b[1], b[3]), a3 = fmaxf(b[1], b[3]);
      s.pbox[tid * 4 + 0] = a0; s.pbox[tid * 4 + 1] = a1; s.pbox[tid * 4 + 2] = a2; s.pbox[tid * 4 + 3] = a3;
      s.parea[tid] = (a2 - a0) * (a3 - a1);
    }
    __syncthreads();
    if (tid < P) {                               // whole waves: P is a multiple of 64
      int gone = 0;
      if (tid < m) {
        const float* bj = &s.pbox[tid * 4];
        const float aj = s.parea[tid];
        for (int q = 0; q < kept && !gone; ++q) gone = iou_gt(&s.kbox[q * 4], s.karea[q], bj, aj, thresh) ? 1 : 0;
      }
      const unsigned long long word = __ballot(gone);
      if ((tid & 63) == 0) s.pre[tid >> 6] = word;
    }
    for (int idx = tid; idx < m * nw; idx += nthr) {
      const int i = idx / nw, w = idx - i * nw;
      unsigned long long bits = 0ull;
      if (w >= (i >> 6)) {
        const float* bi = &s.pbox[i * 4];
        const float ai = s.parea[i];
        const int j0 = w << 6;
        int jb = i + 1 - j0;
        if (jb < 0) jb = 0;
        int je = m - j0;
        if (je > 64) je = 64;
        for (int q = jb; q < je; ++q) {
          const int j = j0 + q;
          if (iou_gt(bi, ai, &s.pbox[j * 4], s.parea[j], thresh)) bits |= 1ull << q;
        }
      }
      s.mask[i * PW + w] = bits;
    }
    __syncthreads();
    if (tid < 64) {   // wave 0: lane w < nw owns word w of the panel's "removed" set
      unsigned long long removed = tid < nw ? s.pre[tid] : 0ull;
      int nkeep = kept;
      for (int i = 0; i < m; ++i) {
        if (nkeep >= max_out) break;
        const unsigned long long r = __shfl(removed, i >> 6);
        if (!((r >> (i & 63)) & 1ull)) {
          if (tid == 0) s.keep[nkeep] = base + i;
          if (tid < 4) s.kbox[nkeep * 4 + tid] = s.pbox[i * 4 + tid];
          if (tid == 4) s.karea[nkeep] = s.parea[i];
          ++nkeep;
          if (tid < nw) removed |= s.mask[i * PW + tid];
        }
      }
      if (tid == 0) s.nkeep = nkeep;
    }
    __syncthreads();
  }
  __syncthreads();
}

}  // namespace odt
