// DeepSORT appearance matching on gfx950: cosine nearest-neighbour distance between every
// track's gallery and the frame's detections, ONE kernel per call.
//
// Restates reference deep_sort/nn_matching.py:31-54 (_cosine_distance: L2-normalise both
// sides, 1 - a.b^T), :78-96 (_nn_cosine_distance: min over the track's gallery rows) and
// :156-177 (NearestNeighborDistanceMetric.distance: the per-track Python loop).  fp32 math,
// float64 cost matrix like the reference.
//
// Latency-bound (T = 64 tracks x budget 5 rows x N = 100 detections x D = 256: 8 MFLOP) and run NEXT TO a busy detector whose
// conv launches leave neither registers nor LDS for a second resident workgroup: this kernel's workgroups take whole CUs at a
// kernel boundary and hold back the next conv launch's workgroups for as long as they run.  Round 5 ran one workgroup per
// (track, quarter of the detections): 256-512 workgroups, each re-normalising the detections it visited and reducing every dot
// product with a shuffle tree (147 us next to the detector).  Round 6: the cost matrix as a small GEMM on the exact-f32 matrix
// instruction --
//   * the host packs the gallery rows (already contiguous per track) into BLOCKS of <= 32 rows made of whole tracks (a track
//     with more than 32 rows -- no budget -- becomes blocks of its own, flagged, whose minima meet through an atomic min);
//   * one wave = one 32 x 32 block of the [gallery rows x detections] dot products: v_mfma_f32_32x32x2_f32 (f32 products, f32
//     accumulation: an fmaf chain over k), operands straight from global memory as 16-byte chunks (lane (r, h) walks the
//     chunks 2 q + h of its row: the k order is a fixed permutation, the same for both operands), two batches of eight chunks
//     in flight; the squared norms are summed from the same registers -- every row and every detection is normalised ONCE per
//     wave that uses it, not once per (track, detection) pair;
//   * 1 - dot / (|a| |b|) to an LDS tile, then a walk over the block's tracks: min over a track's rows, one float64 store per
//     (track, detection).
// T = 64 x budget 5, N = 100: 11 workgroups of 4 waves (one per 32 detections) instead of 512, no shuffle trees.
#include <cstring>

#include "odt_common.hpp"
#include <algorithm>
#include <chrono>
#include <cstdio>

namespace odt {
namespace {

constexpr int kMaxD = 1024;      // feature length bound (the box head's is 256)
constexpr int kCosU = 8;         // 16-byte chunks per operand and batch


__device__ __forceinline__ void atomic_min_f64(double* o, double v) {
  unsigned long long* p = reinterpret_cast<unsigned long long*>(o);
  unsigned long long old = *p;
  while (__longlong_as_double((long long)old) > v) {
    const unsigned long long assumed = old;
    old = atomicCAS(p, assumed, (unsigned long long)__double_as_longlong(v));
    if (old == assumed) break;
  }
}

// blocks[4 b ...] = {first gallery row, rows (<= 32), first track, 1: the rows are a PART of that one track}
template <bool VEC>
__global__ void __launch_bounds__(256) nn_cosine_kernel(const float* __restrict__ gal, const int* __restrict__ seg,
                                                        const float* __restrict__ det, const int* __restrict__ blocks,
                                                        int N, int D, double* __restrict__ cost) {
  __shared__ float dist[4][32 * 33];
  __shared__ float nrm[4][32];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
  const int row0 = blocks[4 * blockIdx.x], nrows = blocks[4 * blockIdx.x + 1], t0 = blocks[4 * blockIdx.x + 2];
  const bool part = blocks[4 * blockIdx.x + 3] != 0;
  const int c0 = ((int)blockIdx.y * 4 + wave) * 32;                 // this wave's 32 detections (none: the wave idles to the barriers)
  const bool aok = r < nrows && c0 < N, bok = c0 + r < N;
  const float* arow = gal + (size_t)(row0 + (aok ? r : 0)) * D;
  const float* brow = det + (size_t)(bok ? c0 + r : 0) * D;
  auto ld = [&](const float* row, bool ok, int q) -> f32x4 {
    const int k = (2 * q + h) * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (ok) {
      if constexpr (VEC) { if (k < D) v = *reinterpret_cast<const f32x4*>(row + k); }
      else {
#pragma unroll
        for (int e = 0; e < 4; ++e) if (k + e < D) v[e] = row[k + e];
      }
    }
    return v;
  };
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  float ssa = 0.f, ssb = 0.f;
  f32x4 a0[kCosU], b0[kCosU], a1[kCosU], b1[kCosU];
  auto fetch = [&](f32x4 (&a)[kCosU], f32x4 (&b)[kCosU], int q0) {
#pragma unroll
    for (int u = 0; u < kCosU; ++u) { a[u] = ld(arow, aok, q0 + u); b[u] = ld(brow, bok, q0 + u); }
  };
  auto mac = [&](const f32x4 (&a)[kCosU], const f32x4 (&b)[kCosU]) {
#pragma unroll
    for (int u = 0; u < kCosU; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        ssa = fmaf(a[u][e], a[u][e], ssa);
        ssb = fmaf(b[u][e], b[u][e], ssb);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][e], b[u][e], acc, 0, 0, 0);
      }
  };
  const int nq = (D + 7) >> 3;
  fetch(a0, b0, 0);
  for (int q0 = 0; q0 < nq; q0 += 2 * kCosU) {
    fetch(a1, b1, q0 + kCosU);                 // (chunks past the row read as zeros)
    mac(a0, b0);
    fetch(a0, b0, q0 + 2 * kCosU);
    mac(a1, b1);
  }
  ssa += __shfl_xor(ssa, 32); ssb += __shfl_xor(ssb, 32);
  const float nb = sqrtf(ssb);
  if (h == 0) nrm[wave][r] = sqrtf(ssa);
  __syncthreads();
  // accumulator register 4 g + e of lane (r, h): gallery row 8 g + 4 h + e, detection c0 + r
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int row = 8 * g + 4 * h + e;
      dist[wave][row * 33 + r] = 1.0f - acc[4 * g + e] / (nrm[wave][row] * nb);      // nn_matching.py:48-54
    }
  __syncthreads();
  if (bok) {
    // the block's tracks in order, even ones (counted from the block's first) by the lanes h = 0, odd ones by h = 1
    int t = t0, row = 0;
    while (row < nrows) {
      int end = part ? nrows : seg[t + 1] - row0;
      end = end < nrows ? end : nrows;
      if (((t - t0) & 1) == h) {
        float m = dist[wave][row * 33 + r];
        for (int q = row + 1; q < end; ++q) m = fminf(m, dist[wave][q * 33 + r]);       // nn_matching.py:95-96
        double* o = cost + (size_t)t * N + c0 + r;
        if (part) atomic_min_f64(o, (double)m); else *o = (double)m;
      }
      row = end; ++t;
    }
  }
}

}  // namespace

// host side of the block table (see the kernel): appends 4 ints per block, returns whether any track spans blocks
bool nn_cosine_blocks(const int* seg, int T, std::vector<int>* blocks) {
  bool any_part = false;
  int t = 0;
  while (t < T) {
    const int n = seg[t + 1] - seg[t];
    if (n > 32) {
      for (int r0 = seg[t]; r0 < seg[t + 1]; r0 += 32) {
        blocks->insert(blocks->end(), {r0, std::min(32, seg[t + 1] - r0), t, 1});
      }
      any_part = true; ++t;
      continue;
    }
    const int row0 = seg[t], t0 = t;
    int rows = 0;
    while (t < T && seg[t + 1] - seg[t] <= 32 && rows + (seg[t + 1] - seg[t]) <= 32) { rows += seg[t + 1] - seg[t]; ++t; }
    blocks->insert(blocks->end(), {row0, rows, t0, 0});
  }
  return any_part;
}

int launch_nn_cosine(const float* gallery, const int* seg, const int* blocks, int nblocks, bool any_part, int T, const float* dets, int N,
                     int D, double* cost, hipStream_t stream) {
  ODT_CHECK(D > 0 && D <= kMaxD, "nn_cosine: feature length above 1024");
  if (T > 0 && N > 0 && nblocks > 0) {
    // (a track cut into several blocks: its minima meet through atomic_min_f64 -- every byte 0x7f is 1.4e306 as a double)
    if (any_part) ODT_HIP(hipMemsetAsync(cost, 0x7f, (size_t)T * N * sizeof(double), stream));
    const dim3 grid(nblocks, (N + 127) / 128);
    if (D % 4 == 0) hipLaunchKernelGGL(nn_cosine_kernel<true>, grid, dim3(256), 0, stream, gallery, seg, dets, blocks, N, D, cost);
    else hipLaunchKernelGGL(nn_cosine_kernel<false>, grid, dim3(256), 0, stream, gallery, seg, dets, blocks, N, D, cost);
  }
  ODT_HIP(hipGetLastError());
  return 0;
}

// ---- CosineCtx: everything a host-to-host call needs, allocated once and grown on demand: pinned staging,
// device buffers, a non-blocking stream of its own (a tracker must never serialise against a detector's
// forward on the null stream) and an event to wait on.  No hipMalloc / hipFree / hipDeviceSynchronize per call.
CosineCtx::~CosineCtx() {
  if (device < 0) return;
  (void)hipSetDevice(device);
  if (h_in) (void)hipHostFree(h_in);
  if (h_cost) (void)hipHostFree(h_cost);
  if (d_in) (void)hipFree(d_in);
  if (d_cost) (void)hipFree(d_cost);
  for (void* q : retired_host) (void)hipHostFree(q);
  for (void* q : retired_dev) (void)hipFree(q);
  if (done) (void)hipEventDestroy(done);
  if (stream) (void)hipStreamDestroy(stream);
}

int CosineCtx::run(int dev, const float* const* gal_rows, int G, const int* seg, int T, const float* const* det_rows,
                   int N, int D, double* cost) {
  if (T == 0 || N == 0) return 0;
  ODT_HIP(hipSetDevice(dev));
  if (device != dev) {
    ODT_CHECK(device < 0, "cosine context moved between devices");
    device = dev;
    // The highest stream priority, for the hardware queue it comes with: the runtime multiplexes all streams of one
    // priority onto a few hardware queues in creation order, and a tracker stream that lands on the queue of a
    // detector's stream waits for the whole forward in flight (seen in bench.py, where earlier handles had shifted the
    // assignment: 6 ms of "tracking" per frame instead of 0.5).  Streams of another priority get queues of their own.
    int prio_least = 0, prio_greatest = 0;
    ODT_HIP(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
    knobs_reload();
    const bool flat = env_knob_off(K_COSINE_STREAM_PRIORITY);   // A/B knob
    if (env_knob(K_COSINE_STREAM_PRIORITY).set && env_knob(K_COSINE_STREAM_PRIORITY).i < 0)      // A/B: the LOWEST priority (its own queues too)
      ODT_HIP(hipStreamCreateWithPriority(&stream, hipStreamNonBlocking, prio_least));
    else if (flat) ODT_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    else ODT_HIP(hipStreamCreateWithPriority(&stream, hipStreamNonBlocking, prio_greatest));
    ODT_HIP(hipEventCreateWithFlags(&done, hipEventDisableTiming));
  }
  // one packed input record: [seg (T+1 ints, padded to 4)][gallery G*D][detections N*D][block table, 4 ints per block]
  std::vector<int>& blocks = blocks_scratch;
  blocks.clear();
  const bool any_part = nn_cosine_blocks(seg, T, &blocks);
  const size_t nseg = ((size_t)T + 1 + 3) & ~(size_t)3;
  const size_t nblk_at = nseg + (size_t)(G + N) * D;
  const size_t nin = nblk_at + blocks.size(), ncost = (size_t)T * N;
  // Capacity: sized generously at first use and doubled when outgrown; an outgrown buffer is only retired -- hipFree /
  // hipHostFree wait for the whole device, i.e. for the detector's forward in flight (measured: 1.9 ms per update,
  // averaged, while the buffers of a young tracker grew next to a busy detector; 0.13 ms with this).
  if (cap_in < nin) {
    if (h_in) retired_host.push_back(h_in);
    if (d_in) retired_dev.push_back(d_in);
    h_in = nullptr; d_in = nullptr; cap_in = 0;
    const size_t want = std::max<size_t>(2 * nin, (size_t)1 << 20);
    ODT_HIP(hipHostMalloc((void**)&h_in, want * 4, 0));
    ODT_HIP(hipMalloc((void**)&d_in, want * 4));
    cap_in = want;
  }
  if (cap_cost < ncost) {
    if (h_cost) retired_host.push_back(h_cost);
    if (d_cost) retired_dev.push_back(d_cost);
    h_cost = nullptr; d_cost = nullptr; cap_cost = 0;
    const size_t want = std::max<size_t>(2 * ncost, (size_t)1 << 18);
    ODT_HIP(hipHostMalloc((void**)&h_cost, want * 8, 0));
    ODT_HIP(hipMalloc((void**)&d_cost, want * 8));
    cap_cost = want;
  }
  static const bool timing = env_knob(K_TRACKER_TIMING).set;      // tuning aid: where a call's wall time goes
  static double acc_t[5] = {0, 0, 0, 0, 0}; static long acc_n = 0;
  auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double c0 = timing ? now() : 0.0;
  std::memcpy(h_in, seg, ((size_t)T + 1) * sizeof(int));
  float* hg = h_in + nseg;
  for (int g = 0; g < G; ++g) std::memcpy(hg + (size_t)g * D, gal_rows[g], sizeof(float) * D);
  float* hd = hg + (size_t)G * D;
  for (int j = 0; j < N; ++j) std::memcpy(hd + (size_t)j * D, det_rows[j], sizeof(float) * D);
  std::memcpy(h_in + nblk_at, blocks.data(), blocks.size() * sizeof(int));
  const double c1 = timing ? now() : 0.0;
  ODT_HIP(hipMemcpyAsync(d_in, h_in, nin * 4, hipMemcpyHostToDevice, stream));
  const double c2 = timing ? now() : 0.0;
  if (launch_nn_cosine(d_in + nseg, (const int*)d_in, (const int*)(d_in + nblk_at), (int)(blocks.size() / 4), any_part, T, d_in + nseg + (size_t)G * D, N, D,
                       d_cost, stream)) return 1;
  const double c3 = timing ? now() : 0.0;
  ODT_HIP(hipMemcpyAsync(h_cost, d_cost, ncost * 8, hipMemcpyDeviceToHost, stream));
  ODT_HIP(hipEventRecord(done, stream));
  const double c4 = timing ? now() : 0.0;
  ODT_HIP(hipEventSynchronize(done));
  const double c5 = timing ? now() : 0.0;
  std::memcpy(cost, h_cost, ncost * 8);
  if (timing) {
    acc_t[0] += c1 - c0; acc_t[1] += c2 - c1; acc_t[2] += c3 - c2; acc_t[3] += c4 - c3; acc_t[4] += c5 - c4;
    if (++acc_n % 160 == 0) {
      fprintf(stderr, "[cosine] per call over 160 (us): pack %.1f  h2d-enqueue %.1f  launch %.1f  d2h-enqueue+record %.1f  wait %.1f   (G=%d T=%d N=%d)\n",
              acc_t[0] / 160 * 1e6, acc_t[1] / 160 * 1e6, acc_t[2] / 160 * 1e6, acc_t[3] / 160 * 1e6, acc_t[4] / 160 * 1e6, G, T, N);
      for (double& v : acc_t) v = 0;
    }
  }
  return 0;
}

}  // namespace odt
