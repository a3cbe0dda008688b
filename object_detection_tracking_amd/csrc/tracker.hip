// DeepSORT appearance matching on gfx950: cosine nearest-neighbour distance between every
// track's gallery and the frame's detections, ONE kernel per call.
//
// Restates reference deep_sort/nn_matching.py:31-54 (_cosine_distance: L2-normalise both
// sides, 1 - a.b^T), :78-96 (_nn_cosine_distance: min over the track's gallery rows) and
// :156-177 (NearestNeighborDistanceMetric.distance: the per-track Python loop).  fp32 math,
// float64 cost matrix like the reference.
//
// Latency-bound (T = 64 tracks x budget 5 rows x N = 100 detections x D = 256: 8 MFLOP), so the
// shape is: one workgroup per track (4 waves).  The track's gallery rows are normalised on the way
// into LDS (one wave per row, wavefront shuffles for the norm); then every wave takes detections
// j = wave, wave + 4, ...: the detection row is read coalesced (D floats = one 1-KB row per wave
// instruction at D = 256), normalised in registers / the wave's private LDS strip, and dotted
// against every staged gallery row with lane-strided partial sums + a shuffle tree; min over the
// rows, 1 - dot, one float64 store per (track, detection).  Galleries larger than the LDS chunk
// (no budget: the gallery grows by one row per matched frame) are walked in chunks of kRows rows.
// Next to a busy detector (round 5): the conv kernels leave neither registers nor LDS for a second resident workgroup, so
// this kernel's workgroups take whole CUs at a kernel boundary and hold back the next conv launch's workgroups for as long
// as they run.  Hence (i) the detections of a track are split over grid.y workgroups (four for N >= 32: a quarter of the
// time per workgroup; the gallery rows are normalised by each -- five rows), (ii) the LDS staging is sized for the feature
// length (D <= 256: 20 KB instead of 80 KB).
#include <cstring>

#include "odt_common.hpp"
#include <algorithm>
#include <chrono>
#include <cstdio>

namespace odt {
namespace {

constexpr int kRows = 16;        // gallery rows staged per chunk
constexpr int kMaxD = 1024;      // feature length bound of the LDS staging (the box head's is 256)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  return v;
}

template <int DMAX>
__global__ void __launch_bounds__(256) nn_cosine_kernel(const float* __restrict__ gal, const int* __restrict__ seg,
                                                        const float* __restrict__ det, int N, int D,
                                                        double* __restrict__ cost) {
  __shared__ float rows[kRows * DMAX];
  __shared__ float dstrip[4 * DMAX];
  const int t = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int jstep = 4 * (int)gridDim.y, j0 = wave + 4 * (int)blockIdx.y;      // this workgroup's detections: j0, j0 + jstep, ...
  const int g0 = seg[t], g1 = seg[t + 1];
  float* mine = dstrip + wave * DMAX;
  for (int c0 = g0; c0 < g1; c0 += kRows) {
    const int nr = g1 - c0 < kRows ? g1 - c0 : kRows;
    if (c0 > g0) __syncthreads();                       // the previous chunk has been consumed
    for (int r = wave; r < nr; r += 4) {                // a / ||a|| (nn_matching.py:48-50), one wave per row
      const float* src = gal + (size_t)(c0 + r) * D;
      float ss = 0.f;
      for (int d = lane; d < D; d += 64) ss += src[d] * src[d];
      const float nrm = sqrtf(wave_sum(ss));
      for (int d = lane; d < D; d += 64) rows[r * DMAX + d] = src[d] / nrm;
    }
    __syncthreads();
    for (int j = j0; j < N; j += jstep) {
      const float* dj = det + (size_t)j * D;
      float ss = 0.f;
      for (int d = lane; d < D; d += 64) ss += dj[d] * dj[d];
      const float nrm = sqrtf(wave_sum(ss));
      for (int d = lane; d < D; d += 64) mine[d] = dj[d] / nrm;     // wave-private strip: same lanes read it back
      float best = 3.402823466e38f;
      for (int r = 0; r < nr; ++r) {
        float dot = 0.f;
        for (int d = lane; d < D; d += 64) dot += rows[r * DMAX + d] * mine[d];
        best = fminf(best, 1.0f - wave_sum(dot));
      }
      if (lane == 0) {
        double* o = cost + (size_t)t * N + j;
        *o = c0 == g0 ? (double)best : fmin(*o, (double)best);
      }
    }
  }
}

}  // namespace

int launch_nn_cosine(const float* gallery, const int* seg, int T, const float* dets, int N, int D, double* cost,
                     hipStream_t stream) {
  ODT_CHECK(D > 0 && D <= kMaxD, "nn_cosine: feature length above 1024");
  if (T > 0 && N > 0) {
    const dim3 grid(T, N >= 64 ? 8 : (N >= 32 ? 4 : 1));
    if (D <= 256) hipLaunchKernelGGL(nn_cosine_kernel<256>, grid, dim3(256), 0, stream, gallery, seg, dets, N, D, cost);
    else hipLaunchKernelGGL(nn_cosine_kernel<kMaxD>, grid, dim3(256), 0, stream, gallery, seg, dets, N, D, cost);
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
    static const bool flat = getenv("ODT_COSINE_STREAM_PRIORITY") != nullptr && getenv("ODT_COSINE_STREAM_PRIORITY")[0] == '0';   // A/B knob
    if (flat) ODT_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    else ODT_HIP(hipStreamCreateWithPriority(&stream, hipStreamNonBlocking, prio_greatest));
    ODT_HIP(hipEventCreateWithFlags(&done, hipEventDisableTiming));
  }
  // one packed input record: [seg (T+1 ints, padded to 4)][gallery G*D][detections N*D]
  const size_t nseg = ((size_t)T + 1 + 3) & ~(size_t)3;
  const size_t nin = nseg + (size_t)(G + N) * D, ncost = (size_t)T * N;
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
  static const bool timing = getenv("ODT_TRACKER_TIMING") != nullptr;      // tuning aid: where a call's wall time goes
  static double acc_t[5] = {0, 0, 0, 0, 0}; static long acc_n = 0;
  auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double c0 = timing ? now() : 0.0;
  std::memcpy(h_in, seg, ((size_t)T + 1) * sizeof(int));
  float* hg = h_in + nseg;
  for (int g = 0; g < G; ++g) std::memcpy(hg + (size_t)g * D, gal_rows[g], sizeof(float) * D);
  float* hd = hg + (size_t)G * D;
  for (int j = 0; j < N; ++j) std::memcpy(hd + (size_t)j * D, det_rows[j], sizeof(float) * D);
  const double c1 = timing ? now() : 0.0;
  ODT_HIP(hipMemcpyAsync(d_in, h_in, nin * 4, hipMemcpyHostToDevice, stream));
  const double c2 = timing ? now() : 0.0;
  if (launch_nn_cosine(d_in + nseg, (const int*)d_in, T, d_in + nseg + (size_t)G * D, N, D, d_cost, stream)) return 1;
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
