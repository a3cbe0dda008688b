// DeepSORT appearance matching on gfx950: cosine nearest-neighbour distance between every
// track's gallery and the frame's detections.
//
// Restates reference deep_sort/nn_matching.py:31-54 (_cosine_distance: L2-normalise both
// sides, 1 - a.b^T), :78-96 (_nn_cosine_distance: min over the track's gallery rows) and
// :156-177 (NearestNeighborDistanceMetric.distance: the per-track Python loop).  fp32 math,
// float64 cost matrix like the reference.  Latency-bound (a few MFLOP): one wave per row for
// the normalisation (wavefront shuffles), one workgroup per track for the segmented min.
#include "odt_common.hpp"

namespace odt {
namespace {

__global__ void __launch_bounds__(64) normalize_rows_kernel(const float* in, int D, float* out) {
  const int r = blockIdx.x, lane = threadIdx.x;
  const float* src = in + (size_t)r * D;
  float ss = 0.f;
  for (int d = lane; d < D; d += 64) ss += src[d] * src[d];
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) ss += __shfl_xor(ss, m);
  const float nrm = sqrtf(ss);
  for (int d = lane; d < D; d += 64) out[(size_t)r * D + d] = src[d] / nrm;
}

__global__ void __launch_bounds__(256) cosine_min_kernel(const float* gal_n, const int* seg,
                                                         const float* det_n, int N, int D,
                                                         double* cost) {
  const int t = blockIdx.x;
  const int g0 = seg[t], g1 = seg[t + 1];
  for (int j = threadIdx.x; j < N; j += blockDim.x) {
    const float* dj = det_n + (size_t)j * D;
    float best = 3.402823466e38f;
    for (int g = g0; g < g1; ++g) {
      const float* gr = gal_n + (size_t)g * D;
      float dot = 0.f;
      for (int d = 0; d < D; ++d) dot += gr[d] * dj[d];
      best = fminf(best, 1.0f - dot);
    }
    cost[(size_t)t * N + j] = (double)best;
  }
}

}  // namespace

int launch_nn_cosine(const float* gallery, int G, const int* seg, int T, const float* dets, int N,
                        int D, float* gal_n, float* det_n, double* cost, hipStream_t stream) {
  if (G > 0) hipLaunchKernelGGL(normalize_rows_kernel, dim3(G), dim3(64), 0, stream, gallery, D, gal_n);
  if (N > 0) hipLaunchKernelGGL(normalize_rows_kernel, dim3(N), dim3(64), 0, stream, dets, D, det_n);
  if (T > 0 && N > 0)
    hipLaunchKernelGGL(cosine_min_kernel, dim3(T), dim3(256), 0, stream, (const float*)gal_n, seg,
                       (const float*)det_n, N, D, cost);
  ODT_HIP(hipGetLastError());
  return 0;
}

}  // namespace odt
