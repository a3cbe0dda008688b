// Implicit-GEMM convolution for gfx950 (CDNA4): NHWC fp32 activations,
// [Cout][kh][kw][Cin] weights, exact-f32 MFMA (v_mfma_f32_32x32x2_f32),
// fused bias(+folded BN) + residual / nearest-2x top-down add + ReLU epilogue.
//
// Replaces tf.nn.conv2d + tf.nn.batch_normalization + relu/add on the path
// (reference nn.py:337-381, :459-521, :947-1014, :1771-1774) and tf.matmul of
// the box head (nn.py:730-774; an FC is a 1x1 conv over "pixels" = RoIs).
//
// GEMM view: C[m][n] = sum_k A[m][k] * W[n][k],  m = (img,ho,wo), n = cout,
// k = (kh,kw,ci).  Both operands are K-contiguous, so a BK=32 slice of either
// is one 128-byte row read (8 lanes x 16 B), gathered on the fly for A.
//
// Tiling (wave64): 256 threads = 4 waves as WM x WN, each wave owns TM x TN
// MFMA tiles of 32x32 -> block tile BM x BN = (WM*TM*32) x (WN*TN*32), BK = 32.
// LDS: two stages of [BM+BN][36] floats (row stride 144 B = 128 + one 16-B
// access: conflict-free ds_read_b128 over the 16-lane groups and conflict-free
// ds_write_b128).  K order inside a 32-slice is permuted (k = 8g + 4*(lane>>5)
// + t) for BOTH operands so one ds_read_b128 feeds four MFMA k-steps.
// Pipeline: global->register prefetch of slice c+1 is issued before the 16*TM*TN
// MFMAs on slice c, written to the other LDS stage after them; one barrier
// per slice.
#include <cstdlib>

#include "odt_common.hpp"

namespace odt {

namespace {

constexpr int LS = 36;  // LDS row stride in floats (32 + 4 pad)

template <int WM, int WN, int TM, int TN>
__global__ void __launch_bounds__(256) conv_igemm_kernel(ConvParams p) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int RA = BM / 32, RB = BN / 32;
  static_assert(WM * WN == 4, "4 waves");
  __shared__ __attribute__((aligned(16))) float lds[2][(BM + BN) * LS];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int ntn = (p.Cout + BN - 1) / BN;
  const int mt = blockIdx.x / ntn, nt = blockIdx.x - mt * ntn;
  const int m0 = mt * BM, n0 = nt * BN;
  const int HoWo = p.Ho * p.Wo;
  const int M = p.B * HoWo;
  const int K = p.kh * p.kw * p.Cin;
  const int cpt = p.Cin >> 5;               // 32-channel slices per tap
  const int nslices = p.kh * p.kw * cpt;

  // ---- loader role: thread -> (row lr + 32*j, 16-byte column lc)
  const int lc = tid & 7, lr = tid >> 3;
  int a_hi0[RA], a_wi0[RA], a_img[RA];
  bool a_ok[RA];
#pragma unroll
  for (int j = 0; j < RA; ++j) {
    const int m = m0 + lr + 32 * j;
    a_ok[j] = m < M;
    const int mm = a_ok[j] ? m : 0;
    const int n = mm / HoWo, r = mm - n * HoWo;
    const int ho = r / p.Wo, wo = r - ho * p.Wo;
    a_hi0[j] = ho * p.stride - p.pad_t;
    a_wi0[j] = wo * p.stride - p.pad_l;
    a_img[j] = n * p.in_Ha * p.in_Wa;
  }
  const float* b_ptr[RB];
  bool b_ok[RB];
#pragma unroll
  for (int j = 0; j < RB; ++j) {
    const int n = n0 + lr + 32 * j;
    b_ok[j] = n < p.Cout;
    b_ptr[j] = p.wt + (size_t)(b_ok[j] ? n : 0) * K + lc * 4;
  }

  f32x4 ra[RA], rb[RB];
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  auto load_slice = [&](int c) {
    const int tap = c / cpt, cc = c - tap * cpt;
    const int khh = tap / p.kw, kww = tap - khh * p.kw;
#pragma unroll
    for (int j = 0; j < RA; ++j) {
      const int hi = a_hi0[j] + khh * p.dil, wi = a_wi0[j] + kww * p.dil;
      const bool v = a_ok[j] && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
      const size_t off = (size_t)(a_img[j] + (v ? hi * p.in_Wa + wi : 0)) * p.in_ldc + cc * 32 + lc * 4;
      ra[j] = v ? *reinterpret_cast<const f32x4*>(p.in + off) : zero4;
    }
#pragma unroll
    for (int j = 0; j < RB; ++j)
      rb[j] = b_ok[j] ? *reinterpret_cast<const f32x4*>(b_ptr[j] + (size_t)c * 32) : zero4;
  };
  auto store_slice = [&](int buf) {
    float* A = lds[buf];
    float* Bm = lds[buf] + BM * LS;
#pragma unroll
    for (int j = 0; j < RA; ++j) *reinterpret_cast<f32x4*>(&A[(lr + 32 * j) * LS + lc * 4]) = ra[j];
#pragma unroll
    for (int j = 0; j < RB; ++j) *reinterpret_cast<f32x4*>(&Bm[(lr + 32 * j) * LS + lc * 4]) = rb[j];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int frag_off = (lane & 31) * LS + (lane >> 5) * 4;
  auto compute = [&](int buf) {
    const float* A = lds[buf] + (wm * TM * 32) * LS + frag_off;
    const float* Bm = lds[buf] + BM * LS + (wn * TN * 32) * LS + frag_off;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const f32x4*>(A + i * 32 * LS + g * 8);
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const f32x4*>(Bm + j * 32 * LS + g * 8);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][t], b[j][t], acc[i][j], 0, 0, 0);
    }
  };

  load_slice(0);
  store_slice(0);
  __syncthreads();
  for (int c = 0; c < nslices; ++c) {
    const bool more = c + 1 < nslices;
    if (more) load_slice(c + 1);
    compute(c & 1);
    if (more) store_slice((c + 1) & 1);
    __syncthreads();
  }

  // ---- epilogue.  D reg r of lane l is C[row (r&3)+8*(r>>2)+4*(l>>5)][col l&31]: stage the
  // block tile through LDS (the A/B stages are dead) so that HBM sees whole 16-byte-per-lane
  // row segments: residual tile prefetched with independent 16-B loads, then bias + residual +
  // ReLU and 16-B stores (a wave covers full 512-byte output rows).
  constexpr int CS = BN + 4;                      // C-tile row stride (floats); +4 keeps b128 reads aligned
  static_assert(BM * CS <= 2 * (BM + BN) * LS, "C tile must fit in the A/B stages");
  float* Ct = &lds[0][0];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int col = wn * TN * 32 + j * 32 + (lane & 31);
        Ct[row * CS + col] = acc[i][j][r];
      }
  __syncthreads();

  constexpr int C4 = BN / 4;                      // 16-byte chunks per tile row
  constexpr int NCH = BM * C4 / 256;              // chunks per thread
  const bool vec_ok = (p.out_ldc & 3) == 0 && (p.res_mode == 0 || (p.res_ldc & 3) == 0);
  f32x4 rv[NCH];
  size_t oaddr[NCH];
  int nval[NCH];
#pragma unroll
  for (int s2 = 0; s2 < NCH; ++s2) {
    const int q = tid + 256 * s2;
    const int row = q / C4, c4 = q - row * C4;
    const int m = m0 + row, col = n0 + c4 * 4;
    int nv = p.Cout - col;
    nv = nv > 4 ? 4 : nv;
    if (m >= M || nv < 0) nv = 0;
    nval[s2] = nv;
    rv[s2] = zero4;
    oaddr[s2] = 0;
    if (nv > 0) {
      const int n = m / HoWo, rr = m - n * HoWo;
      const int ho = rr / p.Wo, wo = rr - ho * p.Wo;
      oaddr[s2] = (((size_t)n * p.out_H + ho + p.out_oy) * p.out_W + wo + p.out_ox) * p.out_ldc + col;
      if (p.res_mode != 0) {
        const size_t rpix = p.res_mode == 1 ? ((size_t)n * p.res_H + ho) * p.res_W + wo
                                            : ((size_t)n * p.res_H + (ho >> 1)) * p.res_W + (wo >> 1);
        const float* rp = p.res + rpix * p.res_ldc + col;
        if (nv == 4 && vec_ok) {
          rv[s2] = *reinterpret_cast<const f32x4*>(rp);
        } else {
          for (int e = 0; e < nv; ++e) rv[s2][e] = rp[e];
        }
      }
    }
  }
#pragma unroll
  for (int s2 = 0; s2 < NCH; ++s2) {
    const int nv = nval[s2];
    if (nv == 0) continue;
    const int q = tid + 256 * s2;
    const int row = q / C4, c4 = q - row * C4;
    const int col = n0 + c4 * 4;
    f32x4 v = *reinterpret_cast<const f32x4*>(&Ct[row * CS + c4 * 4]);
    if (nv == 4) {
      v += *reinterpret_cast<const f32x4*>(p.bias + col);
    } else {
      for (int e = 0; e < nv; ++e) v[e] += p.bias[col + e];
    }
    v += rv[s2];
    if (p.relu) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
    }
    float* op = p.out + oaddr[s2];
    if (nv == 4 && vec_ok) {
      *reinterpret_cast<f32x4*>(op) = v;
    } else {
      for (int e = 0; e < nv; ++e) op[e] = v[e];
    }
  }
}

template <int WM, int WN, int TM, int TN>
void launch_variant(const ConvParams& p, hipStream_t stream) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  const int M = p.B * p.Ho * p.Wo;
  const unsigned grid = (unsigned)(((M + BM - 1) / BM) * ((p.Cout + BN - 1) / BN));
  hipLaunchKernelGGL((conv_igemm_kernel<WM, WN, TM, TN>), dim3(grid), dim3(256), 0, stream, p);
}

}  // namespace

double conv_flops(const ConvParams& p) {
  return 2.0 * (double)p.B * p.Ho * p.Wo * (double)p.Cout * (double)(p.kh * p.kw * p.Cin);
}

int launch_conv(const ConvParams& p, hipStream_t stream) {
  ODT_CHECK(p.Cin % 32 == 0, "conv: Cin must be a multiple of 32");
  ODT_CHECK(p.in_ldc % 4 == 0, "conv: input pixel stride must be a multiple of 4 floats");
  ODT_CHECK(p.B > 0 && p.Ho > 0 && p.Wo > 0 && p.Cout > 0, "conv: empty problem");
  const long M = (long)p.B * p.Ho * p.Wo;
  const long tiles128 = ((M + 127) / 128) * ((p.Cout + 127) / 128);
  int tile = 0;   // 0 auto | 1: 128x64 | 2: 64x64 | 3: 128x128  (ODT_CONV_TILE: tuning / test knob)
  if (const char* e = getenv("ODT_CONV_TILE")) tile = atoi(e);
  if (tile == 0) tile = p.Cout <= 64 ? 1 : (tiles128 < 384 ? 2 : 3);
  if (tile == 1) {
    launch_variant<4, 1, 1, 2>(p, stream);       // 128 x 64
  } else if (tile == 2) {
    launch_variant<2, 2, 1, 1>(p, stream);       // 64 x 64: fill the 256 CUs on small M
  } else {
    launch_variant<2, 2, 2, 2>(p, stream);       // 128 x 128
  }
  ODT_HIP(hipGetLastError());
  return 0;
}

}  // namespace odt
