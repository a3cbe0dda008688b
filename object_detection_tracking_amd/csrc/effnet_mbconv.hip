// MBConv front half in ONE kernel on gfx950: expand 1x1 conv + BN + swish -> depthwise k x k conv + BN + swish (+ the
// squeeze-excite partial sums), without the expanded tensor ever reaching HBM.
//
// Restates reference efficientdet/backbone/efficientnet_model.py:162-330 (MBConvBlock: _expand_conv + _bn0 + swish,
// _depthwise_conv ('same') + _bn1 + swish, the spatial mean of _call_se) exactly as the two launches it replaces do
// (conv_split_kernel<4,1,2> + dwconv_kernel): the block's widest tensor -- six times the block input, written by the expand
// conv and read back (with halo) by the depthwise conv, 2.6 GB of the 33 GB an EfficientDet-D7 frame moved -- stays in LDS.
//
//   * a workgroup owns a 64-channel slice of the expanded tensor and a contiguous range of spatial tiles.  A tile is a
//     16 x 16 patch of INPUT pixels = the 256 rows of one MFMA block tile: the expand conv of the patch is a 256 x 64 x Cin
//     GEMM on the bf16 matrix pipe (x = hi + mid + lo, six exact bf16 products per f32 product: conv_split_common.hpp; the
//     one-stage loop of conv_split1.hip with its weight image), K = the block input's channel stride;
//   * its result (+ folded BN shift, swish; ZERO at patch pixels outside the image: that is the depthwise conv's 'SAME'
//     padding) goes to an LDS tile [256 pixels][64 channels] f32;
//   * the depthwise stencil reads that tile: (16 - k) / s + 1 outputs per side (14 / 12 at stride 1, 7 / 6 at stride 2),
//     thread = (channel quad, one of 16 output slots), taps ky-major / kx inner as in dwconv_kernel (a tap outside the
//     image adds 0 here and is skipped there) but accumulated with fused multiply-adds, + folded BN shift, swish, 16-byte
//     stores, and per-thread sums of what it stores for the squeeze (fixed order: tiles, then slots; a fixed tree over the
//     slots) -> sum_part[b][split][c], which channel_mean_fold_kernel adds up as before.
// The halo is recomputed by the neighbouring tile ((16 / 14)^2 = 1.31x the expand FLOPs at k = 3, 1.78x at k = 5): MFMA
// work the HBM-bound block does not feel.  Deterministic (no atomics).  Built with -ffp-contract=off.
#include <algorithm>

#include "conv_split_common.hpp"

namespace odt {
namespace {

constexpr int kMbP = 16;                         // patch side (input pixels): 16 x 16 = the 256 rows of the block tile
constexpr int kMbBN = 64;                        // expanded channels per workgroup
constexpr int kMbCS = kMbBN + 4;                 // row pitch of the LDS tile (floats): 272 B, 16-byte aligned
// the one-stage loop's LDS image (conv_split1.hip SplitCfg<4,1,2>): 3 A planes + 3 B planes
constexpr int kMbAKG = 256 * 16 + 32, kMbAPL = 4 * kMbAKG, kMbBKG = kMbBN * 16 + 32, kMbBPL = 4 * kMbBKG;
constexpr int kMbLoopLds = 3 * kMbAPL + 3 * kMbBPL;
constexpr int kMbTileLds = 256 * kMbCS * 4;
constexpr int kMbLds = kMbTileLds > kMbLoopLds ? kMbTileLds : kMbLoopLds;
constexpr int kMbStageB = 3 * 4 * kMbBN * 16;    // bytes of pre-imaged weights per K slice of 32
constexpr int kMbNB = kMbStageB / 4096;          // 16-byte weight chunks per thread and slice

// swish on the hardware transcendental units: v * rcp(1 + exp2(-v * log2 e)) -- v_exp_f32 and v_rcp_f32, 1 ulp each, i.e.
// within ~3 ulp of the libm expf + IEEE division form the stand-alone kernels use.  This kernel evaluates 64 + ~50 swishes
// per thread and tile and is bound by exactly that: with the libm form (~60 instructions per value) the first version
// ran no faster than the two launches it replaces (profiles/r05_effdet_mbconv_fusion_v1_ab.txt).  Saturation: v -> +inf
// gives exp2 -> 0, swish -> v; v < -88 gives exp2 -> inf, rcp -> 0, swish -> -0 (the exact value is a denormal there).
__device__ __forceinline__ float mb_swish(float v) {
  return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-v * 1.4426950408889634f));
}

// a * b + c per element with ONE rounding (v_pk_fma_f32): half the stencil's arithmetic instructions; the stand-alone
// dwconv_kernel rounds the product and the sum separately (the build's -ffp-contract=off) -- a difference at the level of the
// last bit of each tap, inside the tolerance the fused handle is held to against the unfused one (2e-5 of the tensor scale)
__device__ __forceinline__ f32x4 mb_fma4(f32x4 a, f32x4 b, f32x4 c) { return __builtin_elementwise_fma(a, b, c); }

template <int K, int S>
__global__ void __launch_bounds__(256, 2) mbconv_expand_dw_kernel(MbExpandDwParams p) {
  constexpr int TO = (kMbP - K) / S + 1;         // outputs per tile side
  constexpr int RA = 8;                          // A rows (16-byte loads) per thread and slice
  // (k = 5: the 25 tap weights of the slice live in LDS behind the tile -- 100 registers otherwise, spilled in the stencil)
  constexpr int WL = K == 5 ? K * K * kMbBN * 4 : 0;
  __shared__ __attribute__((aligned(16))) unsigned char lds[kMbLds + WL];
  unsigned char* const ldsB = lds + 3 * kMbAPL;
  float* const Et = reinterpret_cast<float*>(lds);
  float* const wl = reinterpret_cast<float*>(lds + kMbLds);

  const int tid = threadIdx.x, lane = tid & 63, wm = tid >> 6;
  const int fr = lane & 31, fg = lane >> 5;
  // workgroup -> (channel slice, tile range, image); slices of one tile range are neighbours in the launch order and -- with
  // one contiguous run of the sequence per XCD -- share the input patches in that XCD's L2
  const int ns = p.lmid / kMbBN;
  int wg = (int)blockIdx.x;
  {
    const int nwg = (int)gridDim.x, xcd = wg & 7, q = nwg >> 3, r = nwg & 7;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (wg >> 3);
  }
  const int slice = wg % ns, rest = wg / ns;
  const int split = rest % p.nsplit, b = rest / p.nsplit;
  const int ntiles = p.tiles_y * p.tiles_x;
  const int per = (ntiles + p.nsplit - 1) / p.nsplit;
  const int t_lo = split * per, t_hi = t_lo + per < ntiles ? t_lo + per : ntiles;
  const int nslices = p.in_ldc >> 5;
  const int n0 = slice * kMbBN;

  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(
      (void*)p.x, 0, (int)((unsigned)p.B * p.H * p.W * p.in_ldc * 4u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_wt = __builtin_amdgcn_make_buffer_rsrc(
      (void*)p.w_img, 0, (int)((unsigned)ns * nslices * (unsigned)kMbStageB), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_eb = __builtin_amdgcn_make_buffer_rsrc((void*)p.e_bias, 0, (int)((unsigned)p.mid * 4u), 0x00020000);

  // the expand conv's folded BN shift of this lane's two accumulator columns (past `mid`: zero weights, zero shift)
  float ebias[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
    ebias[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_eb, (n0 + j * 32 + fr) * 4, 0, 0));

  const int lc = tid & 7, lr = tid >> 3;         // A loader: thread -> (row lr + 32 j, 16-byte column lc)
  const int cq = tid & 15, slot = tid >> 4;      // stencil: thread -> (channel quad, output slot)
  const int ch = n0 + cq * 4;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  f32x4 sum = zero;
  const int b_st = (tid / kMbBN) * kMbBKG + (tid % kMbBN) * 16;
  if constexpr (K == 5) {
    for (int i = tid; i < K * K * (kMbBN / 4); i += 256) {
      const int q = i / (kMbBN / 4), c4 = i - q * (kMbBN / 4);
      *reinterpret_cast<f32x4*>(&wl[q * kMbBN + c4 * 4]) = *reinterpret_cast<const f32x4*>(p.dw_wt + (size_t)q * p.lmid + n0 + c4 * 4);
    }
    // (visible to every wave after the first barrier of the first tile's GEMM loop)
  }

  for (int t = t_lo; t < t_hi; ++t) {
    const int ty = t / p.tiles_x, tx = t - ty * p.tiles_x;
    const int iy0 = ty * TO * S - p.pad_t, ix0 = tx * TO * S - p.pad_l;
    // ---- expand conv of the patch: 256 x 64 x in_ldc, the one-stage loop of conv_split_kernel<4,1,2>
    unsigned a_row[RA];
#pragma unroll
    for (int j = 0; j < RA; ++j) {
      const int m = lr + 32 * j;
      const int y = iy0 + (m >> 4), x = ix0 + (m & 15);
      const bool v = (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
      a_row[j] = v ? (((unsigned)b * p.H + (unsigned)y) * p.W + (unsigned)x) * (unsigned)p.in_ldc * 4u + lc * 16u : kOOB;
    }
    unsigned l_b = (unsigned)slice * (unsigned)nslices * (unsigned)kMbStageB;
    int l_cc = 0;
    f32x4 ga[RA];
    u32x4 gb[kMbNB];
    auto load_slice = [&]() {
#pragma unroll
      for (int j = 0; j < RA; ++j) ga[j] = (f32x4)__builtin_amdgcn_raw_buffer_load_b128(rs_in, (int)a_row[j], l_cc * 128, 0);
#pragma unroll
      for (int i = 0; i < kMbNB; ++i) gb[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_wt, tid * 16 + i * 4096, (int)l_b, 0);
      l_b += (unsigned)kMbStageB;
      ++l_cc;
    };
    auto store_slice = [&]() {
#pragma unroll
      for (int j = 0; j < RA; ++j) {
        unsigned h0, m0_, l0, h1, m1, l1;
        split2(ga[j][0], ga[j][1], h0, m0_, l0);
        split2(ga[j][2], ga[j][3], h1, m1, l1);
        const int off = (lc >> 1) * kMbAKG + (lr + 32 * j) * 16 + (lc & 1) * 8;
        *reinterpret_cast<u32x2*>(lds + 0 * kMbAPL + off) = u32x2{h0, h1};
        *reinterpret_cast<u32x2*>(lds + 1 * kMbAPL + off) = u32x2{m0_, m1};
        *reinterpret_cast<u32x2*>(lds + 2 * kMbAPL + off) = u32x2{l0, l1};
      }
#pragma unroll
      for (int i = 0; i < kMbNB; ++i) {        // chunk tid + 256 i of the stage image [piece][k-group][n]
        constexpr int perp = 4 * kMbBN / 256;  // chunks-of-256 per piece
        *reinterpret_cast<u32x4*>(ldsB + (i / perp) * kMbBPL + (((i % perp) * 256) / kMbBN) * kMbBKG + b_st) = gb[i];
      }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    load_slice();
    for (int c = 0; c < nslices; ++c) {
      store_slice();
      __syncthreads();
      if (c + 1 < nslices) load_slice();
      {
        bf16x8 fa[3][2], fb[3];
        auto rdA = [&](int q, int ks) {
#pragma unroll
          for (int u = 0; u < 2; ++u)
            fa[q][u] = *reinterpret_cast<const bf16x8*>(lds + q * kMbAPL + (ks * 2 + fg) * kMbAKG + (wm * 64 + u * 32 + fr) * 16);
        };
        auto rdB = [&](int q, int ks, int j) {
          fb[q] = *reinterpret_cast<const bf16x8*>(ldsB + q * kMbBPL + (ks * 2 + fg) * kMbBKG + (j * 32 + fr) * 16);
        };
#define ODT_MF(qa, qb, j) { acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[qa][0], fb[qb], acc[0][j], 0, 0, 0); \
                            acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[qa][1], fb[qb], acc[1][j], 0, 0, 0); }
#define ODT_FENCE() __builtin_amdgcn_sched_barrier(0)
#pragma unroll
        for (int q = 0; q < 3; ++q) rdA(q, 0);
#pragma unroll
        for (int q = 0; q < 3; ++q) rdB(q, 0, 0);
        ODT_FENCE();
#pragma unroll
        for (int g = 0; g < 4; ++g) {            // two k16 steps x two 32-column groups (the order of conv_split_kernel)
          const int j = g % 2;
          const int nks = (g + 1) / 2, nj = (g + 1) % 2;
          const bool has_next = g < 3, a_next = has_next && nj == 0;
          ODT_MF(2, 0, j); ODT_FENCE();          // lo * hi
          if (a_next) rdA(2, nks);
          ODT_FENCE();
          ODT_MF(1, 0, j); ODT_MF(0, 0, j); ODT_FENCE();   // mid * hi, hi * hi
          if (has_next) rdB(0, nks, nj);
          ODT_FENCE();
          ODT_MF(1, 1, j); ODT_FENCE();          // mid * mid
          if (a_next) rdA(1, nks);
          ODT_FENCE();
          ODT_MF(0, 1, j); ODT_FENCE();          // hi * mid
          if (has_next) rdB(1, nks, nj);
          ODT_FENCE();
          ODT_MF(0, 2, j); ODT_FENCE();          // hi * lo
          if (a_next) rdA(0, nks);
          if (has_next) rdB(2, nks, nj);
          ODT_FENCE();
        }
#undef ODT_MF
#undef ODT_FENCE
      }
      __syncthreads();
    }
    // ---- BN shift + swish; pixels of the patch outside the image are the depthwise conv's zero padding
    // (the lane's row / column base passes through an opaque asm: otherwise its 64 row numbers and LDS addresses are
    // computed once in front of the tile loop, spilled, and reloaded per tile)
    int rb = wm * 64 + 4 * fg, cb = fr;
    ODT_PIN2(rb, cb);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rb + i * 32 + (r & 3) + 8 * (r >> 2);
        const int y = iy0 + (row >> 4), x = ix0 + (row & 15);
        const bool v = (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const float e = mb_swish(acc[i][j][r] + ebias[j]);
          Et[row * kMbCS + j * 32 + cb] = v ? e : 0.f;
        }
      }
    __syncthreads();
    // ---- depthwise stencil over the tile: this thread's outputs q = slot, slot + 16, ...; taps ky-major, kx inner
    {
      asm volatile("" ::: "memory");             // (keeps the weight loads below out of the GEMM loop's register budget)
      f32x4 wreg[K == 3 ? K * K : 1];
      if constexpr (K == 3) {
#pragma unroll
        for (int q = 0; q < K * K; ++q) wreg[q] = *reinterpret_cast<const f32x4*>(p.dw_wt + (size_t)q * p.lmid + ch);
      }
      auto w_at = [&](int q) -> f32x4 {
        if constexpr (K == 3) return wreg[q];
        else return *reinterpret_cast<const f32x4*>(&wl[q * kMbBN + cq * 4]);
      };
      const f32x4 dbias = *reinterpret_cast<const f32x4*>(p.dw_bias + ch);
      auto finish = [&](f32x4 a, int oy, int ox) {
        f32x4 v = a + dbias;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = mb_swish(v[e]);
        *reinterpret_cast<f32x4*>(p.out + (((size_t)b * p.Ho + oy) * p.Wo + ox) * p.lmid + ch) = v;
        sum += v;
      };
      if constexpr (S == 1) {
        // stride 1: TWO horizontally adjacent outputs per iteration share K + 1 of their 2 K reads per kernel row (TO is even)
        constexpr int PR = TO / 2;
#pragma unroll 1
        for (int q = slot; q < TO * PR; q += 16) {
          const int oyl = q / PR, oxl = (q - oyl * PR) * 2;
          const int oy = ty * TO + oyl, ox = tx * TO + oxl;
          if (oy >= p.Ho || ox >= p.Wo) continue;
          if constexpr (K == 5) asm volatile("" ::: "memory");      // (the LDS tap weights are re-read per iteration, not hoisted into 100 registers)
          const float* e0 = &Et[(oyl * kMbP + oxl) * kMbCS + cq * 4];
          f32x4 a0 = zero, a1 = zero;
#pragma unroll
          for (int ky = 0; ky < K; ++ky) {
            f32x4 ev[K + 1];
#pragma unroll
            for (int kx = 0; kx <= K; ++kx) ev[kx] = *reinterpret_cast<const f32x4*>(e0 + (ky * kMbP + kx) * kMbCS);
            __builtin_amdgcn_sched_barrier(0);   // (a kernel row's reads issued together, then its products: ky-major, kx inner)
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
              const f32x4 wt = w_at(ky * K + kx);
              a0 = mb_fma4(ev[kx], wt, a0);
              a1 = mb_fma4(ev[kx + 1], wt, a1);
            }
          }
          finish(a0, oy, ox);
          if (ox + 1 < p.Wo) finish(a1, oy, ox + 1);
        }
      } else {
#pragma unroll 1
        for (int q = slot; q < TO * TO; q += 16) {
          const int oyl = q / TO, oxl = q - oyl * TO;
          const int oy = ty * TO + oyl, ox = tx * TO + oxl;
          if (oy >= p.Ho || ox >= p.Wo) continue;
          const float* e0 = &Et[((oyl * S) * kMbP + oxl * S) * kMbCS + cq * 4];
          f32x4 a = zero;
#pragma unroll
          for (int ky = 0; ky < K; ++ky) {
            f32x4 ev[K];
#pragma unroll
            for (int kx = 0; kx < K; ++kx) ev[kx] = *reinterpret_cast<const f32x4*>(e0 + (ky * kMbP + kx) * kMbCS);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kx = 0; kx < K; ++kx) a = mb_fma4(ev[kx], w_at(ky * K + kx), a);
          }
          finish(a, oy, ox);
        }
      }
    }
    __syncthreads();                             // the tile aliases the next patch's operand planes
  }
  if (p.sum_part == nullptr) return;
  f32x4* red = reinterpret_cast<f32x4*>(lds);
  red[tid] = sum;
  __syncthreads();
  if (slot == 0) {
    f32x4 s = red[cq];
#pragma unroll
    for (int g = 1; g < 16; ++g) s += red[g * 16 + cq];               // fixed order
    *reinterpret_cast<f32x4*>(p.sum_part + ((size_t)b * p.nsplit + split) * p.lmid + ch) = s;
  }
}

}  // namespace

static int mb_tile_out(int k, int stride) { return (kMbP - k) / stride + 1; }

// tile ranges per image: about 2048 workgroups in all (two resident per CU, a few rounds), never more ranges than tiles,
// at most 1024 partial sums per channel for the fold kernel
int mbconv_expand_dw_splits(const MbExpandDwParams& p) {
  const int to = mb_tile_out(p.k, p.stride);
  const long ntiles = (long)((p.Ho + to - 1) / to) * ((p.Wo + to - 1) / to);
  const long ns = p.lmid / kMbBN;
  const long want = std::max<long>(1, 2048 / std::max<long>(1, ns * p.B));
  return (int)std::max<long>(1, std::min<long>(std::min<long>(want, 1024), ntiles));
}

size_t mbconv_expand_weight_bytes(int lmid, int in_ldc) { return (size_t)lmid * in_ldc * 6; }

int launch_mbconv_expand_dw(const MbExpandDwParams& p0, hipStream_t stream) {
  MbExpandDwParams p = p0;
  ODT_CHECK((p.k == 3 || p.k == 5) && (p.stride == 1 || p.stride == 2), "mbconv_expand_dw: kernel 3 / 5, stride 1 / 2");
  ODT_CHECK(p.lmid % kMbBN == 0 && p.in_ldc % 32 == 0 && p.mid <= p.lmid && p.mid > 0, "mbconv_expand_dw: channel strides must be multiples of 64 / 32");
  ODT_CHECK((double)p.B * p.H * p.W * p.in_ldc * 4.0 < 2147483648.0 && (double)p.lmid * p.in_ldc * 6.0 < 2147483648.0,
            "mbconv_expand_dw: tensor above 2 GiB");
  const int to = mb_tile_out(p.k, p.stride);
  p.tiles_y = (p.Ho + to - 1) / to; p.tiles_x = (p.Wo + to - 1) / to;
  if (p.nsplit <= 0) p.nsplit = mbconv_expand_dw_splits(p);
  ODT_CHECK(p.nsplit <= p.tiles_y * p.tiles_x, "mbconv_expand_dw: more tile ranges than tiles");
  const unsigned grid = (unsigned)((p.lmid / kMbBN) * p.nsplit * p.B);
  if (p.k == 3 && p.stride == 1) hipLaunchKernelGGL((mbconv_expand_dw_kernel<3, 1>), dim3(grid), dim3(256), 0, stream, p);
  else if (p.k == 3) hipLaunchKernelGGL((mbconv_expand_dw_kernel<3, 2>), dim3(grid), dim3(256), 0, stream, p);
  else if (p.stride == 1) hipLaunchKernelGGL((mbconv_expand_dw_kernel<5, 1>), dim3(grid), dim3(256), 0, stream, p);
  else hipLaunchKernelGGL((mbconv_expand_dw_kernel<5, 2>), dim3(grid), dim3(256), 0, stream, p);
  ODT_HIP(hipGetLastError());
  return 0;
}

}  // namespace odt
