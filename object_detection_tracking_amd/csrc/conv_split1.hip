// conv_split_kernel: the one-stage loop of the bf16x3 split convolution (round 1) -- 4 waves, BK = 32, one LDS stage
// (3 A planes + 3 B planes) + register prefetch of the next slice, two barriers per slice, two workgroups per CU.  The plans
// use its 256 x 64 tile for the 64-wide layers (conv0, the 1x1 convs of res2, EfficientNet's narrow projections), where
// the 8-wave kernels' 64 x 32 wave tile reads too many fragments per MFMA; conv_split_family = 1 puts every split layer on it.
// A: f32 NHWC activations, gathered per tap exactly as in conv_igemm.hip, split on the way into LDS.  B: the pre-split
// weight image [n-tile][k-slice][piece][k-group][BN n][8 k] (conv_make_split_weights), one linear copy per stage.
// LDS planes are [k-group][row][8 bf16]: a wave's ds_read_b128 of an MFMA operand is one contiguous 512-byte run per 32
// lanes.  Residuals (same shape / nearest-2x) become the accumulators' start value; optional K-concatenated second A source.
#include "conv_split_epilogue.hpp"

namespace odt {

namespace {

// Tile configurations <WM, WN, TN>: 4 waves as WM x WN, wave tile 64 x (32*TN):
//   <2,2,4> 128 x 256  (Cout % 256 == 0)     <4,1,4> 256 x 128  (Cout % 128 == 0: res3)
//   <4,1,2> 256 x 64   (Cout % 64 == 0: res2)
template <int WM, int WN, int TN>
struct SplitCfg {
  static constexpr int BM = WM * 64, BN = WN * TN * 32;
  static constexpr int AKG = BM * 16 + 32, APL = 4 * AKG;     // bytes: one k-group / one piece plane of A
  static constexpr int BKG = BN * 16 + 32, BPL = 4 * BKG;
  static constexpr int LDS = 3 * APL + 3 * BPL;               // 74,496 B (62,208 for 256 x 64)
  static constexpr int STAGE_B = 3 * 4 * BN * 16;             // bytes of pre-imaged weights per stage
  static constexpr int RA = BM / 32;                          // A rows (16-B loads) per thread and slice
  static constexpr int NB = STAGE_B / 4096;                   // B 16-B chunks per thread and slice
};

// TRACE: tuning builds only (ODT_CONV_TRACE through odt_op_conv2d): wall-clock stamps per workgroup in
// the slots of conv_igemm.hip (0 start, 6 first loads issued, 7 first stage stored, 1 main loop, 2 epilogue,
// 3/4 first pass staged / stored, 5 end; 8/9 HW_ID / XCC_ID).  Compiled out of the production kernels.
template <int WM, int WN, int TN, bool TRACE = false>
__global__ void __launch_bounds__(256, 2) conv_split_kernel(const ConvParams* __restrict__ pp) {
  using Cfg = SplitCfg<WM, WN, TN>;
  constexpr int SBM = Cfg::BM, SBN = Cfg::BN, AKG = Cfg::AKG, APL = Cfg::APL, BKG = Cfg::BKG, BPL = Cfg::BPL;
  constexpr int LDS_SPLIT = Cfg::LDS, STAGE_B_BYTES = Cfg::STAGE_B, RA = Cfg::RA, NB = Cfg::NB;
  static_assert(WM * WN == 4, "4 waves");
  const ConvParams p = *pp;
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_SPLIT];
  unsigned char* const ldsB = lds + 3 * APL;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  auto stamp = [&](int i) {
    if constexpr (TRACE) {
      if (tid == 0) p.trace[(size_t)blockIdx.x * 16 + i] = wall_clock64();
    }
  };
  stamp(0);
  if constexpr (TRACE) {
    if (tid == 0) {
      p.trace[(size_t)blockIdx.x * 16 + 8] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
      p.trace[(size_t)blockIdx.x * 16 + 9] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    }
  }
  const int ntn = cout_padded(p.Cout) / SBN;
  // XCD-aware tile order (see conv_igemm.hip): one contiguous run of tiles per XCD
  int wg = (int)blockIdx.x;
  {
    const int nwg = (int)gridDim.x, xcd = wg & 7, q = nwg >> 3, r = nwg & 7;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (wg >> 3);
  }
  const int mt = wg / ntn, nt = wg - mt * ntn;
  const int m0 = mt * SBM, n0 = nt * SBN;
  const int HoWo = p.Ho * p.Wo;
  const int M = p.B * HoWo;
  const int cpt = p.Cin >> 5;
  const int cpt2 = p.in2 != nullptr ? p.Cin2 >> 5 : 0;    // slices of the second A source (1x1 only)
  const int nslices = p.kh * p.kw * cpt + cpt2;

  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(
      (void*)p.in, 0, (int)((unsigned)p.B * p.in_Ha * p.in_Wa * p.in_ldc * 4u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_in2 = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.in2 != nullptr ? p.in2 : p.in), 0,
      (int)(p.in2 != nullptr ? (unsigned)p.B * p.in2_Ha * p.in2_Wa * p.in2_ldc * 4u : 0u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_wt = __builtin_amdgcn_make_buffer_rsrc(
      (void*)p.wt_split, 0, (int)((unsigned)ntn * nslices * (unsigned)STAGE_B_BYTES), 0x00020000);

  // ---- A loader: thread -> (row lr + 32*j, 16-byte column lc), as in conv_igemm.hip
  const int lc = tid & 7, lr = tid >> 3;
  int a_hw0[RA];
  unsigned a_img[RA];
  const bool dense_in = p.kh == 1 && p.kw == 1 && p.stride == 1 && p.pad_t == 0 && p.pad_l == 0 &&
                        p.H == p.in_Ha && p.W == p.in_Wa && p.Ho == p.H && p.Wo == p.W;
#pragma unroll
  for (int j = 0; j < RA; ++j) {
    const int m = m0 + lr + 32 * j;
    const bool ok = m < M;
    if (dense_in) {
      a_hw0[j] = 0;
      a_img[j] = ok ? (unsigned)m * p.in_ldc * 4u + lc * 16u : kOOB;
    } else {
      const int mm = ok ? m : 0;
      const int n = sfast_div(mm, p.div_howo_mul, p.div_howo_sh), r = mm - n * HoWo;
      const int ho = sfast_div(r, p.div_wo_mul, p.div_wo_sh), wo = r - ho * p.Wo;
      a_hw0[j] = (int)(((unsigned)(ho * p.stride - p.pad_t) << 16) | ((unsigned)(wo * p.stride - p.pad_l) & 0xffffu));
      a_img[j] = ok ? (unsigned)n * p.in_Ha * p.in_Wa * p.in_ldc * 4u + lc * 16u : kOOB;
    }
  }
  const unsigned pix_bytes = (unsigned)p.in_ldc * 4u;
  int l_cc = 0, l_kh = 0, l_kw = 0;
  unsigned a_row[RA];
  auto set_tap = [&](int khh, int kww) {
#pragma unroll
    for (int j = 0; j < RA; ++j) {
      const int hi = (a_hw0[j] >> 16) + khh * p.dil, wi = (int)(short)(a_hw0[j] & 0xffff) + kww * p.dil;
      const bool v = (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W && a_img[j] != kOOB;
      a_row[j] = v ? a_img[j] + (unsigned)(hi * p.in_Wa + wi) * pix_bytes : kOOB;
    }
  };
  set_tap(0, 0);
  // second source (stage-entry bottleneck: conv3(t2) + convshortcut(x) as one K-concatenated GEMM):
  // output row m reads pixel (n, ho * in2_stride, wo * in2_stride) of in2
  bool l_src2 = false;
  int l_cpt = cpt;
  auto set_src2 = [&]() {
#pragma unroll
    for (int j = 0; j < RA; ++j) {
      const int m = m0 + lr + 32 * j;
      const bool ok = m < M;
      const int mm = ok ? m : 0;
      const int n = sfast_div(mm, p.div_howo_mul, p.div_howo_sh), r = mm - n * HoWo;
      const int ho = sfast_div(r, p.div_wo_mul, p.div_wo_sh), wo = r - ho * p.Wo;
      const unsigned pix = ((unsigned)n * p.in2_Ha + (unsigned)(ho * p.in2_stride)) * p.in2_Wa + (unsigned)(wo * p.in2_stride);
      a_row[j] = ok ? pix * (unsigned)p.in2_ldc * 4u + lc * 16u : kOOB;
    }
  };
  unsigned l_b = (unsigned)nt * (unsigned)nslices * (unsigned)STAGE_B_BYTES;   // weight-image offset of the load stream

  f32x4 ga[RA];
  u32x4 gb[NB];
  // a 1x1 layer with one n-tile reads every activation byte once, by one workgroup: non-temporal hint (ConvParams::debug 0x800,
  // set by the plan; see conv_h2.hip)
  const bool a_nt = (p.debug & 0x800) != 0 && ntn == 1 && p.kh * p.kw == 1;
  const int b_st = (tid / SBN) * BKG + (tid % SBN) * 16;   // this thread's place inside a 256-chunk run
  auto load_slice = [&]() {
#pragma unroll
    for (int j = 0; j < RA; ++j)
      ga[j] = a_nt ? (f32x4)__builtin_amdgcn_raw_buffer_load_b128(l_src2 ? rs_in2 : rs_in, (int)a_row[j], l_cc * 128, 2)
                   : (f32x4)__builtin_amdgcn_raw_buffer_load_b128(l_src2 ? rs_in2 : rs_in, (int)a_row[j], l_cc * 128, 0);
#pragma unroll
    for (int i = 0; i < NB; ++i)
      gb[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_wt, tid * 16 + i * 4096, (int)l_b, 0);
    l_b += (unsigned)STAGE_B_BYTES;
    if (++l_cc == l_cpt) {
      l_cc = 0;
      if (!l_src2) {
        if (++l_kw == p.kw) { l_kw = 0; ++l_kh; }
        if (l_kh == p.kh && cpt2 > 0) {
          l_src2 = true; l_cpt = cpt2;
          set_src2();
        } else {
          set_tap(l_kh, l_kw);      // harmless past the last tap (never loaded)
        }
      }
    }
  };
  auto store_slice = [&]() {
#pragma unroll
    for (int j = 0; j < RA; ++j) {
      unsigned h0, m0_, l0, h1, m1, l1;
      split2(ga[j][0], ga[j][1], h0, m0_, l0);
      split2(ga[j][2], ga[j][3], h1, m1, l1);
      const int off = (lc >> 1) * AKG + (lr + 32 * j) * 16 + (lc & 1) * 8;
      *reinterpret_cast<u32x2*>(lds + 0 * APL + off) = u32x2{h0, h1};
      *reinterpret_cast<u32x2*>(lds + 1 * APL + off) = u32x2{m0_, m1};
      *reinterpret_cast<u32x2*>(lds + 2 * APL + off) = u32x2{l0, l1};
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {        // chunk tid + 256 i of the stage image [piece][k-group][n]
      constexpr int per = 4 * SBN / 256;  // chunks-of-256 per piece
      *reinterpret_cast<u32x4*>(ldsB + (i / per) * BPL + (((i % per) * 256) / SBN) * BKG + b_st) = gb[i];
    }
  };

  f32x16 acc[2][TN];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int fr = lane & 31, fg = lane >> 5;
  load_slice();
  stamp(6);
  if (p.res_mode != 0) {
    // residual of the same shape (bottleneck conv3) or the nearest-2x upsampled coarser level (FPN
    // lateral, res_mode 2): the accumulators START at the residual, read in the MFMA C layout (a
    // 32-lane group covers one 128-byte row segment) while the first slice is in flight -- no
    // residual traffic in the epilogue.
    const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.res, 0, (int)((unsigned)p.B * p.res_H * p.res_W * p.res_ldc * 4u), 0x00020000);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fg;
        unsigned rpix = (unsigned)row;
        if (p.res_mode == 2) {
          const int mm = row < M ? row : 0;
          const int n = sfast_div(mm, p.div_howo_mul, p.div_howo_sh), rr = mm - n * HoWo;
          const int ho = sfast_div(rr, p.div_wo_mul, p.div_wo_sh), wo = rr - ho * p.Wo;
          rpix = ((unsigned)n * p.res_H + (unsigned)(ho >> 1)) * p.res_W + (unsigned)(wo >> 1);
        }
        const unsigned roff = row < M ? rpix * (unsigned)p.res_ldc * 4u + (unsigned)(n0 + wn * TN * 32 + fr) * 4u : kOOB;
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_res, (int)roff, j * 128, 0));
      }
  }
  for (int c = 0; c < nslices; ++c) {
    store_slice();
    __syncthreads();
    if constexpr (TRACE) { if (c == 0) { stamp(7); stamp(1); } }
    if (c + 1 < nslices) load_slice();
    {
      // Two k16 steps x TN 32-column groups.  Within a group the b0 (hi) products run first,
      // then b1, then b2; each piece's fragment of the NEXT group is re-read right after its last
      // use, behind the remaining MFMAs of this group (sched_barrier fences pin the order: left
      // alone the scheduler issues a group's three reads and waits for them in front of its MFMAs).
      bf16x8 fa[3][2], fb[3];
      auto rdA = [&](int q, int ks) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
          fa[q][t] = *reinterpret_cast<const bf16x8*>(lds + q * APL + (ks * 2 + fg) * AKG + (wm * 64 + t * 32 + fr) * 16);
      };
      auto rdB = [&](int q, int ks, int j) {
        fb[q] = *reinterpret_cast<const bf16x8*>(ldsB + q * BPL + (ks * 2 + fg) * BKG + (wn * TN * 32 + j * 32 + fr) * 16);
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
      for (int g = 0; g < 2 * TN; ++g) {
        const int j = g % TN;
        const int nks = (g + 1) / TN, nj = (g + 1) % TN;
        const bool has_next = g < 2 * TN - 1, a_next = has_next && nj == 0;
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

  stamp(2);
  // ---- epilogue (the fast path of conv_igemm.hip without a residual): stage the tile through
  // LDS in two passes of RP rows, bias + activation, whole 16-byte-per-lane row segments.
  constexpr int CS = SBN + 4;
  constexpr int RP = SBM / 2, WPP = RP / 64;               // rows / wave-rows per pass
  constexpr int C4 = SBN / 4, RSTEP = 256 / C4;            // 16-byte chunks per row; rows per sweep of the block
  constexpr int NCH = RP / RSTEP;                          // chunks per thread and pass
  static_assert(RP * CS * 4 <= LDS_SPLIT, "C tile pass must fit");
  float* Ct = reinterpret_cast<float*>(lds);
  const bool dense_io = p.out_oy == 0 && p.out_ox == 0 && p.out_H == p.Ho && p.out_W == p.Wo;
  const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(
      (void*)p.out, 0, (int)((unsigned)p.B * p.out_H * p.out_W * p.out_ldc * 4u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_bias =
      __builtin_amdgcn_make_buffer_rsrc((void*)p.bias, 0, (int)((unsigned)p.Cout * 4u), 0x00020000);
  const int c4 = tid % C4, row0 = tid / C4;
  const int col = n0 + c4 * 4;
  const f32x4 bias4 = (f32x4)__builtin_amdgcn_raw_buffer_load_b128(rs_bias, col * 4, 0, 0);
  float vmax = 0.f;                          // |max| of what this thread stores (ConvParams::out_amax)
  auto run = [&](auto act_c) {
    constexpr int ACT = decltype(act_c)::value;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      if (pass > 0) __syncthreads();
      if (wm / WPP == pass) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
              Ct[((wm % WPP) * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fg) * CS + wn * TN * 32 + j * 32 + fr] = acc[i][j][r];
      }
      __syncthreads();
      if (pass == 0) stamp(3);
#pragma unroll
      for (int s2 = 0; s2 < NCH; ++s2) {
        const int rl = row0 + s2 * RSTEP;
        const int m = m0 + pass * RP + rl;
        const bool ok = m < M;
        unsigned opix;
        if (dense_io) {
          opix = (unsigned)m;
        } else {
          const int mm = ok ? m : 0;
          const int n = sfast_div(mm, p.div_howo_mul, p.div_howo_sh), rr = mm - n * HoWo;
          const int ho = sfast_div(rr, p.div_wo_mul, p.div_wo_sh), wo = rr - ho * p.Wo;
          opix = ((unsigned)n * p.out_H + ho + p.out_oy) * p.out_W + wo + p.out_ox;
        }
        const unsigned ooff = ok ? (opix * p.out_ldc + col) * 4u : kOOB;
        f32x4 v = *reinterpret_cast<const f32x4*>(&Ct[rl * CS + c4 * 4]);
        v += bias4;
        if (ACT == 1) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        } else if (ACT == 2) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = v[e] * (1.0f / (1.0f + expf(-v[e])));
        } else if (ACT == 3) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = 1.0f / (1.0f + expf(-v[e]));
        }
        __builtin_amdgcn_raw_buffer_store_b128((u32x4)v, rs_out, (int)ooff, 0, 0);
        if (ok) vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
      }
      if (pass == 0) stamp(4);
    }
  };
  if (p.relu == 1) run(std::integral_constant<int, 1>{});
  else if (p.relu == 2) run(std::integral_constant<int, 2>{});
  else if (p.relu == 3) run(std::integral_constant<int, 3>{});
  else run(std::integral_constant<int, 0>{});
  publish_amax(p.out_amax, vmax, tid);
  stamp(5);
}

}  // namespace

int launch_conv_split1(const ConvParams& p, const ConvParams* dev, hipStream_t stream) {
  const long M = (long)p.B * p.Ho * p.Wo;
  const int bn = p.wt_split_bn != 0 ? p.wt_split_bn : conv_split_bn(p.Cout);
  const int bm = conv_split_bm(p.Cout);
  const unsigned grid = (unsigned)(((M + bm - 1) / bm) * (cout_padded(p.Cout) / bn));
  if (p.trace != nullptr) {        // tuning: the stamped instantiations
    if (bn == 256) hipLaunchKernelGGL((conv_split_kernel<2, 2, 4, true>), dim3(grid), dim3(256), 0, stream, dev);
    else if (bn == 128) hipLaunchKernelGGL((conv_split_kernel<4, 1, 4, true>), dim3(grid), dim3(256), 0, stream, dev);
    else hipLaunchKernelGGL((conv_split_kernel<4, 1, 2, true>), dim3(grid), dim3(256), 0, stream, dev);
  } else if (bn == 256) hipLaunchKernelGGL((conv_split_kernel<2, 2, 4>), dim3(grid), dim3(256), 0, stream, dev);
  else if (bn == 128) hipLaunchKernelGGL((conv_split_kernel<4, 1, 4>), dim3(grid), dim3(256), 0, stream, dev);
  else hipLaunchKernelGGL((conv_split_kernel<4, 1, 2>), dim3(grid), dim3(256), 0, stream, dev);
  ODT_HIP(hipGetLastError());
  return 0;
}

}  // namespace odt
