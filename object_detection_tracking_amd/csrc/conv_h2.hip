// ---------------------------------------------------------------------------------------------------------
// fp16x2 split convolution: f32 implicit GEMM through THREE exact f16 MFMA products per MAC (conv_split_common.hpp has the
// arithmetic: x 2^s = hi + lo to 2^-22, hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16, f32 accumulation) -- half the
// matrix-pipe work of the bf16x3 kernels of conv_split3.hip, whose loop structure these kernels keep:
//   * 8 waves / 512 threads, ONE workgroup per CU, 256-row tiles, wave tile 64 x 32 TN, <TN = 4> 256 x 256, <TN = 2> 256 x 128
//     (and 4-wave 128 x 128 / 128 x 64 tiles for the layers with few tiles / 64 output channels: several workgroups per CU);
//   * BK = 32 per LDS stage (two MFMA k-steps: the stage carries the MFMA time of a bf16x3 BK = 16 stage), TWO stages
//     of A and of B, ONE barrier per stage in front of the stage's last column group;
//   * weights: the pre-split, pre-scaled image goes global -> LDS by LDS-DMA (buffer_load ... lds) into the buffer the
//     barrier has just released, one stage ahead (counted vmcnt, raw s_barrier);
//   * activations: f32 -> registers (two stages ahead) -> x 2^s -> hi / lo -> LDS behind the first column groups;
//   * the scale 2^s comes from the |max| the producer recorded for the source tensor(s) (ConvParams::in_amax); the
//     epilogue (conv_split_epilogue.hpp) multiplies by 2^-s and the weight column's 2^-t_n in front of the bias and
//     records the |max| of what it stores for the next layer.
// conv_h2_kernel (this file): any taps / stride / dilation, optional K-concatenated second source (1x1), split-K.
// conv_h2k_kernel (conv_h2k.hip): stride-1 KH x 3 convs over rows of the output's pitch -- the three kw taps of a (32-channel slice, kh)
// group read ONE staged, once-split run of input pixels (conv_split3k_kernel's scheme).
// K order: (32-channel slice, kh, kw), then the second source's slices.  Reference ops: as conv_split.hip.
#include "conv_split_epilogue.hpp"

namespace odt {

namespace {

#define ODT_MFMA_F16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)

template <int TN, int WM, int WN = 2>
struct H2Cfg {
  static constexpr int BM = 64 * WM, BN = 32 * TN * WN;
  static constexpr int NWV = WN * WM, NTHR = 64 * NWV;                      // waves (WM x WN), threads
  // A stage: [piece 2][k-group 4][row][8 f16]; 32-B pad per k-group: the four k-groups a store instruction touches (8 lanes
  // per row, two rows per 16-lane group) start 8 banks apart (ds_write_b64 banks are mod 32; a 64-B pad put k-groups 0 / 2 and
  // 1 / 3 on the same banks: 18 % of the LDS cycles were conflict cycles, profiles/r03_pmc_lds_wait_by_kernel.txt)
  static constexpr int AKG = BM * 16 + 32, APL = 4 * AKG, ASTG = 2 * APL;
  static constexpr int BKG = BN * 16, BPL = 4 * BKG, STAGE_B = 2 * BPL;     // B stage: the linear image the DMA writes
  static constexpr int BOFF = 2 * ASTG;
  static constexpr int RING = BOFF + 2 * STAGE_B;
  static constexpr int CTILE = (BM < 128 ? BM : 128) * (BN + 4) * 4;        // 128-row epilogue passes
  static constexpr int LDS = RING > CTILE ? RING : CTILE;
  static constexpr int NW = STAGE_B / 1024 / NWV;                           // DMA instructions per wave and stage
  static constexpr int RA = BM * 8 / NTHR;                                  // A rows (16-byte loads) per thread and stage: BM / (NTHR / 8)
  static_assert(LDS <= 160 * 1024 && STAGE_B % (1024 * NWV) == 0, "LDS ring");
};

// <TN, WM>: <4, 4> 256 x 256, <2, 4> 256 x 128 (8 waves, one workgroup per CU); <2, 2> 128 x 128 with 4 waves and 66 KB of LDS --
// two workgroups per CU -- for the layers with too few 256-row tiles (res5, P5; with split-K below res3 at b=1; as an A/B knob
// also for short 1x1 reductions: measured no gain); <1, 2> 128 x 64 on 4 waves for the 64-wide layers -- one column group per
// k-step, 12 MFMAs per wave and stage, 50 KB of LDS: three workgroups per CU
// <2, 8, ., 1>: 512 x 64 on 8 waves stacked along M (a 64 x 64 wave tile: 24 MFMAs per stage) for the 64-wide layers with many rows
template <int TN, int WM, bool TRACE = false, int WN = 2>
__global__ void __launch_bounds__(64 * WN * WM, 2) conv_h2_kernel(const ConvParams* __restrict__ pp) {
  using G = H2Cfg<TN, WM, WN>;
  constexpr int NWV = G::NWV, AR = G::NTHR / 8;
  constexpr int BM = G::BM, BN = G::BN, AKG = G::AKG, APL = G::APL, ASTG = G::ASTG, BKG = G::BKG, BPL = G::BPL;
  constexpr int STAGE_B = G::STAGE_B, BOFF = G::BOFF, NW = G::NW, RA = G::RA;
  const ConvParams p = *pp;
  __shared__ __attribute__((aligned(16))) unsigned char lds[G::LDS];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
  const int ntn = cout_padded(p.Cout) / BN;
  int wg = (int)blockIdx.x;
  {
    const int nwg = (int)gridDim.x, xcd = wg & 7, q = nwg >> 3, r = nwg & 7;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (wg >> 3);
  }
  // split-K: consecutive workgroups are the K ranges of one tile (they share its activation rows in L2)
  const int splitk = p.splitk > 1 ? p.splitk : 1;
  const int ks = wg % splitk;
  wg /= splitk;
  const int mt = wg / ntn, nt = wg - mt * ntn;
  const int m0 = mt * BM, n0 = nt * BN;
  const int HoWo = p.Ho * p.Wo;
  const int M = p.B * HoWo;
  const int ntaps = p.kh * p.kw;
  const int cpt = p.Cin >> 5;                                // 32-channel slices of the first source
  const int cpt2 = p.in2 != nullptr ? p.Cin2 >> 5 : 0;       // ... of the K-concatenated second source (1x1 only)
  const int nsteps1 = ntaps * cpt, nsteps_all = nsteps1 + cpt2;
  const int s_begin = (int)(((long)nsteps_all * ks) / splitk);
  const int nsteps = (int)(((long)nsteps_all * (ks + 1)) / splitk) - s_begin;
  // the A operand's power of two (from the recorded |max| of the source tensors) and its inverse for the epilogue
  const int sexp = h2_in_scale_exp(p);
  const float a_scale = pow2f(sexp), h2_inv = pow2f(-sexp);

  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(
      (void*)p.in, 0, (int)((unsigned)p.B * p.in_Ha * p.in_Wa * p.in_ldc * 4u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_in2 = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.in2 != nullptr ? p.in2 : p.in), 0,
      (int)(p.in2 != nullptr ? (unsigned)p.B * p.in2_Ha * p.in2_Wa * p.in2_ldc * 4u : 0u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_wt = __builtin_amdgcn_make_buffer_rsrc(
      (void*)p.wt_split, 0, (int)((unsigned)ntn * nsteps_all * (unsigned)STAGE_B), 0x00020000);

  // ---- weights: wave w, instruction i copies the 1-KB piece i * 8 + w of the stage image
  // K-slice rotation (single-source 1x1 convs without split-K): workgroup (mt, .) starts its reduction at slice mt mod
  // nsteps and wraps.  All workgroups of such a launch start together and walk the channels at the same pace: without
  // the rotation every request in flight addresses the SAME 128-byte column of the pixels' 4-KB (1-KB ...) rows -- the same
  // few HBM channels (res4 conv1 streamed its input at 2 TB/s).  The summation order of a tile depends on mt.
  const int rot = (ntaps == 1 && cpt2 == 0 && splitk == 1 && (p.debug & 0x100) == 0) ? mt % nsteps : 0;
  unsigned l_b = ((unsigned)nt * (unsigned)nsteps_all + (unsigned)(s_begin + rot)) * (unsigned)STAGE_B;
  int b_wrap = nsteps - rot;                 // stages until the weight stream wraps to the first slice
  auto dma_b = [&](int boff) {
#pragma unroll
    for (int i = 0; i < NW; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_wt, ODT_LDS_PTR(lds + boff + (i * NWV + wave) * 1024), 16,
                                               lane * 16 + (i * NWV + wave) * 1024, (int)l_b, 0, 0);
    l_b += (unsigned)STAGE_B;
    if (--b_wrap == 0) l_b -= (unsigned)nsteps * (unsigned)STAGE_B;
  };
  dma_b(BOFF);                               // stage 0's weights: requested before the per-row address set-up below

  // ---- activations: thread -> (row (t >> 3) + (threads / 8) j, 16-byte column t & 7): eight lanes cover the 128 contiguous
  // bytes (32 channels) of a row's stage.  Per row: the byte offset of the tap-(0,0) input pixel and a bit per tap
  // (inside the image and m < M); a stage's offset is base + tap offset, or out of range (the load returns zeros).
  const int a_c = tid & 7, a_r = tid >> 3;
  const unsigned pix_bytes = (unsigned)p.in_ldc * 4u;
  int a_base[RA];
  unsigned a_mask[RA];
  const bool dense_in = ntaps == 1 && p.stride == 1 && p.pad_t == 0 && p.pad_l == 0 &&
                        p.H == p.in_Ha && p.W == p.in_Wa && p.Ho == p.H && p.Wo == p.W;
#pragma unroll
  for (int j = 0; j < RA; ++j) {
    const int m = m0 + a_r + AR * j;
    const bool ok = m < M;
    if (dense_in) {
      a_base[j] = (int)((unsigned)m * pix_bytes + a_c * 16u);
      a_mask[j] = ok ? 1u : 0u;
    } else {
      const int mm = ok ? m : 0;
      const int n = sfast_div(mm, p.div_howo_mul, p.div_howo_sh), r = mm - n * HoWo;
      const int ho = sfast_div(r, p.div_wo_mul, p.div_wo_sh), wo = r - ho * p.Wo;
      const int hi0 = ho * p.stride - p.pad_t, wi0 = wo * p.stride - p.pad_l;
      a_base[j] = (int)(((unsigned)n * p.in_Ha * p.in_Wa + (unsigned)(hi0 * p.in_Wa + wi0)) * pix_bytes + a_c * 16u);
      unsigned mk = 0;
      for (int t = 0, khh = 0, kww = 0; t < ntaps; ++t) {
        const int hi = hi0 + khh * p.dil, wi = wi0 + kww * p.dil;
        if (ok && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W) mk |= 1u << t;
        if (++kww == p.kw) { kww = 0; ++khh; }
      }
      a_mask[j] = mk;
    }
  }
  // load stream position: (32-channel slice, tap) with the tap innermost; then the second source's slices
  // (a split-K range starts inside the first source: the policy keeps split-K off for second-source convs)
  int l_cs = s_begin / ntaps + rot, l_tap = s_begin - (s_begin / ntaps) * ntaps;
  int l_kh = l_tap / p.kw, l_kw = l_tap - l_kh * p.kw;
  bool l_src2 = false;
  unsigned tapoff = (unsigned)(l_kh * p.dil * p.in_Wa + l_kw * p.dil) * pix_bytes;     // byte offset of the stream's tap (wave-uniform)
  // second source: its pixel offsets replace the first source's (one "tap", valid wherever the row is)
  auto set_src2 = [&]() {
#pragma unroll
    for (int j = 0; j < RA; ++j) {
      const int m = m0 + a_r + AR * j;
      const bool ok = m < M;
      const int mm = ok ? m : 0;
      const int n = sfast_div(mm, p.div_howo_mul, p.div_howo_sh), r = mm - n * HoWo;
      const int ho = sfast_div(r, p.div_wo_mul, p.div_wo_sh), wo = r - ho * p.Wo;
      const unsigned pix = ((unsigned)n * p.in2_Ha + (unsigned)(ho * p.in2_stride)) * p.in2_Wa + (unsigned)(wo * p.in2_stride);
      a_base[j] = (int)(pix * (unsigned)p.in2_ldc * 4u + a_c * 16u);
      a_mask[j] = ok ? 1u : 0u;
    }
    tapoff = 0; l_tap = 0;
  };
  f32x4 ga[RA];
  // A/B: a 1x1 layer with one n-tile reads every activation byte once, by one workgroup: non-temporal hint on those loads
  const bool a_nt = (p.debug & 0x800) != 0 && ntn == 1 && ntaps == 1;
  auto load_a = [&]() {
#pragma unroll
    for (int j = 0; j < RA; ++j) {
      const unsigned off = ((a_mask[j] >> l_tap) & 1u) ? (unsigned)a_base[j] + tapoff : kOOB;
      ga[j] = a_nt ? (f32x4)__builtin_amdgcn_raw_buffer_load_b128(l_src2 ? rs_in2 : rs_in, (int)off, l_cs * 128, 2)
                   : (f32x4)__builtin_amdgcn_raw_buffer_load_b128(l_src2 ? rs_in2 : rs_in, (int)off, l_cs * 128, 0);
    }
    // advance
    if (l_src2) {
      ++l_cs;
    } else if (ntaps == 1) {
      if (++l_cs == cpt) {
        l_cs = 0;                              // (second source: its first slice; rotation: wrap to the first slice)
        if (cpt2 > 0) { l_src2 = true; set_src2(); }
      }
    } else {
      ++l_tap;
      if (++l_kw == p.kw) { l_kw = 0; ++l_kh; }
      if (l_tap == ntaps) { l_tap = 0; l_kh = 0; l_kw = 0; ++l_cs; }
      tapoff = (unsigned)(l_kh * p.dil * p.in_Wa + l_kw * p.dil) * pix_bytes;
    }
  };
  auto store_slot = [&](int abuf, int j) {
    unsigned h0, l0, h1, l1;
    split2h(ga[j][0], ga[j][1], a_scale, h0, l0);
    split2h(ga[j][2], ga[j][3], a_scale, h1, l1);
    unsigned char* d = lds + abuf + (a_c >> 1) * AKG + (a_r + AR * j) * 16 + (a_c & 1) * 8;
    *reinterpret_cast<u32x2*>(d) = u32x2{h0, h1};
    *reinterpret_cast<u32x2*>(d + APL) = u32x2{l0, l1};
  };

  f32x16 acc[2][TN];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int fr = lane & 31, fg = lane >> 5;
  // ---- prologue: stage 0 complete, stage 1's A in registers, its weights in flight behind the barrier
  load_a();
  stamp(6);
#pragma unroll
  for (int j = 0; j < RA; ++j) store_slot(0, j);
  if (nsteps > 1) load_a();
  if (nsteps > 1) ODT_WAIT_VM_LGKM0(RA); else ODT_WAIT_VM_LGKM0(0);
  __builtin_amdgcn_s_barrier();
  if (nsteps > 1) dma_b(BOFF + STAGE_B);
  stamp(7); stamp(1);

  // fragments: fa[k-step][piece][t], fb[buffer][piece] (the next column group's operands are read while this one computes)
  f16x8 fa[2][2][2], fb[2][2];
  const int a_rd = fg * AKG + (wm * 64 + fr) * 16;
  const int b_rd = fg * BKG + (wn * TN * 32 + fr) * 16;
  auto rdA = [&](int abuf, int kst, int q) {
#pragma unroll
    for (int t = 0; t < 2; ++t) fa[kst][q][t] = *reinterpret_cast<const f16x8*>(lds + abuf + q * APL + kst * 2 * AKG + a_rd + t * 512);
  };
  auto rdB = [&](int bbuf, int kst, int j, int dst) {
#pragma unroll
    for (int q = 0; q < 2; ++q) fb[dst][q] = *reinterpret_cast<const f16x8*>(lds + bbuf + q * BPL + kst * 2 * BKG + b_rd + j * 512);
  };
  rdA(0, 0, 1); rdA(0, 0, 0);
  rdB(BOFF, 0, 0, 0);

#define ODT_MF(kst, qa, qb, j, bsel) { acc[0][j] = ODT_MFMA_F16(fa[kst][qa][0], fb[bsel][qb], acc[0][j]); \
                                        acc[1][j] = ODT_MFMA_F16(fa[kst][qa][1], fb[bsel][qb], acc[1][j]); }
#define ODT_FENCE() __builtin_amdgcn_sched_barrier(0)
  int a_cur = 0, a_nxt = ASTG, b_cur = BOFF, b_nxt = BOFF + STAGE_B;
  // One stage = 2 TN column groups (k-step, j).  NEXT: stage c+1 exists (its A: registers -> LDS in the first groups; its
  // first fragments are read behind the barrier, under the last group's MFMAs); PRE: stage c+2 exists (fetch its A, start
  // its weight DMA behind the barrier, into the buffer this stage is leaving).  The K loop is peeled so that no MFMA sits
  // in a conditional arm.
  auto step = [&](auto NEXT, auto PRE) {
    constexpr bool next = decltype(NEXT)::value, pre = decltype(PRE)::value;
    constexpr int NG = 2 * TN;
    ODT_FENCE();
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int kst = g / TN, j = g % TN, bsel = g & 1;
      const bool last = g == NG - 1;
      if (last) {
        // stage c+1 must be complete before its first fragment reads below: own LDS stores, own DMA of stage c+1 (issued
        // one stage ago; this stage's A fetch may stay in flight), then the barrier -- which also releases stage c's buffers
        if constexpr (pre) ODT_WAIT_VM_LGKM0(RA); else ODT_WAIT_VM_LGKM0(0);
        __builtin_amdgcn_s_barrier();
        ODT_FENCE();
        if constexpr (pre) dma_b(b_cur);
        if constexpr (next) rdB(b_nxt, 0, 0, bsel ^ 1);
      } else {
        rdB(b_cur, (g + 1) / TN, (g + 1) % TN, bsel ^ 1);
        // (a single column group per k-step: the second k-step's A fragments go out with the first group's operands)
        if (TN == 1) { rdA(a_cur, 1, 1); rdA(a_cur, 1, 0); }
      }
      ODT_FENCE();
      ODT_MF(kst, 1, 0, j, bsel); ODT_FENCE();             // lo * hi
      if (last) {
        if constexpr (next) rdA(a_nxt, 0, 1);
      } else if (TN == 4) {
        if (g == 0) { if constexpr (next) store_slot(a_nxt, 0); }
        if (g == 1) { if constexpr (next) store_slot(a_nxt, 2); }
        if (g == 2) { if constexpr (pre) load_a(); }
        if (g == 3) rdA(a_cur, 1, 0);
      } else if (TN == 2) {
        if constexpr (next) {                  // RA slots over the three column groups in front of the barrier
          constexpr int SPG = (RA + 2) / 3;
#pragma unroll
          for (int q = 0; q < SPG; ++q)
            if (g * SPG + q < RA) store_slot(a_nxt, g * SPG + q);
        }
      } else {
        if constexpr (next) { store_slot(a_nxt, 0); store_slot(a_nxt, 1); }
      }
      ODT_FENCE();
      ODT_MF(kst, 0, 1, j, bsel); ODT_FENCE();             // hi * lo
      if (last) {
        if constexpr (next) rdA(a_nxt, 0, 0);
      } else if (TN == 4) {
        if (g == 0) { if constexpr (next) store_slot(a_nxt, 1); }
        if (g == 1) { if constexpr (next) store_slot(a_nxt, 3); }
        if (g == 2) rdA(a_cur, 1, 1);
      } else if (TN == 2) {
        if (g == 0) { rdA(a_cur, 1, 1); rdA(a_cur, 1, 0); }
        // (the next fetch reuses the registers: behind the last slot's store)
        if (g == (RA > 2 * ((RA + 2) / 3) ? 2 : 1)) { if constexpr (pre) load_a(); }
      } else {
        if constexpr (next) { store_slot(a_nxt, 2); store_slot(a_nxt, 3); }
        if constexpr (pre) load_a();
      }
      ODT_FENCE();
      ODT_MF(kst, 0, 0, j, bsel); ODT_FENCE();             // hi * hi
    }
    { const int t = a_cur; a_cur = a_nxt; a_nxt = t; }
    { const int t = b_cur; b_cur = b_nxt; b_nxt = t; }
  };
  {
    int c = 0;
    for (; c + 2 < nsteps; ++c) step(std::true_type{}, std::true_type{});
    if (c + 1 < nsteps) { step(std::true_type{}, std::false_type{}); ++c; }
    step(std::false_type{}, std::false_type{});
  }
  stamp(2);
  split3_epilogue<WM, WN, TN, G::LDS, TRACE, G::NTHR>(p, acc, lds, m0, n0, M, HoWo, ks, splitk, tid, wm, wn, fr, fg, h2_inv);
  stamp(5);
}

#undef ODT_MF
#undef ODT_FENCE

// ---- weight image -----------------------------------------------------------------------------------------------------
// t_n of a weight row: the power of two that takes its |max| into [2^14, 2^15); chinv[n] = 2^-t_n (1 for the zero rows of
// the padding).  One workgroup per row.
__global__ void __launch_bounds__(256) h2_rowscale_kernel(const float* __restrict__ wt, int Cout, int K, float* __restrict__ chinv) {
  __shared__ float red[4];
  const int n = (int)blockIdx.x;
  float m = 0.f;
  if (n < Cout)
    for (int k = (int)threadIdx.x; k < K; k += 256) m = fmaxf(m, fabsf(wt[(size_t)n * K + k]));
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    chinv[n] = pow2f(-h2_scale_exp(__float_as_uint(m)));
  }
}

// image: [n-tile][stage][piece 2][k-group 4][BN n][8 k] f16, stage order = (32-channel slice, tap) of the first source, then
// the second source's slices; wt is [Cout][tap][Cin] (+ [Cin2] behind it); row n is multiplied by 2^t_n = 1 / chinv[n]
__global__ void split_weights_h2_kernel(const float* __restrict__ wt, int Cout, int K, int SBN, int ntaps, int Cin,
                                        unsigned short* __restrict__ img, const float* __restrict__ chinv) {
  const int nst = K >> 5, nst1 = ntaps * (Cin >> 5);
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;       // (n, stage, k-group)
  const long total = (long)cout_padded(Cout) * nst * 4;       // n over the padded Cout: zero rows behind the last channel
  if (idx >= total) return;
  const int n = (int)(idx / (nst * 4)), rem = (int)(idx - (long)n * (nst * 4)), st = rem >> 2, kg = rem & 3;
  int k0;
  if (st < nst1) { const int cs = st / ntaps, tap = st - cs * ntaps; k0 = tap * Cin + cs * 32; }
  else k0 = ntaps * Cin + (st - nst1) * 32;
  k0 += kg * 8;
  const int tn = n / SBN, nn = n - tn * SBN;
  const float s = 1.0f / chinv[n];                            // (a power of two: exact)
  for (int e = 0; e < 8; e += 2) {
    unsigned piece[2];
    const float w0 = n < Cout ? wt[(size_t)n * K + k0 + e] : 0.f, w1 = n < Cout ? wt[(size_t)n * K + k0 + e + 1] : 0.f;
    split2h(w0, w1, s, piece[0], piece[1]);
    for (int q = 0; q < 2; ++q) {
      const size_t at = ((((size_t)(tn * nst + st) * 2 + q) * 4 + kg) * SBN + nn) * 8 + e;
      img[at] = (unsigned short)(piece[q] & 0xffffu);
      img[at + 1] = (unsigned short)(piece[q] >> 16);
    }
  }
}

// image of a fused 1x1 conv (ConvParams::f_wt; conv_h2k_kernel<.., FUSE>): [chunk of 32 columns][piece 2][K half 2][k16 step
// NS][lane half fg 2][column 32][8 f16].  The producer's wave (wm, wn) holds, for each of its pixels, the channels
// wn * KH + [0, KH) (KH = K / 2) in accumulator registers; step s = (32-channel block j = s / 2, register half h = s % 2) takes,
// from lane half fg, element e the channel j * 32 + 16 h + (e % 4) + 8 (e / 4) + 4 fg -- the MFMA C layout read as an operand
// fragment -- so the weights' k runs in that order too.  wt is [Cout][K]; row n is multiplied by 2^t_n = 1 / chinv[n].
__global__ void split_weights_h2f_kernel(const float* __restrict__ wt, int Cout, int K, unsigned short* __restrict__ img,
                                         const float* __restrict__ chinv) {
  const int KH = K >> 1, NS = KH >> 4;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;       // (n, K half, step, lane half)
  const long total = (long)Cout * 2 * NS * 2;
  if (idx >= total) return;
  const int fg = (int)(idx & 1), s = (int)((idx >> 1) % NS), wn = (int)((idx / (2 * NS)) & 1), n = (int)(idx / (4 * NS));
  const int c = n >> 5, fr = n & 31;
  const float sc = 1.0f / chinv[n];
  for (int e = 0; e < 8; e += 2) {
    const int k0 = wn * KH + (s >> 1) * 32 + (s & 1) * 16 + (e & 3) + 8 * (e >> 2) + 4 * fg;     // (e even: k0 + 1 is element e + 1)
    unsigned piece[2];
    split2h(wt[(size_t)n * K + k0], wt[(size_t)n * K + k0 + 1], sc, piece[0], piece[1]);
    for (int q = 0; q < 2; ++q) {
      const size_t at = ((((((size_t)c * 2 + q) * 2 + wn) * NS + s) * 2 + fg) * 32 + fr) * 8 + e;
      img[at] = (unsigned short)(piece[q] & 0xffffu);
      img[at + 1] = (unsigned short)(piece[q] >> 16);
    }
  }
}

// |max| of a dense f32 array into *slot (stand-alone calls: a tensor nobody recorded a range for)
__global__ void __launch_bounds__(256) tensor_amax_kernel(const float* __restrict__ x, size_t n, unsigned* __restrict__ slot) {
  float m = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) m = fmaxf(m, fabsf(x[i]));
  publish_amax(slot, m, (int)threadIdx.x);
}

}  // namespace

size_t conv_h2_weight_bytes(int Cout, int K) { return (size_t)cout_padded(Cout) * K * 4 + (size_t)cout_padded(Cout) * 4; }

// image + the per-column inverse scales behind it (p.h2_chinv must point there: conv_h2_chinv)
const float* conv_h2_chinv(const void* img, int Cout, int K) {
  return reinterpret_cast<const float*>(reinterpret_cast<const char*>(img) + (size_t)cout_padded(Cout) * K * 4);
}

int conv_make_h2_weights(const ConvParams& p, void* img_dev, hipStream_t stream) {
  const int K = p.kh * p.kw * p.Cin + (p.in2 != nullptr ? p.Cin2 : 0);
  const int bn = p.wt_split_bn;
  ODT_CHECK((bn == 256 || bn == 128 || bn == 64) && p.Cin % 32 == 0 && (p.in2 == nullptr || p.Cin2 % 32 == 0) && cout_padded(p.Cout) % bn == 0,
            "conv_make_h2_weights: 256- / 128- / 64-wide n-tiles and 32-channel slices required");
  float* chinv = const_cast<float*>(conv_h2_chinv(img_dev, p.Cout, K));
  hipLaunchKernelGGL(h2_rowscale_kernel, dim3((unsigned)cout_padded(p.Cout)), dim3(256), 0, stream, p.wt, p.Cout, K, chinv);
  const long total = (long)cout_padded(p.Cout) * (K >> 5) * 4;
  hipLaunchKernelGGL(split_weights_h2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, p.wt, p.Cout, K,
                     bn, p.kh * p.kw, p.Cin, (unsigned short*)img_dev, chinv);
  ODT_HIP(hipGetLastError());
  return 0;
}

size_t conv_h2f_weight_bytes(int Cout, int K) { return (size_t)Cout * K * 4 + (size_t)Cout * 4; }
const float* conv_h2f_chinv(const void* img, int Cout, int K) {
  return reinterpret_cast<const float*>(reinterpret_cast<const char*>(img) + (size_t)Cout * K * 4);
}
int conv_make_h2f_weights(const float* wt, int Cout, int K, void* img_dev, hipStream_t stream) {
  ODT_CHECK((K == 256 || K == 128 || K == 64) && Cout % 32 == 0 && Cout > 0, "conv_make_h2f_weights: K = 64 / 128 / 256 and Cout % 32 == 0 required");
  float* chinv = const_cast<float*>(conv_h2f_chinv(img_dev, Cout, K));
  hipLaunchKernelGGL(h2_rowscale_kernel, dim3((unsigned)Cout), dim3(256), 0, stream, wt, Cout, K, chinv);
  const long total = (long)Cout * (K >> 3);
  hipLaunchKernelGGL(split_weights_h2f_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, wt, Cout, K,
                     (unsigned short*)img_dev, chinv);
  ODT_HIP(hipGetLastError());
  return 0;
}

// may the 1x1 conv b (reading a.out and nothing else of a) run in the epilogue of the KH x 3 conv a?  a: conv_h2k_kernel on
// 256-wide (res4), 128-wide (res3) or 64-wide (res2) n-tiles = its whole Cout, plain dense output, ReLU or none; b: dense same-size 1x1, single source,
// no residual or a same-shape one, on the fp16x2 family as well (so its consumers find a recorded range)
bool conv_h2f_fusable(const ConvParams& a, const ConvParams& b) {
  const bool a_ok = a.wt_split != nullptr && a.wt_split_kind == 2 && a.wt_split_kwr == 1 && a.wt_split_bm == 256 &&
                    a.wt_split_bn == a.Cout && (a.Cout == 256 || a.Cout == 128 || a.Cout == 64) && a.splitk <= 1 && a.head_wt == nullptr &&
                    a.res_mode == 0 && a.in2 == nullptr && a.relu <= 1 && a.nlvl <= 1 && a.out_oy == 0 && a.out_ox == 0 &&
                    a.out_H == a.Ho && a.out_W == a.Wo && a.f_wt == nullptr;
  const bool b_ok = b.in == a.out && b.kh == 1 && b.kw == 1 && b.stride == 1 && b.pad_t == 0 && b.pad_l == 0 && b.Cin == a.Cout &&
                    b.in_ldc == a.out_ldc && b.in2 == nullptr && b.B == a.B && b.H == a.Ho && b.W == a.Wo && b.Ho == a.Ho &&
                    b.Wo == a.Wo && b.in_Ha == a.out_H && b.in_Wa == a.out_W && b.Cout % 32 == 0 && b.out_oy == 0 && b.out_ox == 0 &&
                    b.out_H == b.Ho && b.out_W == b.Wo && b.out_ldc % 4 == 0 && b.out_ldc >= b.Cout &&
                    (b.res_mode == 0 || (b.res_mode == 1 && b.res_H == b.Ho && b.res_W == b.Wo && b.res_ldc % 4 == 0)) &&
                    b.relu <= 1 && b.nlvl <= 1 && b.head_wt == nullptr && b.splitk <= 1 && b.wt_split != nullptr && b.wt_split_kind == 2 &&
                    b.f_wt == nullptr && (double)b.B * b.Ho * b.Wo * b.out_ldc * 4.0 < 2147483648.0;
  return a_ok && b_ok;
}

int launch_tensor_amax(const float* x, size_t n, unsigned* slot, hipStream_t stream) {
  const unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, 2048);
  hipLaunchKernelGGL(tensor_amax_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, stream, x, n, slot);
  ODT_HIP(hipGetLastError());
  return 0;
}

int launch_conv_h2(const ConvParams& p, const ConvParams* dev, hipStream_t stream) {
  if (p.stem_pool) return launch_conv_stem(p, dev, stream);
  const long M = (long)p.B * p.Ho * p.Wo;
  const int bn = p.wt_split_bn;
  const int bm = p.wt_split_bm;
  ODT_CHECK(((bm == 256 && (bn >= 128 || p.wt_split_kwr)) || (bm == 128 && bn <= 128 && !p.wt_split_kwr) ||
             (bm == 512 && bn == 64 && (!p.wt_split_kwr || p.Ho * p.Wo >= 512)) || (bm == 64 && (bn == 128 || bn == 64) && !p.wt_split_kwr)) && (bn == 256 || bn == 128 || bn == 64) && p.Cin % 32 == 0 && p.kh * p.kw <= 32 && p.in_amax != nullptr &&
            p.h2_chinv != nullptr && (p.in2 == nullptr || (p.in2_amax != nullptr && p.Cin2 % 32 == 0)) && p.nlvl <= 1,
            "conv h2: unsupported tile / shape, or no recorded input range");
  const int sk = p.splitk > 1 ? p.splitk : 1;
  ODT_CHECK(sk == 1 || (p.partial != nullptr && p.in2 == nullptr && p.head_wt == nullptr && (p.kh * p.kw * p.Cin >> 5) >= sk),
            "conv h2: split-K needs a partial buffer, a single source and at least one stage per range");
  const unsigned grid = (unsigned)(((M + bm - 1) / bm) * (cout_padded(p.Cout) / bn) * sk);
  if (p.wt_split_kwr) {
    ODT_CHECK(p.kw == 3 && p.stride == 1 && p.in_Wa == p.Wo && p.in2 == nullptr && 2 * p.dil <= 4 &&
              (sk == 1 || (bm == 256 && bn >= 128 && p.f_wt == nullptr && p.kh * (p.Cin >> 5) >= sk)), "conv h2k: unsupported shape");
    ODT_CHECK(p.f_wt == nullptr || (bm == 256 && bn == p.Cout && (bn == 256 || bn == 128 || bn == 64) && p.head_wt == nullptr && p.res_mode == 0 && p.relu <= 1 && p.f_cout % 32 == 0 &&
                                    p.f_cout > 0 && p.f_cout <= 1024 && p.f_out != nullptr && p.f_chinv != nullptr && p.f_bias != nullptr && p.f_out_ldc % 4 == 0 &&
                                    (p.f_res == nullptr || p.f_res_ldc % 4 == 0) && (double)M * p.f_out_ldc * 4.0 < 2147483648.0 &&
                                    (p.f_res == nullptr || (double)M * p.f_res_ldc * 4.0 < 2147483648.0)),
              "conv h2k: unsupported fused 1x1 tail");
    launch_conv_h2k(p, dev, grid, stream);
  } else if (p.f_wt != nullptr) {
    ODT_CHECK(false, "conv h2: a fused 1x1 tail needs the kw-reuse kernel");
  } else if (bm == 64 && conv_h2d_fits(p) && !env_knob_off(K_CONV_H2_BK64)) {
    // the dense 1x1 reductions on two-wave tiles: double stages (conv_h2d.hip; ODT_CONV_H2_BK64=0: the single-stage kernel, A/B)
    launch_conv_h2d(p, dev, grid, stream);
  } else if (bm == 64 && bn == 64) {         // ... 64 x 64 tiles: twice the workgroups again (latency-bound reductions at b = 1)
    hipLaunchKernelGGL((conv_h2_kernel<1, 1, false>), dim3(grid), dim3(128), 0, stream, dev);
  } else if (bm == 64) {                     // few-row layers (b = 1 below res3): 64 x 128 tiles on two waves, three workgroups per CU, no split-K
    hipLaunchKernelGGL((conv_h2_kernel<2, 1, false>), dim3(grid), dim3(128), 0, stream, dev);
  } else if (bn == 64 && bm == 512) {
    hipLaunchKernelGGL((conv_h2_kernel<2, 8, false, 1>), dim3(grid), dim3(512), 0, stream, dev);
  } else if (bn == 64) {
    hipLaunchKernelGGL((conv_h2_kernel<1, 2, false>), dim3(grid), dim3(256), 0, stream, dev);
  } else if (bn == 256) {
    if (p.trace != nullptr) hipLaunchKernelGGL((conv_h2_kernel<4, 4, true>), dim3(grid), dim3(512), 0, stream, dev);
    else hipLaunchKernelGGL((conv_h2_kernel<4, 4, false>), dim3(grid), dim3(512), 0, stream, dev);
  } else if (bm == 256) {
    hipLaunchKernelGGL((conv_h2_kernel<2, 4, false>), dim3(grid), dim3(512), 0, stream, dev);
  } else {
    if (p.trace != nullptr) hipLaunchKernelGGL((conv_h2_kernel<2, 2, true>), dim3(grid), dim3(256), 0, stream, dev);
    else hipLaunchKernelGGL((conv_h2_kernel<2, 2, false>), dim3(grid), dim3(256), 0, stream, dev);
  }
  if (sk > 1) launch_split_reduce(p, dev, stream);
  ODT_HIP(hipGetLastError());
  return 0;
}

}  // namespace odt
