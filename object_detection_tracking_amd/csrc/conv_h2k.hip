// conv_h2k_kernel: the kw-reuse kernel of the fp16x2 family (see conv_h2.hip for the family's arithmetic and loop structure; a
// translation unit of its own: the two kernels' instantiations compile in parallel).
#include "conv_split_epilogue.hpp"

namespace odt {

namespace {

#define ODT_MFMA_F16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
// (FUSE: operands swapped -- the tile comes out transposed, lanes along the pixels and registers along the channels, which
// is the layout the fused 1x1 conv consumes as its operand fragment)
#define ODT_MF(kst, qa, qb, j, bsel) { if constexpr (FUSE) {                                                   \
                                          acc[0][j] = ODT_MFMA_F16(fb[bsel][qb], fa[kst][qa][0], acc[0][j]);   \
                                          acc[1][j] = ODT_MFMA_F16(fb[bsel][qb], fa[kst][qa][1], acc[1][j]);   \
                                        } else {                                                                \
                                          acc[0][j] = ODT_MFMA_F16(fa[kst][qa][0], fb[bsel][qb], acc[0][j]);   \
                                          acc[1][j] = ODT_MFMA_F16(fa[kst][qa][1], fb[bsel][qb], acc[1][j]); } }
#define ODT_FENCE() __builtin_amdgcn_sched_barrier(0)

// ---------------------------------------------------------------------------------------------------------
// conv_h2k_kernel: conv_h2_kernel for stride-1 KH x 3 convs whose input rows have the output's pitch (in_Wa == Wo: the 3x3
// layers of res3 / res4 / res5, the FPN post-hoc and RPN convs).  A group = (32-channel slice, kh): its three stages (kw =
// 0, 1, 2) read ONE staged run of input pixels at row offsets 0, dil, 2 dil (conv_split3k_kernel's scheme):
//   * the 256 output pixels of a tile are consecutive in (n, ho, wo); within an image their tap-(kh, 0) input pixels are
//     consecutive too, so a group's stage is the run [first - pad_l, last - pad_l + 2 dil]; a tile that crosses an image
//     boundary stages two runs back to back (capacity 256 + 2 x 2 dil rows);
//   * taps outside the image read a zero row of the stage: a per-lane 9-bit validity mask picks the fragment address;
//   * A: two buffers (this group / next group); the next group's fetch (5 x 16 B per thread) is issued in the group's first
//     stage and split + stored in its third; B: the two-deep DMA ring of conv_h2_kernel.
// WN = 2 (default): waves 4 x 2, tile 256 x 64 TN; WN = 1: the eight waves stacked along M, tile 512 x 32 TN -- for the 64-wide
// layers (TN = 2): a wave tile of 64 x 64 carries 24 MFMAs per stage instead of the 12 of a 64 x 32 one (res2 conv2)
template <int TN, bool FUSE = false, int WN = 2>
struct H2kCfg {
  static constexpr int WM = 8 / WN;
  static constexpr int BM = 64 * WM, BN = 32 * TN * WN;
  static constexpr int PR = BM + 16;                         // stage rows: BM + 2 runs x 2 dil (dil <= 2) + the zero row, padded
  static constexpr int ZR = PR - 1;                          // the zero row
  static constexpr int AKG = PR * 16 + 32, APL = 4 * AKG, ABUF = 2 * APL;   // (32-B pad: see H2Cfg)
  static constexpr int BKG = BN * 16, BPL = 4 * BKG, STAGE_B = 2 * BPL;
  static constexpr int BOFF = 2 * ABUF;
  static constexpr int RING = BOFF + 2 * STAGE_B;
  static constexpr int CTILE = 128 * (BN + 4) * 4;
  // fused 1x1 tail: two 32-column chunks of its weight image + two result tiles [256][32 + 4] f32 (this chunk / the previous one)
  // (+ the producer's own column constants, [2][256] f32 behind everything the main loop and the tail use)
  static constexpr int F_WCH = 2 * 2 * (2 * TN) * 1024, F_CS = 36, F_COFF = 2 * F_WCH, F_CEND = F_COFF + 2 * 256 * F_CS * 4;
  static constexpr int LDS0 = RING > CTILE ? RING : CTILE;
  static constexpr int F_KOFF = F_CEND > LDS0 ? F_CEND : LDS0;
  static constexpr int F_K3OFF = F_KOFF + 2048;              // the fused conv's column constants, [2][1024] f32
  static constexpr int LDS = FUSE ? F_K3OFF + 8192 : LDS0;
  static constexpr int NW = STAGE_B / 1024 / 8;
  static constexpr int RA = (PR + 63) / 64;                  // A fetch instructions per thread and group (rows t >> 3 + 64 j)
  static_assert(LDS <= 160 * 1024 && STAGE_B % 8192 == 0 && (!FUSE || WN == 2), "LDS");
};

// ---------------------------------------------------------------------------------------------------------
// Fused 1x1 conv behind the KH x 3 conv (ConvParams::f_wt; the bottleneck's conv2 -> conv3 (+ shortcut) + ReLU,
// nn.py:503-521).  On entry acc[i][j] holds the TRANSPOSED tile of the wave (operands swapped in the main loop): lane
// (fr, fg), register r = pixel row wm 64 + i 32 + fr, channel wn 32 TN + j 32 + (r % 4) + 8 (r / 4) + 4 fg, in scaled units.
//   1. y = act(acc * 2^-s 2^-t_c + bias_c) in registers (the values the unfused conv would have stored, bit for bit);
//   2. per pixel row and K half (= per lane pair fr / fr + 32 of a wave) the power of two that takes the row's |max| into
//      [2^14, 2^15); y 2^sy = hi + lo (f16 pairs): registers 8 h .. 8 h + 7 of acc[i][j] ARE the operand fragment of k16 step
//      (j, h) -- nothing moves (the weight image carries k in this order: split_weights_h2f_kernel);
//   3. the two waves of a row block (wm, 0) / (wm, 1) swap halves through LDS, once per tile: wave (wm, wn) keeps pixel
//      block i = wn and receives that block's other K half (pieces + the rows' powers of two), lane for lane -- afterwards
//      every wave owns 32 pixels x the whole K, and no partial sums ever have to meet;
//   4. per 32-column chunk of the 1x1 conv: weight pieces by LDS-DMA two chunks ahead, 4 TN k16 steps x 3 products into two
//      accumulators (own half / received half: each scaled back by its rows' 2^-sy), the [256][32] result to one of two LDS
//      tiles, ONE barrier, and the rows of 16-byte chunks (x 2^-t_n + bias (+ residual), activation, store, |max|) go out
//      UNDER the next chunk's MFMAs; the residual chunks are fetched two chunks ahead into the registers the previous
//      chunk's have just left.
template <int TN, bool TRACE>
__device__ __forceinline__ void h2f_tail(const ConvParams& p, f32x16 (&acc)[2][TN], unsigned char* lds, int m0, int M,
                                         int wave, int wm, int wn, float h2_inv) {
  using G = H2kCfg<TN, true>;
  constexpr int NS = 2 * TN, WCH = G::F_WCH, CS = G::F_CS, COFF = G::F_COFF, CBUF = 256 * CS * 4;
  // (the lane id is recomputed here: nothing per-lane stays live across the main loop, whose registers are all taken)
  ODT_FENCE();
  const int lane = ODT_LANE_ID();
  const int tid = wave * 64 + lane, fr = lane & 31, fg = lane >> 5;
  const int nch = p.f_cout >> 5;

  // ---- 1. + 2. the producer's epilogue arithmetic in registers, the per-row power of two, the pieces.  Two passes over the
  // accumulators (row |max| first, then value -> pieces), the column constants (2^-s 2^-t_c, bias_c: staged in LDS by the
  // prologue) read twice: the values are never written back, so the pieces take the registers the accumulators leave
  const float* kc = reinterpret_cast<const float*>(lds + G::F_KOFF);
  const float act2_lo = p.relu == 1 ? 0.f : -__builtin_huge_valf();
  float mx[2] = {0.f, 0.f};
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int col = wn * 32 * TN + j * 32 + 8 * g + 4 * fg;
      const f32x4 sc = *reinterpret_cast<const f32x4*>(kc + col), bs = *reinterpret_cast<const f32x4*>(kc + 256 + col);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v = acc[i][j][4 * g + e] * sc[e];
          v += bs[e];
          mx[i] = fmaxf(mx[i], fabsf(fmaxf(v, act2_lo)));
        }
      if (g & 1) ODT_FENCE();               // (bounds the constants in flight: the accumulators hold half the registers)
    }
  // (both maxima are complete HERE and the second pass re-reads the constants: without the pins the compiler sinks one
  // row block's pass behind the other's exponent arithmetic and keeps all 128 constants in registers across -- spills)
  ODT_PIN2(mx[0], mx[1]);
  asm volatile("" ::: "memory");
  // the row's |max| over BOTH K halves: the partner wave (wm, 1 - wn) covers the other 32 TN channels of the same 64 rows
  constexpr int XS = 8 * 2 * NS * 1024;     // (behind the piece exchange area of step 3)
  float* xs = reinterpret_cast<float*>(lds + XS);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    mx[i] = fmaxf(mx[i], __shfl_xor(mx[i], 32));
    xs[wave * 128 + i * 64 + lane] = mx[i];
  }
  ODT_BARRIER_LDS();
  float ys[2], yinv[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const float m = fmaxf(mx[i], xs[(wave ^ 1) * 128 + i * 64 + lane]);
    // h2_scale_exp without control flow
    const int be = (int)((__float_as_uint(m) >> 23) & 0xffu);
    int e = 14 - (be - 127);
    e = e > 100 ? 100 : (e < -100 ? -100 : e);
    e = (be == 0) | (be == 255) ? 0 : e;
    ys[i] = pow2f(e); yinv[i] = pow2f(-e);
  }
  ODT_PIN2(ys[0], ys[1]);
  // pieces: [pixel block][hi / lo][k16 step t = 2 j + h of this wave's K half]
  u32x4 yq[2][2][NS];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int col = wn * 32 * TN + j * 32 + 16 * h + 4 * fg;
      const f32x4 sc0 = *reinterpret_cast<const f32x4*>(kc + col), bs0 = *reinterpret_cast<const f32x4*>(kc + 256 + col);
      const f32x4 sc1 = *reinterpret_cast<const f32x4*>(kc + col + 8), bs1 = *reinterpret_cast<const f32x4*>(kc + 256 + col + 8);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float t0 = acc[i][j][8 * h + e] * sc0[e], t1 = acc[i][j][8 * h + 4 + e] * sc1[e];
          t0 += bs0[e]; t1 += bs1[e];
          v[e] = fmaxf(t0, act2_lo); v[4 + e] = fmaxf(t1, act2_lo);
        }
        u32x4 hq, lq;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          unsigned a, b;
          split2h(v[2 * t], v[2 * t + 1], ys[i], a, b);
          hq[t] = a; lq[t] = b;
        }
        ODT_PIN2(hq, lq);                   // (computed HERE, from constants that die here: see the pins above)
        yq[i][0][2 * j + h] = hq; yq[i][1][2 * j + h] = lq;
      }
      ODT_FENCE();
    }
  ODT_STAMP(3);

  // ---- 3. swap halves with the partner wave (wm, 1 - wn): give pixel block 1 - wn, keep block wn.  Exchange area: wave w's
  // 2 * NS fragments at w * XW (lane-linear kilobytes), its rows' inverse powers of two behind all of them.  The main loop's
  // ring is free (its last barrier sits behind every fragment read) and the weight DMA starts after the swap.
  constexpr int XW = 2 * NS * 1024;
  static_assert(XS == 8 * XW && XS + 8 * 512 <= G::F_KOFF, "exchange area");
  u32x4 yo[2][NS], yr[2][NS];               // own / received K half of the kept block: [hi / lo][step]
  const float yinv_k = wn == 0 ? yinv[0] : yinv[1];
  {
    unsigned char* xw = lds + wave * XW + lane * 16;
    const unsigned char* xr = lds + (wave ^ 1) * XW + lane * 16;
    auto give = [&](auto IC) {
      constexpr int ig = decltype(IC)::value;       // the block given away
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int t = 0; t < NS; ++t) {
          *reinterpret_cast<u32x4*>(xw + (q * NS + t) * 1024) = yq[ig][q][t];
          yo[q][t] = yq[1 - ig][q][t];
        }
    };
    if (wn == 0) give(std::integral_constant<int, 1>{}); else give(std::integral_constant<int, 0>{});
    ODT_BARRIER_LDS();
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int t = 0; t < NS; ++t) yr[q][t] = *reinterpret_cast<const u32x4*>(xr + (q * NS + t) * 1024);
    ODT_BARRIER_LDS();
  }
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.f_wt, 0, (int)((unsigned)nch * (unsigned)WCH), 0x00020000);
  // chunk rotation: the workgroups of a launch reach this phase together and walk the output columns at the same pace --
  // without it every residual fetch and store in flight on the chip addresses the SAME 128-byte column of the pixels' rows,
  // i.e. the same few HBM channels (the 1x1 kernels' K-slice rotation, conv_h2.hip, for the same reason).  Workgroup mt
  // starts at column chunk mt mod nch and wraps; `c` below counts the steps, ce(c) is the chunk they work on.
  const int rot = (p.debug & 0x100) == 0 ? (m0 / G::BM) % nch : 0;
  auto ce = [&](int c) { return c < nch ? (c + rot >= nch ? c + rot - nch : c + rot) : c; };
  auto dma_w = [&](int cs, int buf) {
    const int c = ce(cs);          // (a chunk past the image: out-of-range offsets, zeros -- every issue count below is static)
#pragma unroll
    for (int i = 0; i < WCH / 8192; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, ODT_LDS_PTR(lds + buf * WCH + (i * 8 + wave) * 1024), 16,
                                               c < nch ? lane * 16 + (i * 8 + wave) * 1024 : (int)kOOB, c < nch ? c * WCH : 0, 0, 0);
  };
  dma_w(0, 0);
  dma_w(1, 1);

  // ---- 4. the 1x1 conv, 32 output columns at a time
  const unsigned mrows = (unsigned)M;
  const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.f_res != nullptr ? p.f_res : p.f_bias), 0, (int)(p.f_res != nullptr ? mrows * (unsigned)p.f_res_ldc * 4u : 0u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)p.f_out, 0, (int)(mrows * (unsigned)p.f_out_ldc * 4u), 0x00020000);
  const int c4 = tid & 7, row0 = tid >> 3;
  // rows row0 + 64 s2 of the tile: byte offsets of the thread's 16-byte chunk in the residual / output rows
  const unsigned m_first = (unsigned)(m0 + row0);
  const unsigned roff0 = (m_first * (unsigned)p.f_res_ldc + c4 * 4u) * 4u, rstep = 64u * (unsigned)p.f_res_ldc * 4u;
  const unsigned ooff0 = (m_first * (unsigned)p.f_out_ldc + c4 * 4u) * 4u, ostep = 64u * (unsigned)p.f_out_ldc * 4u;
  const bool has_res = p.f_res != nullptr;
  const float act_lo = p.f_relu == 1 ? 0.f : -__builtin_huge_valf();
  const float* k3 = reinterpret_cast<const float*>(lds + G::F_K3OFF);      // [0] 2^-t_n, [1] bias_n of the 1x1 conv (prologue)
  auto fetch_res = [&](int cs, int s2) -> f32x4 {
    const int c = ce(cs);
    // (read once, by this workgroup only: non-temporal, like the unfused epilogue's residual chunks -- plain fetches, or
    // non-temporal stores of the result, measured 0.3 - 0.7 % slower; past the last chunk or without a residual: out of range, zeros)
    const unsigned off = has_res && c < nch && m_first + 64u * s2 < mrows ? roff0 + s2 * rstep : kOOB;
    return (f32x4)__builtin_amdgcn_raw_buffer_load_b128(rs_res, (int)off, c * 128, 2);
  };
  // residual chunks in flight: ra = chunk c - 1 (consumed under chunk c's MFMAs, each register refilled with chunk c + 1's
  // right behind its use), rb = chunk c
  f32x4 ra[4], rb[4];
#pragma unroll
  for (int s2 = 0; s2 < 4; ++s2) ra[s2] = fetch_res(0, s2);
#pragma unroll
  for (int s2 = 0; s2 < 4; ++s2) rb[s2] = fetch_res(1, s2);
  const unsigned char* wrd = lds + lane * 16;
  float* Cst = reinterpret_cast<float*>(lds + COFF);
  const int cw_at = (wm * 64 + wn * 32 + fr) * CS + 4 * fg;     // this lane's pixel row in the result tile
  const int cr_at = row0 * CS + c4 * 4;
  float vmax = 0.f;
  // one row-phase item: rows row0 + 64 s2 of chunk c's result tile -> global; the residual register is refilled for chunk c + 2
  auto row_item = [&](int cs, int s2, f32x4& rr) {
    const int c = ce(cs);
    const float* Cb = Cst + (cs & 1) * (CBUF / 4);
    f32x4 v = *reinterpret_cast<const f32x4*>(Cb + cr_at + 64 * s2 * CS);
    const f32x4 sc3 = *reinterpret_cast<const f32x4*>(k3 + c * 32 + c4 * 4), b3 = *reinterpret_cast<const f32x4*>(k3 + 1024 + c * 32 + c4 * 4);
    v = v * sc3;
    v += b3;
    v += rr;                                // (no residual: the descriptor is empty, the chunks read as zeros)
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], act_lo);
    const float vm = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
    const bool ok = m_first + 64u * s2 < mrows;
    vmax = fmaxf(vmax, ok ? vm : 0.f);
    __builtin_amdgcn_raw_buffer_store_b128((u32x4)v, rs_out, (int)(ok ? ooff0 + s2 * ostep : kOOB), c * 128, 0);
    rr = fetch_res(cs + 2, s2);
  };
  ODT_WAIT_VM_LGKM0(8);                     // (own pieces of chunks 0 / 1 have landed; the residual fetches may fly)
  __builtin_amdgcn_s_barrier();
  // chunk c: MFMAs (own half: steps wn NS + t, received half: (1 - wn) NS + t of the image) with the row phase of chunk
  // c - 1 under the first steps; result -> LDS tile c & 1; barrier; weight DMA of chunk c + 2 into the buffer just left
  auto chunk = [&](int c, f32x4 (&rprev)[4], auto HP) {
    constexpr bool has_prev = decltype(HP)::value;
    const int buf = c & 1;
    const unsigned char* wo = wrd + buf * WCH + wn * NS * 1024;
    const unsigned char* wr = wrd + buf * WCH + (1 - wn) * NS * 1024;
    f32x16 co;
#pragma unroll
    for (int r = 0; r < 16; ++r) co[r] = 0.f;
    // half-steps u = 2 t + (0: own K half, 1: received half): the next half-step's weight fragments (hi, lo) are read under
    // this one's three MFMAs; the previous chunk's row items go behind half-steps 1, 3, 5, 7 (TN = 1: behind each of the four)
    f16x8 wf[2][2];
    wf[0][0] = *reinterpret_cast<const f16x8*>(wo);
    wf[0][1] = *reinterpret_cast<const f16x8*>(wo + 2 * NS * 1024);
#pragma unroll
    for (int u = 0; u < 2 * NS; ++u) {
      if (u + 1 < 2 * NS) {
        const unsigned char* w = ((u + 1) & 1 ? wr : wo) + ((u + 1) >> 1) * 1024;
        wf[(u + 1) & 1][0] = *reinterpret_cast<const f16x8*>(w);
        wf[(u + 1) & 1][1] = *reinterpret_cast<const f16x8*>(w + 2 * NS * 1024);
      }
      f16x8 yhh, yll;
      if (u & 1) { __builtin_memcpy(&yhh, &yr[0][u >> 1], 16); __builtin_memcpy(&yll, &yr[1][u >> 1], 16); }
      else { __builtin_memcpy(&yhh, &yo[0][u >> 1], 16); __builtin_memcpy(&yll, &yo[1][u >> 1], 16); }
      ODT_FENCE();
      co = ODT_MFMA_F16(wf[u & 1][1], yhh, co);
      co = ODT_MFMA_F16(wf[u & 1][0], yll, co);
      co = ODT_MFMA_F16(wf[u & 1][0], yhh, co);
      ODT_FENCE();
      if constexpr (has_prev) {
        if constexpr (NS >= 4) { if ((u & 1) && u < 8) row_item(c - 1, u >> 1, rprev[u >> 1]); }
        else row_item(c - 1, u, rprev[u]);            // (64-wide producer: four half-steps, one row item behind each)
      }
    }
    float* Cb = Cst + buf * (CBUF / 4) + cw_at;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 a = {co[4 * g], co[4 * g + 1], co[4 * g + 2], co[4 * g + 3]};
      a = a * yinv_k;
      *reinterpret_cast<f32x4*>(Cb + 8 * g) = a;
    }
    // own pieces of chunk c + 1's weights have landed (issued a chunk ago; the row phase's 4 stores + 4 fetches may fly)
    if constexpr (has_prev) ODT_WAIT_VM_LGKM0(8); else ODT_WAIT_VM_LGKM0(0);
    __builtin_amdgcn_s_barrier();
    ODT_FENCE();
    dma_w(c + 2, buf);
    ODT_FENCE();
  };
  chunk(0, ra, std::false_type{});
  int c = 1;
#pragma unroll 1
  for (; c + 1 < nch; c += 2) {
    chunk(c, ra, std::true_type{});         // consumes chunk c - 1's residual (ra), refills ra with chunk c + 1's
    chunk(c + 1, rb, std::true_type{});
  }
  if (c < nch) {
    chunk(c, ra, std::true_type{});
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) row_item(c, s2, rb[s2]);
  } else {
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) row_item(c - 1, s2, ra[s2]);
  }
  publish_amax_wg<512>(p.f_out_amax, vmax, tid, lds);
}

template <int TN, bool TRACE = false, bool FUSE = false, int WN = 2>
__global__ void __launch_bounds__(512, 2) conv_h2k_kernel(const ConvParams* __restrict__ pp) {
  using G = H2kCfg<TN, FUSE, WN>;
  constexpr int WM = G::WM, KW = 3;
  constexpr int BM = G::BM, BN = G::BN, AKG = G::AKG, APL = G::APL, ABUF = G::ABUF, BKG = G::BKG, BPL = G::BPL;
  constexpr int STAGE_B = G::STAGE_B, BOFF = G::BOFF, NW = G::NW, ZR = G::ZR, RA = G::RA;
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
  // split-K (layers of few rows: res3 / res4 conv2 at b = 1): consecutive workgroups are the ranges of one tile's (slice, kh)
  // groups; each writes its raw partial tile, split_reduce_kernel adds them in range order
  const int splitk = p.splitk > 1 ? p.splitk : 1;
  const int ks = wg % splitk;
  wg /= splitk;
  const int mt = wg / ntn, nt = wg - mt * ntn;
  const int m0 = mt * BM, n0 = nt * BN;
  const int HoWo = p.Ho * p.Wo;
  const int M = p.B * HoWo;
  const int cpt = p.Cin >> 5;
  const int nsteps = p.kh * KW * cpt, ngroups_all = p.kh * cpt;
  const int g_begin = (int)(((long)ngroups_all * ks) / splitk);
  const int ngroups = (int)(((long)ngroups_all * (ks + 1)) / splitk) - g_begin;
  const int halo = (KW - 1) * p.dil;
  const int sexp = h2_in_scale_exp(p);
  const float a_scale = pow2f(sexp), h2_inv = pow2f(-sexp);

  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(
      (void*)p.in, 0, (int)((unsigned)p.B * p.in_Ha * p.in_Wa * p.in_ldc * 4u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_wt = __builtin_amdgcn_make_buffer_rsrc(
      (void*)p.wt_split, 0, (int)((unsigned)ntn * nsteps * (unsigned)STAGE_B), 0x00020000);

  unsigned l_b = ((unsigned)nt * (unsigned)nsteps + (unsigned)(g_begin * KW)) * (unsigned)STAGE_B;
  auto dma_b = [&](int boff) {
#pragma unroll
    for (int i = 0; i < NW; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_wt, ODT_LDS_PTR(lds + boff + (i * 8 + wave) * 1024), 16,
                                               lane * 16 + (i * 8 + wave) * 1024, (int)l_b, 0, 0);
    l_b += (unsigned)STAGE_B;
  };
  dma_b(BOFF);                               // stage 0's weights: requested before the address set-up below

  // ---- the tile's two runs of input pixels (tap (kh, 0) of row r: run0 for r < len0, run1 behind it)
  const int pix_bytes = p.in_ldc * 4;
  const int n_first = sfast_div(m0, p.div_howo_mul, p.div_howo_sh), r_img = m0 - n_first * HoWo;
  const int len0 = HoWo - r_img < BM ? HoWo - r_img : BM;
  const int pix0 = (n_first * p.in_Ha - p.pad_t) * p.in_Wa + r_img - p.pad_l;          // (pitch == Wo: r_img = ho * Wo + wo)
  const int pix1 = ((n_first + 1) * p.in_Ha - p.pad_t) * p.in_Wa - p.pad_l;
  // loader: thread -> stage row (t >> 3) + 64 j, 16-byte column t & 7 (eight lanes: the 128 bytes of a row's 32 channels)
  const int a_c = tid & 7, a_r = tid >> 3;
  int a_base[RA];
#pragma unroll
  for (int j = 0; j < RA; ++j) {
    const int pr = a_r + 64 * j;
    const int pix = pr < len0 + halo ? pix0 + pr : pix1 + (pr - len0 - halo);
    a_base[j] = pr < BM + 2 * halo ? pix * pix_bytes + a_c * 16 : (int)kOOB;
  }
  int l_cs = g_begin / p.kh, l_kh = g_begin - (g_begin / p.kh) * p.kh;      // next group to fetch (groups: kh innermost)
  f32x4 ga[RA];
  auto load_group = [&]() {
    const int khoff = l_kh * p.dil * p.in_Wa * pix_bytes;
#pragma unroll
    for (int j = 0; j < RA; ++j) {
      const unsigned v = (unsigned)a_base[j] == kOOB ? kOOB : (unsigned)(a_base[j] + khoff);
      ga[j] = (f32x4)__builtin_amdgcn_raw_buffer_load_b128(rs_in, (int)v, l_cs * 128, 0);
    }
    if (++l_kh == p.kh) { l_kh = 0; ++l_cs; }
  };
  auto store_slot = [&](int abuf, int j) {
    const int pr = a_r + 64 * j;
    if (pr < BM + 2 * halo) {
      unsigned h0, l0, h1, l1;
      split2h(ga[j][0], ga[j][1], a_scale, h0, l0);
      split2h(ga[j][2], ga[j][3], a_scale, h1, l1);
      unsigned char* d = lds + abuf + (a_c >> 1) * AKG + pr * 16 + (a_c & 1) * 8;
      *reinterpret_cast<u32x2*>(d) = u32x2{h0, h1};
      *reinterpret_cast<u32x2*>(d + APL) = u32x2{l0, l1};
    }
  };
  // the zero rows of both A buffers (never overwritten: stage rows stop at BM + 2 halo <= ZR)
  if (tid < 16) {
    const int b = tid >> 3, q = (tid >> 2) & 1, kg = tid & 3;
    *reinterpret_cast<u32x4*>(lds + b * ABUF + q * APL + kg * AKG + ZR * 16) = u32x4{0u, 0u, 0u, 0u};
  }

  // ---- fragment rows of this lane: stage row of (row, tap kw = 0) and the 9-bit tap validity
  const int fr = lane & 31, fg = lane >> 5;
  int fa_base[2];
  unsigned fa_mask[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int row = wm * 64 + t * 32 + fr, m = m0 + row;
    const bool ok = m < M;
    const int mm = ok ? m : 0;
    const int n = sfast_div(mm, p.div_howo_mul, p.div_howo_sh), r = mm - n * HoWo;
    const int ho = sfast_div(r, p.div_wo_mul, p.div_wo_sh), wo = r - ho * p.Wo;
    unsigned mk = 0;
    for (int khh = 0; khh < p.kh; ++khh)
      for (int kww = 0; kww < KW; ++kww) {
        const int hi = ho - p.pad_t + khh * p.dil, wi = wo - p.pad_l + kww * p.dil;
        if (ok && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W) mk |= 1u << (khh * KW + kww);
      }
    fa_mask[t] = mk;
    fa_base[t] = fg * AKG + (row < len0 ? row : row + halo) * 16;
  }
  const int fa_zero = fg * AKG + ZR * 16;
  const int b_rd = fg * BKG + (wn * TN * 32 + fr) * 16;

  f32x16 acc[2][TN];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if constexpr (FUSE) {
    // the fused tail's view of this conv's epilogue constants: [0] 2^-s 2^-t_c, [1] bias_c (visible behind the prologue's barrier)
    if (tid < 128) {
      const int q = tid & 63;
      const __amdgpu_buffer_rsrc_t rs_ch = __builtin_amdgcn_make_buffer_rsrc((void*)p.h2_chinv, 0, (int)((unsigned)cout_padded(p.Cout) * 4u), 0x00020000);
      const __amdgpu_buffer_rsrc_t rs_bias = __builtin_amdgcn_make_buffer_rsrc((void*)p.bias, 0, (int)((unsigned)p.Cout * 4u), 0x00020000);
      f32x4 v = tid < 64 ? (f32x4)__builtin_amdgcn_raw_buffer_load_b128(rs_ch, q * 16, 0, 0) * h2_inv
                         : (f32x4)__builtin_amdgcn_raw_buffer_load_b128(rs_bias, q * 16, 0, 0);
      *reinterpret_cast<f32x4*>(lds + G::F_KOFF + (tid >> 6) * 1024 + q * 16) = v;
    }
    {   // ... and the fused conv's: [0] 2^-t_n, [1] bias_n (f_cout <= 1024: launch_conv_h2)
      const int q = tid & 255;
      const __amdgpu_buffer_rsrc_t rs_k3 = __builtin_amdgcn_make_buffer_rsrc((void*)(tid < 256 ? p.f_chinv : p.f_bias), 0, (int)((unsigned)p.f_cout * 4u), 0x00020000);
      const f32x4 v = (f32x4)__builtin_amdgcn_raw_buffer_load_b128(rs_k3, q * 16, 0, 0);
      *reinterpret_cast<f32x4*>(lds + G::F_K3OFF + (tid >> 8) * 4096 + q * 16) = v;
    }
  }
  // ---- prologue: group 0 staged, B stage 0 landed, stage 1's DMA in flight behind the barrier
  load_group();
  stamp(6);
#pragma unroll
  for (int j = 0; j < RA; ++j) store_slot(0, j);
  ODT_WAIT_VM_LGKM0(0);
  __builtin_amdgcn_s_barrier();
  if (nsteps > 1) dma_b(BOFF + STAGE_B);
  stamp(7); stamp(1);

  f16x8 fa[2][2][2], fb[2][2];
  int fa_addr[2];                            // this stage's fragment addresses (A buffer + row + tap, or the zero row)
  int c_kh = g_begin - (g_begin / p.kh) * p.kh;      // kh of the group being computed
  auto tap_addr = [&](int abuf, int khh, int kww) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
      fa_addr[t] = abuf + (((fa_mask[t] >> (khh * KW + kww)) & 1u) ? fa_base[t] + kww * p.dil * 16 : fa_zero);
  };
  int fa_addr_n[2];                          // ... of the stage behind the barrier
  auto tap_addr_n = [&](int abuf, int khh, int kww) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
      fa_addr_n[t] = abuf + (((fa_mask[t] >> (khh * KW + kww)) & 1u) ? fa_base[t] + kww * p.dil * 16 : fa_zero);
  };
  auto rdA = [&](int kst, int q) {
#pragma unroll
    for (int t = 0; t < 2; ++t) fa[kst][q][t] = *reinterpret_cast<const f16x8*>(lds + q * APL + kst * 2 * AKG + fa_addr[t]);
  };
  auto rdA_n = [&](int q) {
#pragma unroll
    for (int t = 0; t < 2; ++t) fa[0][q][t] = *reinterpret_cast<const f16x8*>(lds + q * APL + fa_addr_n[t]);
  };
  auto rdB = [&](int bbuf, int kst, int j, int dst) {
#pragma unroll
    for (int q = 0; q < 2; ++q) fb[dst][q] = *reinterpret_cast<const f16x8*>(lds + bbuf + q * BPL + kst * 2 * BKG + b_rd + j * 512);
  };
  int a_cur = 0, a_nxt = ABUF;
  int b_cur = BOFF, b_nxt = BOFF + STAGE_B;
  tap_addr(a_cur, c_kh, 0);
  rdA(0, 1); rdA(0, 0);
  rdB(b_cur, 0, 0, 0);

  // One stage = tap kw = KWI of the current group.  NEXT / PRE as in conv_h2_kernel (stage c+1 / c+2 exist); GN: a next
  // group exists (fetch it in the first stage, split + store it in the third)
  auto step = [&](auto KWIC, auto NEXT, auto PRE, auto GNC) {
    constexpr int KWI = decltype(KWIC)::value;
    constexpr bool next = decltype(NEXT)::value, pre = decltype(PRE)::value, gn = decltype(GNC)::value;
    constexpr int NG = 2 * TN;
    ODT_FENCE();
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int kst = g / TN, j = g % TN, bsel = g & 1;
      const bool last = g == NG - 1;
      if (last) {
        // (first stage of a group: the group fetch issued in it may stay in flight; later stages: the fetch is older than
        // the DMA this wait is about, so everything has landed)
        if constexpr (KWI == 0 && gn) ODT_WAIT_VM_LGKM0(RA); else ODT_WAIT_VM_LGKM0(0);
        __builtin_amdgcn_s_barrier();
        ODT_FENCE();
        if constexpr (pre) dma_b(b_cur);
        // fragment addresses of the next stage: next tap of this group, or tap 0 of the next group's buffer
        if constexpr (next) {
          if constexpr (KWI + 1 < KW) tap_addr_n(a_cur, c_kh, KWI + 1);
          else tap_addr_n(a_nxt, c_kh + 1 == p.kh ? 0 : c_kh + 1, 0);
          rdB(b_nxt, 0, 0, bsel ^ 1);
        }
      } else {
        rdB(b_cur, (g + 1) / TN, (g + 1) % TN, bsel ^ 1);
        if (TN == 1) { rdA(1, 1); rdA(1, 0); }
      }
      ODT_FENCE();
      ODT_MF(kst, 1, 0, j, bsel); ODT_FENCE();             // lo * hi
      if (last) {
        if constexpr (next) rdA_n(1);
      } else {
        if constexpr (gn && KWI == 2) {
          // the next group's run: registers -> LDS, in the group's third stage, two stages behind the fetch
          if (TN > 1) {
            constexpr int SPG = (RA + NG - 2) / (NG - 1);       // slots per column group (the last group sits behind the barrier)
#pragma unroll
            for (int q = 0; q < SPG; ++q)
              if (g * SPG + q < RA) store_slot(a_nxt, g * SPG + q);
          } else { store_slot(a_nxt, 0); store_slot(a_nxt, 1); store_slot(a_nxt, 2); }
        }
        if constexpr (gn && KWI == 0) { if (g == (TN == 1 ? 0 : 1)) load_group(); }
      }
      ODT_FENCE();
      ODT_MF(kst, 0, 1, j, bsel); ODT_FENCE();             // hi * lo
      if (last) {
        if constexpr (next) rdA_n(0);
      } else {
        if (TN > 1 && g == TN - 2) rdA(1, 1);
        if (TN > 1 && g == TN - 1) rdA(1, 0);
        if constexpr (gn && KWI == 2) { if (TN == 1) { store_slot(a_nxt, 3); store_slot(a_nxt, 4); } }
      }
      ODT_FENCE();
      ODT_MF(kst, 0, 0, j, bsel); ODT_FENCE();             // hi * hi
    }
    fa_addr[0] = fa_addr_n[0]; fa_addr[1] = fa_addr_n[1];
    { const int t = b_cur; b_cur = b_nxt; b_nxt = t; }
    if constexpr (KWI == KW - 1) {
      const int u = a_cur; a_cur = a_nxt; a_nxt = u;
      if (++c_kh == p.kh) c_kh = 0;
    }
  };
  {
    using T = std::true_type; using F = std::false_type;
    using K0 = std::integral_constant<int, 0>; using K1 = std::integral_constant<int, 1>; using K2 = std::integral_constant<int, 2>;
    for (int g = 0; g + 1 < ngroups; ++g) { step(K0{}, T{}, T{}, T{}); step(K1{}, T{}, T{}, T{}); step(K2{}, T{}, T{}, T{}); }
    step(K0{}, T{}, T{}, F{});
    step(K1{}, T{}, F{}, F{});
    step(K2{}, F{}, F{}, F{});
  }
#undef ODT_MF
  stamp(2);
  if constexpr (FUSE) h2f_tail<TN, TRACE>(p, acc, lds, m0, M, wave, wm, wn, h2_inv);
  else split3_epilogue<WM, WN, TN, G::LDS, TRACE>(p, acc, lds, m0, n0, M, HoWo, ks, splitk, tid, wm, wn, fr, fg, h2_inv);
  stamp(5);
}
#undef ODT_FENCE

}  // namespace

void launch_conv_h2k(const ConvParams& p, const ConvParams* dev, unsigned grid, hipStream_t stream) {
  const int bn = p.wt_split_bn;
  if (p.f_wt != nullptr) {                   // fused 1x1 tail (launch_conv_h2 checked the shape)
    if (bn == 64) hipLaunchKernelGGL((conv_h2k_kernel<1, false, true>), dim3(grid), dim3(512), 0, stream, dev);
    else if (bn == 128) hipLaunchKernelGGL((conv_h2k_kernel<2, false, true>), dim3(grid), dim3(512), 0, stream, dev);
    else if (p.trace != nullptr) hipLaunchKernelGGL((conv_h2k_kernel<4, true, true>), dim3(grid), dim3(512), 0, stream, dev);
    else hipLaunchKernelGGL((conv_h2k_kernel<4, false, true>), dim3(grid), dim3(512), 0, stream, dev);
  } else if (bn == 256) {
    if (p.trace != nullptr) hipLaunchKernelGGL((conv_h2k_kernel<4, true>), dim3(grid), dim3(512), 0, stream, dev);
    else hipLaunchKernelGGL((conv_h2k_kernel<4, false>), dim3(grid), dim3(512), 0, stream, dev);
  } else if (bn == 128) {
    hipLaunchKernelGGL((conv_h2k_kernel<2, false>), dim3(grid), dim3(512), 0, stream, dev);
  } else if (p.wt_split_bm == 512) {          // 64-wide layer, eight waves stacked along M: 512 x 64 tiles
    hipLaunchKernelGGL((conv_h2k_kernel<2, false, false, 1>), dim3(grid), dim3(512), 0, stream, dev);
  } else {
    hipLaunchKernelGGL((conv_h2k_kernel<1, false>), dim3(grid), dim3(512), 0, stream, dev);
  }
}

}  // namespace odt
