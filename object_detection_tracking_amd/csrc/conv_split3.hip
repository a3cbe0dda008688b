// ---------------------------------------------------------------------------------------------------------
// conv_split3_kernel: the round-2 loop structure.  Same arithmetic (bf16x3, six piece products), but
//   * 8 waves / 512 threads, ONE workgroup per CU, 256-row tiles: a weight stage is fetched once per 256
//     output rows (half the L2 -> LDS weight traffic of the 128-row tiles);
//   * BK = 16 per LDS stage, a ring of THREE stages, ONE barrier per stage;
//   * weights: the pre-split image goes global -> LDS by LDS-DMA (buffer_load ... lds), two stages ahead --
//     no prefetch registers, no ds_write; the pipeline never drains (counted vmcnt, raw s_barrier);
//   * activations: f32 -> registers (one stage ahead) -> split -> LDS behind the MFMAs of the first
//     column group; the last column group's operands are read before the barrier and its MFMAs cover
//     the first fragment reads of the next stage;
//   * K order = (16-channel slice, kh, kw): the nine taps of a channel slice are consecutive stages, so
//     a 3x3 conv re-reads its activation lines from L1 / L2 instead of the fabric (tap-major order
//     streamed the whole tile's input through L2 once per tap);
//   * residual (same shape / nearest-2x) added in the epilogue from 16-byte row chunks that are fetched
//     while the C tile is staged through LDS (the MFMA-layout accumulator preload of the one-stage
//     kernel cost 128 four-byte loads per lane: 9 us of a 63 us res4 conv3 tile).
// Tile configurations <WM, WN, TN> (WM x WN = 8 waves, wave tile 64 x 32 TN):
//   <4,2,4> 256 x 256 (Cout % 256 == 0)   <4,2,2> 256 x 128 (Cout % 128 == 0)   <4,2,1> 256 x 64 (Cout % 64 == 0)
#include "conv_split_epilogue.hpp"

namespace odt {

namespace {

template <int WM, int WN, int TN>
struct Split3Cfg {
  static constexpr int BM = 64 * WM, BN = 32 * TN * WN;
  static constexpr int AKG = BM * 16 + 64, APL = 2 * AKG;   // A piece plane: [k-group 2][row][8 bf16], 64-B pad per k-group
  static constexpr int BKG = BN * 16, BPL = 2 * BKG;        // B: the linear image the DMA writes
  static constexpr int STAGE_B = 3 * BPL;                   // bytes of weight image per stage
  static constexpr int STAGE = 3 * APL + STAGE_B;
  static constexpr int LDS = 3 * STAGE;
  static constexpr int NCHUNK = STAGE_B / 1024;             // 1-KB DMA pieces (one wave instruction each) per stage
  static constexpr int RA = BM / 128;                       // A rows (16-byte loads) per thread and stage
  static_assert(LDS <= 160 * 1024, "LDS ring");
};

template <int WM, int WN, int TN, bool TRACE = false>
__global__ void __launch_bounds__(512, 2) conv_split3_kernel(const ConvParams* __restrict__ pp) {
  using G = Split3Cfg<WM, WN, TN>;
  constexpr int BM = G::BM, BN = G::BN, AKG = G::AKG, APL = G::APL, BKG = G::BKG, BPL = G::BPL;
  constexpr int STAGE = G::STAGE, STAGE_B = G::STAGE_B, NCHUNK = G::NCHUNK, RA = G::RA;
  static_assert(WM * WN == 8, "8 waves");
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
  const int cpt = p.Cin >> 4;                                // 16-channel slices of the first source
  const int cpt2 = p.in2 != nullptr ? p.Cin2 >> 4 : 0;       // ... of the K-concatenated second source (1x1 only)
  const int nsteps1 = ntaps * cpt, nsteps_all = nsteps1 + cpt2;
  // this workgroup's stages [s_begin, s_begin + nsteps): split-K ranges are balanced to within one stage
  const int s_begin = (int)(((long)nsteps_all * ks) / splitk);
  const int nsteps = (int)(((long)nsteps_all * (ks + 1)) / splitk) - s_begin;

  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(
      (void*)p.in, 0, (int)((unsigned)p.B * p.in_Ha * p.in_Wa * p.in_ldc * 4u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_in2 = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.in2 != nullptr ? p.in2 : p.in), 0,
      (int)(p.in2 != nullptr ? (unsigned)p.B * p.in2_Ha * p.in2_Wa * p.in2_ldc * 4u : 0u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_wt = __builtin_amdgcn_make_buffer_rsrc(
      (void*)p.wt_split, 0, (int)((unsigned)ntn * nsteps_all * (unsigned)STAGE_B), 0x00020000);

  // ---- weights: wave w, instruction i copies the 1-KB piece i * 8 + w of the stage image
  unsigned l_b = ((unsigned)nt * (unsigned)nsteps_all + (unsigned)s_begin) * (unsigned)STAGE_B;
  auto dma_b = [&](int st) {
#pragma unroll
    for (int i = 0; i < (NCHUNK + 7) / 8; ++i) {
      if ((i + 1) * 8 <= NCHUNK || i * 8 + wave < NCHUNK)    // wave-uniform
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_wt, ODT_LDS_PTR(lds + st + 3 * APL + (i * 8 + wave) * 1024), 16,
                                                 lane * 16 + (i * 8 + wave) * 1024, (int)l_b, 0, 0);
    }
    l_b += (unsigned)STAGE_B;
  };
  // this wave's DMA instructions per stage: NW or NW - 1 (the counted wait in front of a barrier needs the exact
  // number).  wait_stage: this wave's share of the stage that the barrier publishes has landed -- its DMA (issued
  // one stage earlier) and its LDS stores; YOUNGER: one stage's worth of A fetches + DMA was issued behind that
  // DMA and may stay in flight.
  constexpr int NW = (NCHUNK + 7) / 8;
  const bool dma_full = (NCHUNK % 8) == 0 || wave < (NCHUNK % 8);
  auto wait_stage = [&](auto YOUNGER) {
    if constexpr (decltype(YOUNGER)::value) {
      if (dma_full) ODT_WAIT_VM_LGKM0(RA + NW); else ODT_WAIT_VM_LGKM0(RA + NW - 1);
    } else {
      ODT_WAIT_VM_LGKM0(0);
    }
  };
  dma_b(0);
  if (nsteps > 1) dma_b(STAGE);

  // ---- activations: thread -> (row (t >> 2) + 128 j, 16-byte column t & 3): four lanes cover the 64 contiguous
  // bytes (16 channels) of a row's stage.  Per row: the byte offset of the tap-(0,0) input pixel and a bit per tap
  // (inside the image and m < M); a stage's offset is base + tap offset, or out of range (the load returns zeros).
  const int a_c = tid & 3, a_r = tid >> 2;
  const unsigned pix_bytes = (unsigned)p.in_ldc * 4u;
  int a_base[RA];
  unsigned a_mask[RA];
  const bool dense_in = ntaps == 1 && p.stride == 1 && p.pad_t == 0 && p.pad_l == 0 &&
                        p.H == p.in_Ha && p.W == p.in_Wa && p.Ho == p.H && p.Wo == p.W;
#pragma unroll
  for (int j = 0; j < RA; ++j) {
    const int m = m0 + a_r + 128 * j;
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
  // load stream position: (16-channel slice, tap) with the tap innermost; then the second source's slices
  // (a split-K range starts inside the first source: launch_conv_split keeps split-K off for second-source convs)
  int l_cs = s_begin / ntaps, l_tap = s_begin - l_cs * ntaps;
  int l_kh = l_tap / p.kw, l_kw = l_tap - l_kh * p.kw;
  bool l_src2 = false;
  unsigned a_row[RA];
  auto set_rows = [&]() {
    const unsigned tapoff = (unsigned)(l_kh * p.dil * p.in_Wa + l_kw * p.dil) * pix_bytes;
#pragma unroll
    for (int j = 0; j < RA; ++j) a_row[j] = ((a_mask[j] >> l_tap) & 1u) ? (unsigned)a_base[j] + tapoff : kOOB;
  };
  auto set_src2 = [&]() {
#pragma unroll
    for (int j = 0; j < RA; ++j) {
      const int m = m0 + a_r + 128 * j;
      const bool ok = m < M;
      const int mm = ok ? m : 0;
      const int n = sfast_div(mm, p.div_howo_mul, p.div_howo_sh), r = mm - n * HoWo;
      const int ho = sfast_div(r, p.div_wo_mul, p.div_wo_sh), wo = r - ho * p.Wo;
      const unsigned pix = ((unsigned)n * p.in2_Ha + (unsigned)(ho * p.in2_stride)) * p.in2_Wa + (unsigned)(wo * p.in2_stride);
      a_row[j] = ok ? pix * (unsigned)p.in2_ldc * 4u + a_c * 16u : kOOB;
    }
  };
  set_rows();
  f32x4 ga[RA];
  const bool a_nt = (p.debug & 0x800) != 0 && ntn == 1 && ntaps == 1;     // single-use activations: non-temporal hint (see conv_h2.hip)
  auto load_a = [&]() {
#pragma unroll
    for (int j = 0; j < RA; ++j)
      ga[j] = a_nt ? (f32x4)__builtin_amdgcn_raw_buffer_load_b128(l_src2 ? rs_in2 : rs_in, (int)a_row[j], l_cs * 64, 2)
                   : (f32x4)__builtin_amdgcn_raw_buffer_load_b128(l_src2 ? rs_in2 : rs_in, (int)a_row[j], l_cs * 64, 0);
    // advance
    if (l_src2) {
      ++l_cs;
    } else if (ntaps == 1) {
      if (++l_cs == cpt && cpt2 > 0) { l_cs = 0; l_src2 = true; set_src2(); }
    } else {
      ++l_tap;
      if (++l_kw == p.kw) { l_kw = 0; ++l_kh; }
      if (l_tap == ntaps) { l_tap = 0; l_kh = 0; l_kw = 0; ++l_cs; }
      set_rows();            // (harmless past the last stage: never loaded)
    }
  };
  auto store_a = [&](int st) {
#pragma unroll
    for (int j = 0; j < RA; ++j) {
      unsigned h0, m0_, l0, h1, m1, l1;
      split2(ga[j][0], ga[j][1], h0, m0_, l0);
      split2(ga[j][2], ga[j][3], h1, m1, l1);
      unsigned char* d = lds + st + (a_c >> 1) * AKG + (a_r + 128 * j) * 16 + (a_c & 1) * 8;
      *reinterpret_cast<u32x2*>(d + 0 * APL) = u32x2{h0, h1};
      *reinterpret_cast<u32x2*>(d + 1 * APL) = u32x2{m0_, m1};
      *reinterpret_cast<u32x2*>(d + 2 * APL) = u32x2{l0, l1};
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
  // ---- prologue: stage 0 complete, stage 1's weights in flight, A of stage 1 in registers
  load_a();
  stamp(6);
  store_a(0);
  if (nsteps > 1) load_a();
  if (nsteps > 1) wait_stage(std::true_type{}); else wait_stage(std::false_type{});
  __builtin_amdgcn_s_barrier();
  stamp(7); stamp(1);

  bf16x8 fa[3][2], fb[3];
  const int a_rd = fg * AKG + (wm * 64 + fr) * 16;
  const int b_rd = 3 * APL + fg * BKG + (wn * TN * 32 + fr) * 16;
  auto rdA = [&](int st, int q) {
#pragma unroll
    for (int t = 0; t < 2; ++t) fa[q][t] = *reinterpret_cast<const bf16x8*>(lds + st + q * APL + a_rd + t * 512);
  };
  auto rdB = [&](int st, int q, int j) {
    fb[q] = *reinterpret_cast<const bf16x8*>(lds + st + q * BPL + b_rd + j * 512);
  };
#pragma unroll
  for (int q = 0; q < 3; ++q) rdA(0, q);
#pragma unroll
  for (int q = 0; q < 3; ++q) rdB(0, q, 0);

#define ODT_MF(qa, qb, j) { acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[qa][0], fb[qb], acc[0][j], 0, 0, 0); \
                            acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[qa][1], fb[qb], acc[1][j], 0, 0, 0); }
#define ODT_FENCE() __builtin_amdgcn_sched_barrier(0)
  int st_cur = 0, st_nxt = STAGE, st_nn = 2 * STAGE;
  // One stage.  NEXT: stage c+1 exists (its A: registers -> LDS; read its first fragments behind the last column
  // group); PRE: stage c+2 exists (fetch its A, start its weight DMA).  The K loop is peeled so that no MFMA sits
  // in a conditional arm.
  auto step = [&](auto NEXT, auto PRE) {
    constexpr bool next = decltype(NEXT)::value, pre = decltype(PRE)::value;
    ODT_FENCE();
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const bool last = j == TN - 1, first = j == 0;
      // where the side work of a stage rides: A store behind the first product of the first group, the A fetch
      // behind its fourth, the DMA issue behind the fifth product of the second group (TN = 1: all in the one
      // group, in front of the barrier)
      if (last) {
        if (TN == 1) {
          // single column group: the side work precedes the barrier
          if constexpr (next) store_a(st_nxt);
          if constexpr (pre) { load_a(); dma_b(st_nn); }
        }
        // stage nxt must be complete before its first fragment reads below: own LDS stores, own DMA of stage c+1
        // (issued one stage ago; this stage's A fetch and DMA may stay in flight), then the barrier
        wait_stage(std::integral_constant<bool, pre>{});
        __builtin_amdgcn_s_barrier();
        ODT_FENCE();
      }
      ODT_MF(2, 0, j); ODT_FENCE();
      if (last) { if constexpr (next) rdA(st_nxt, 2); }
      else if (first) { if constexpr (next) store_a(st_nxt); }
      ODT_FENCE();
      ODT_MF(1, 0, j); ODT_MF(0, 0, j); ODT_FENCE();
      if (!last) rdB(st_cur, 0, j + 1); else if constexpr (next) rdB(st_nxt, 0, 0);
      ODT_FENCE();
      ODT_MF(1, 1, j); ODT_FENCE();
      if (last) { if constexpr (next) rdA(st_nxt, 1); }
      else if (first) { if constexpr (pre) load_a(); }
      ODT_FENCE();
      ODT_MF(0, 1, j); ODT_FENCE();
      if (!last) rdB(st_cur, 1, j + 1); else if constexpr (next) rdB(st_nxt, 1, 0);
      if (!last && j == (TN > 2 ? 1 : 0)) { if constexpr (pre) dma_b(st_nn); }
      ODT_FENCE();
      ODT_MF(0, 2, j); ODT_FENCE();
      if (!last) rdB(st_cur, 2, j + 1); else if constexpr (next) { rdA(st_nxt, 0); rdB(st_nxt, 2, 0); }
      ODT_FENCE();
    }
    const int t = st_cur; st_cur = st_nxt; st_nxt = st_nn; st_nn = t;
  };
  {
    int c = 0;
    for (; c + 2 < nsteps; ++c) step(std::true_type{}, std::true_type{});
    if (c + 1 < nsteps) { step(std::true_type{}, std::false_type{}); ++c; }
    step(std::false_type{}, std::false_type{});
  }
#undef ODT_MF
#undef ODT_FENCE
  stamp(2);
  split3_epilogue<WM, WN, TN, G::LDS, TRACE>(p, acc, lds, m0, n0, M, HoWo, ks, splitk, tid, wm, wn, fr, fg);
  stamp(5);
}

// ---------------------------------------------------------------------------------------------------------
// conv_split3k_kernel: conv_split3_kernel for stride-1 KH x 3 convs whose input rows have the output's pitch
// (in_Wa == Wo: the 3x3 layers of res3 / res4, the FPN post-hoc and RPN convs) -- the three kw taps of a
// (16-channel slice, kh) group read ONE staged, once-split run of input pixels at row offsets 0, dil, 2 dil instead
// of fetching and splitting the activations per tap.  Ablation on the MI355X (A work on every third stage only,
// profiles/r02_ablate_a_third.txt): P2-level 3x3 229 -> 296 TF, res4 conv2 227 -> 253, all conv launches -7 %.
//   * the 256 output pixels of a tile are consecutive in (n, ho, wo); within an image their tap-(kh, 0) input pixels are
//     consecutive too (stride 1, equal pitch), so a group's stage is the run [first - pad_l, last - pad_l + 2 dil];
//     a tile that crosses an image boundary stages two runs back to back (capacity 256 + 2 x 2 dil rows);
//   * taps that fall outside the image (left / right / top / bottom border, rows past M) read a zero row of the
//     stage instead: a per-lane 9-bit validity mask picks the fragment address -- no masking of data;
//   * A stages: two buffers (this group / next group), B stages: the three-deep DMA ring as before; the group's
//     fetch (3 x 16 B per thread) is issued in its first stage, split + stored in the second and third.
// Same arithmetic, same K order, same weight image and epilogue as conv_split3_kernel: results are bit-identical.
template <int TN>
struct Split3kCfg {
  static constexpr int BM = 256, BN = 64 * TN;
  static constexpr int PR = 272;                             // stage rows: 256 + 2 runs x 2 dil (dil <= 2) + the zero row, padded
  static constexpr int ZR = PR - 1;                          // the zero row
  static constexpr int AKG = PR * 16 + 64, APL = 2 * AKG, ABUF = 3 * APL;
  static constexpr int BKG = BN * 16, BPL = 2 * BKG, STAGE_B = 3 * BPL;
  static constexpr int BOFF = 2 * ABUF;
  static constexpr int RING = BOFF + 3 * STAGE_B;
  static constexpr int CTILE = 128 * (BN + 4) * 4;            // two 128-row epilogue passes
  static constexpr int LDS = RING > CTILE ? RING : CTILE;
  static constexpr int NCHUNK = STAGE_B / 1024;
  static_assert(LDS <= 160 * 1024, "LDS");
};

template <int TN, bool TRACE = false>
__global__ void __launch_bounds__(512, 2) conv_split3k_kernel(const ConvParams* __restrict__ pp) {
  using G = Split3kCfg<TN>;
  constexpr int WM = 4, WN = 2, KW = 3;
  constexpr int BM = G::BM, BN = G::BN, AKG = G::AKG, APL = G::APL, ABUF = G::ABUF, BKG = G::BKG, BPL = G::BPL;
  constexpr int STAGE_B = G::STAGE_B, BOFF = G::BOFF, NCHUNK = G::NCHUNK, ZR = G::ZR;
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
  const int mt = wg / ntn, nt = wg - mt * ntn;
  const int m0 = mt * BM, n0 = nt * BN;
  const int HoWo = p.Ho * p.Wo;
  const int M = p.B * HoWo;
  const int cpt = p.Cin >> 4;
  const int nsteps = p.kh * KW * cpt, ngroups = p.kh * cpt;
  const int halo = (KW - 1) * p.dil;

  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(
      (void*)p.in, 0, (int)((unsigned)p.B * p.in_Ha * p.in_Wa * p.in_ldc * 4u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_wt = __builtin_amdgcn_make_buffer_rsrc(
      (void*)p.wt_split, 0, (int)((unsigned)ntn * nsteps * (unsigned)STAGE_B), 0x00020000);

  // ---- weights: as conv_split3_kernel
  unsigned l_b = (unsigned)nt * (unsigned)nsteps * (unsigned)STAGE_B;
  auto dma_b = [&](int boff) {
#pragma unroll
    for (int i = 0; i < (NCHUNK + 7) / 8; ++i) {
      if ((i + 1) * 8 <= NCHUNK || i * 8 + wave < NCHUNK)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_wt, ODT_LDS_PTR(lds + boff + (i * 8 + wave) * 1024), 16,
                                                 lane * 16 + (i * 8 + wave) * 1024, (int)l_b, 0, 0);
    }
    l_b += (unsigned)STAGE_B;
  };
  constexpr int NW = (NCHUNK + 7) / 8;
  const bool dma_full = (NCHUNK % 8) == 0 || wave < (NCHUNK % 8);
  constexpr int RA = 3;                      // A fetch instructions per thread and group
  // wait_stage<A, Y>: this wave's DMA of the stage the barrier publishes has landed (+ all its LDS stores); Y: this
  // stage's DMA was issued behind it and may stay in flight; A: so may the group fetch (RA loads) issued in this stage
  auto wait_stage = [&](auto AF, auto YF) {
    constexpr bool a = decltype(AF)::value, y = decltype(YF)::value;
    if constexpr (!y) {
      ODT_WAIT_VM_LGKM0(0);
    } else if constexpr (a) {
      if (dma_full) ODT_WAIT_VM_LGKM0(RA + NW); else ODT_WAIT_VM_LGKM0(RA + NW - 1);
    } else {
      if (dma_full) ODT_WAIT_VM_LGKM0(NW); else ODT_WAIT_VM_LGKM0(NW - 1 > 0 ? NW - 1 : 0);
    }
  };
  dma_b(BOFF);
  if (nsteps > 1) dma_b(BOFF + STAGE_B);

  // ---- the tile's two runs of input pixels (tap (kh, 0) of row r: run0 for r < len0, run1 behind it)
  const int pix_bytes = p.in_ldc * 4;
  const int n_first = sfast_div(m0, p.div_howo_mul, p.div_howo_sh), r_img = m0 - n_first * HoWo;
  const int len0 = HoWo - r_img < BM ? HoWo - r_img : BM;
  const int pix0 = (n_first * p.in_Ha - p.pad_t) * p.in_Wa + r_img - p.pad_l;          // (pitch == Wo: r_img = ho * Wo + wo)
  const int pix1 = ((n_first + 1) * p.in_Ha - p.pad_t) * p.in_Wa - p.pad_l;
  // loader: thread -> stage row (t >> 2) + 128 j, 16-byte column t & 3
  const int a_c = tid & 3, a_r = tid >> 2;
  int a_base[RA];
#pragma unroll
  for (int j = 0; j < RA; ++j) {
    const int pr = a_r + 128 * j;
    const int pix = pr < len0 + halo ? pix0 + pr : pix1 + (pr - len0 - halo);
    a_base[j] = pr < BM + 2 * halo ? pix * pix_bytes + a_c * 16 : (int)kOOB;
  }
  int l_cs = 0, l_kh = 0;                    // next group to fetch
  f32x4 ga[RA];
  auto load_group = [&]() {
    const int khoff = l_kh * p.dil * p.in_Wa * pix_bytes;
#pragma unroll
    for (int j = 0; j < RA; ++j) {
      const unsigned v = (unsigned)a_base[j] == kOOB ? kOOB : (unsigned)(a_base[j] + khoff);
      ga[j] = (f32x4)__builtin_amdgcn_raw_buffer_load_b128(rs_in, (int)v, l_cs * 64, 0);
    }
    if (++l_kh == p.kh) { l_kh = 0; ++l_cs; }
  };
  auto store_slot = [&](int abuf, int j) {
    const int pr = a_r + 128 * j;
    if (pr < BM + 2 * halo) {
      unsigned h0, m0_, l0, h1, m1, l1;
      split2(ga[j][0], ga[j][1], h0, m0_, l0);
      split2(ga[j][2], ga[j][3], h1, m1, l1);
      unsigned char* d = lds + abuf + (a_c >> 1) * AKG + pr * 16 + (a_c & 1) * 8;
      *reinterpret_cast<u32x2*>(d + 0 * APL) = u32x2{h0, h1};
      *reinterpret_cast<u32x2*>(d + 1 * APL) = u32x2{m0_, m1};
      *reinterpret_cast<u32x2*>(d + 2 * APL) = u32x2{l0, l1};
    }
  };
  // the zero rows of both A buffers (never overwritten: stage rows stop at 256 + 2 halo <= ZR)
  if (tid < 12) {
    const int b = tid / 6, q = (tid % 6) >> 1, kg = tid & 1;
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

  // ---- prologue: group 0 staged, B stages 0 / 1 in flight
  load_group();
  stamp(6);
#pragma unroll
  for (int j = 0; j < RA; ++j) store_slot(0, j);
  if (nsteps > 1) {
    if (dma_full) ODT_WAIT_VM_LGKM0(NW); else ODT_WAIT_VM_LGKM0(NW - 1 > 0 ? NW - 1 : 0);
  } else {
    ODT_WAIT_VM_LGKM0(0);
  }
  __builtin_amdgcn_s_barrier();
  stamp(7); stamp(1);

  bf16x8 fa[3][2], fb[3];
  int fa_addr[2];                            // this stage's fragment addresses (A buffer + row + tap, or the zero row)
  int c_kh = 0;                              // kh of the group being computed
  auto tap_addr = [&](int abuf, int khh, int kww) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
      fa_addr[t] = abuf + (((fa_mask[t] >> (khh * KW + kww)) & 1u) ? fa_base[t] + kww * p.dil * 16 : fa_zero);
  };
  auto rdA = [&](int q) {
#pragma unroll
    for (int t = 0; t < 2; ++t) fa[q][t] = *reinterpret_cast<const bf16x8*>(lds + q * APL + fa_addr[t]);
  };
  auto rdB = [&](int boff, int q, int j) {
    fb[q] = *reinterpret_cast<const bf16x8*>(lds + boff + q * BPL + b_rd + j * 512);
  };
  int a_cur = 0, a_nxt = ABUF;
  int b_cur = BOFF, b_nxt = BOFF + STAGE_B, b_nn = BOFF + 2 * STAGE_B;
  tap_addr(a_cur, 0, 0);
#pragma unroll
  for (int q = 0; q < 3; ++q) rdA(q);
#pragma unroll
  for (int q = 0; q < 3; ++q) rdB(b_cur, q, 0);

#define ODT_MF(qa, qb, j) { acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[qa][0], fb[qb], acc[0][j], 0, 0, 0); \
                            acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[qa][1], fb[qb], acc[1][j], 0, 0, 0); }
#define ODT_FENCE() __builtin_amdgcn_sched_barrier(0)
  // One stage = tap kw = KWI of the current group.  NEXT / PRE as in conv_split3_kernel (stage c+1 / c+2 exist);
  // GN: a next group exists (fetch it in the first stage, split + store it in the second and third)
  auto step = [&](auto KWIC, auto NEXT, auto PRE, auto GNC) {
    constexpr int KWI = decltype(KWIC)::value;
    constexpr bool next = decltype(NEXT)::value, pre = decltype(PRE)::value, gn = decltype(GNC)::value;
    ODT_FENCE();
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const bool last = j == TN - 1, first = j == 0;
      if (last) {
        if (TN == 1) {
          // single column group: the stage's side work precedes the barrier (fetch before the DMA: the counted wait
          // assumes that issue order)
          if constexpr (gn) {
            if constexpr (KWI == 0) load_group();
            if constexpr (KWI == 2) { store_slot(a_nxt, 0); store_slot(a_nxt, 1); store_slot(a_nxt, 2); }
          }
          if constexpr (pre) dma_b(b_nn);
        }
        wait_stage(std::integral_constant<bool, (KWI == 0 && gn)>{}, std::integral_constant<bool, pre>{});
        __builtin_amdgcn_s_barrier();
        // fragment addresses of the next stage: next tap of this group, or tap 0 of the next group's buffer
        if constexpr (next) {
          if constexpr (KWI + 1 < KW) tap_addr(a_cur, c_kh, KWI + 1);
          else tap_addr(a_nxt, c_kh + 1 == p.kh ? 0 : c_kh + 1, 0);
        }
        ODT_FENCE();
      }
      ODT_MF(2, 0, j); ODT_FENCE();
      if (last) { if constexpr (next) rdA(2); }
      else if constexpr (gn) {
        // the next group's run: registers -> LDS, all of it in the group's third stage, two stages behind the fetch
        // (the P2-level runs come from HBM / MALL: one stage of lead left the split waiting)
        if constexpr (KWI == 2) {
          if (TN > 2) { if (j < 3) store_slot(a_nxt, j); }
          else if (first) { store_slot(a_nxt, 0); store_slot(a_nxt, 1); store_slot(a_nxt, 2); }
        }
      }
      ODT_FENCE();
      ODT_MF(1, 0, j); ODT_MF(0, 0, j); ODT_FENCE();
      if (!last) rdB(b_cur, 0, j + 1); else if constexpr (next) rdB(b_nxt, 0, 0);
      ODT_FENCE();
      ODT_MF(1, 1, j); ODT_FENCE();
      if (last) { if constexpr (next) rdA(1); }
      else if (first) { if constexpr (gn && KWI == 0) load_group(); }
      ODT_FENCE();
      ODT_MF(0, 1, j); ODT_FENCE();
      if (!last) rdB(b_cur, 1, j + 1); else if constexpr (next) rdB(b_nxt, 1, 0);
      if (!last && j == (TN > 2 ? 1 : 0)) { if constexpr (pre) dma_b(b_nn); }
      ODT_FENCE();
      ODT_MF(0, 2, j); ODT_FENCE();
      if (!last) rdB(b_cur, 2, j + 1); else if constexpr (next) { rdA(0); rdB(b_nxt, 2, 0); }
      ODT_FENCE();
    }
    const int t = b_cur; b_cur = b_nxt; b_nxt = b_nn; b_nn = t;
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
#undef ODT_FENCE
  stamp(2);
  split3_epilogue<WM, WN, TN, G::LDS, TRACE>(p, acc, lds, m0, n0, M, HoWo, 0, 1, tid, wm, wn, fr, fg);
  stamp(5);
}

// split-K combine: out = act(sum over ranges (in range order: deterministic) + bias (+ residual)); a thread per 16-byte
// chunk of an output row, grid-stride; the block's |max| goes to the output's range slot by (at most) one atomic
__global__ void __launch_bounds__(256) split_reduce_kernel(const ConvParams* __restrict__ pp) {
  const ConvParams p = *pp;
  __shared__ float red[4];
  const int Np = cout_padded(p.Cout), C4 = Np >> 2;
  const long M = (long)p.B * p.Ho * p.Wo;
  const size_t slab = (size_t)M * Np;
  const int HoWo = p.Ho * p.Wo;
  // fp16x2 pieces: undo the powers of two of the weight columns and of the A operand
  const float inv = p.h2_chinv != nullptr ? pow2f(-h2_in_scale_exp(p)) : 1.0f;
  float vmax = 0.f;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < M * C4; idx += (long)gridDim.x * 256) {
    const int m = (int)(idx / C4), col = (int)(idx - (long)m * C4) * 4;
    f32x4 v = *reinterpret_cast<const f32x4*>(p.partial + (size_t)m * Np + col);
    for (int k = 1; k < p.splitk; ++k) v += *reinterpret_cast<const f32x4*>(p.partial + (size_t)k * slab + (size_t)m * Np + col);
    if (p.h2_chinv != nullptr) v = v * (*reinterpret_cast<const f32x4*>(p.h2_chinv + col) * inv);     // ([cout_padded]: whole chunks)
    for (int e = 0; e < 4; ++e) v[e] += col + e < p.Cout ? p.bias[col + e] : 0.f;
    const int n = sfast_div(m, p.div_howo_mul, p.div_howo_sh), rr = m - n * HoWo;
    const int ho = sfast_div(rr, p.div_wo_mul, p.div_wo_sh), wo = rr - ho * p.Wo;
    if (p.res_mode != 0) {
      const size_t rpix = p.res_mode == 2 ? ((size_t)n * p.res_H + (size_t)(ho >> 1)) * p.res_W + (size_t)(wo >> 1)
                                          : ((size_t)n * p.res_H + (size_t)ho) * p.res_W + (size_t)wo;
      v += *reinterpret_cast<const f32x4*>(p.res + rpix * p.res_ldc + col);
    }
    if (p.relu == 1) {
      for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
    } else if (p.relu == 2) {
      for (int e = 0; e < 4; ++e) v[e] = v[e] * (1.0f / (1.0f + expf(-v[e])));
    } else if (p.relu == 3) {
      for (int e = 0; e < 4; ++e) v[e] = 1.0f / (1.0f + expf(-v[e]));
    }
    const size_t opix = ((size_t)n * p.out_H + ho + p.out_oy) * p.out_W + wo + p.out_ox;
    *reinterpret_cast<f32x4*>(p.out + opix * p.out_ldc + col) = v;
    vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
  }
  if (p.out_amax != nullptr) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, d));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = vmax;
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned b = __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));
      unsigned* w = amax_way(p.out_amax);
      if (b > __atomic_load_n(w, __ATOMIC_RELAXED)) atomicMax(w, b);
    }
  }
}

}  // namespace

template <int WM, int WN, int TN>
static void launch_split3(const ConvParams& p, const ConvParams* dev, unsigned grid, hipStream_t stream) {
  if (p.trace != nullptr) hipLaunchKernelGGL((conv_split3_kernel<WM, WN, TN, true>), dim3(grid), dim3(512), 0, stream, dev);
  else hipLaunchKernelGGL((conv_split3_kernel<WM, WN, TN, false>), dim3(grid), dim3(512), 0, stream, dev);
}


void launch_split_reduce(const ConvParams& p, const ConvParams* dev, hipStream_t stream) {
  const long chunks = (long)p.B * p.Ho * p.Wo * (cout_padded(p.Cout) / 4);
  const long blocks = (chunks + 255) / 256;
  // at most two blocks per CU (grid-stride; same-box A/B at b = 1: 2048 blocks 150.4, 512 167.5, 256 166.8, 128 159.6 FPS): every block ends with a conditional atomicMax on the ONE range slot of the output,
  // and the blocks of a short pass all find the slot empty -- 2040 same-address atomics serialised in L2 made this pass 35 us
  // per call at b = 1 (1.2 ms of the 6.8 ms frame, profiles/r04_kernel_stats_bench_b1_single_before.txt)
  const long capv = env_knob_long(K_SPLIT_REDUCE_BLOCKS, 512L), cap = capv > 0 ? capv : 512L;
  hipLaunchKernelGGL(split_reduce_kernel, dim3((unsigned)(blocks < cap ? blocks : cap)), dim3(256), 0, stream, dev);
}

int launch_conv_split3(const ConvParams& p, const ConvParams* dev, hipStream_t stream) {
  const long M = (long)p.B * p.Ho * p.Wo;
  const int bn = p.wt_split_bn != 0 ? p.wt_split_bn : conv_split_bn(p.Cout);
  const int bm = p.wt_split_bm;
  ODT_CHECK((bm == 256 || (bm == 128 && bn >= 128)) && p.Cin % 16 == 0 && p.kh * p.kw <= 32, "conv split3: unsupported tile / shape");
  const int sk = p.splitk > 1 ? p.splitk : 1;
  ODT_CHECK(sk == 1 || (p.partial != nullptr && p.in2 == nullptr && (p.kh * p.kw * p.Cin >> 4) >= sk),
            "conv split3: split-K needs a partial buffer, a single source and at least one stage per range");
  const unsigned grid = (unsigned)(((M + bm - 1) / bm) * (cout_padded(p.Cout) / bn) * sk);
  if (p.wt_split_kwr) {
    ODT_CHECK(bm == 256 && sk == 1 && p.kw == 3 && p.stride == 1 && p.in_Wa == p.Wo && p.in2 == nullptr,
              "conv split3k: unsupported shape");
    if (bn == 256) {
      if (p.trace != nullptr) hipLaunchKernelGGL((conv_split3k_kernel<4, true>), dim3(grid), dim3(512), 0, stream, dev);
      else hipLaunchKernelGGL((conv_split3k_kernel<4, false>), dim3(grid), dim3(512), 0, stream, dev);
    } else if (bn == 128) {
      if (p.trace != nullptr) hipLaunchKernelGGL((conv_split3k_kernel<2, true>), dim3(grid), dim3(512), 0, stream, dev);
      else hipLaunchKernelGGL((conv_split3k_kernel<2, false>), dim3(grid), dim3(512), 0, stream, dev);
    } else {
      hipLaunchKernelGGL((conv_split3k_kernel<1, false>), dim3(grid), dim3(512), 0, stream, dev);
    }
  } else if (bm == 256) {
    if (bn == 256) launch_split3<4, 2, 4>(p, dev, grid, stream);
    else if (bn == 128) launch_split3<4, 2, 2>(p, dev, grid, stream);
    else launch_split3<4, 2, 1>(p, dev, grid, stream);
  } else {
    if (bn == 256) launch_split3<2, 4, 2>(p, dev, grid, stream);
    else launch_split3<2, 4, 1>(p, dev, grid, stream);
  }
  if (sk > 1) launch_split_reduce(p, dev, stream);
  ODT_HIP(hipGetLastError());
  return 0;
}

}  // namespace odt
