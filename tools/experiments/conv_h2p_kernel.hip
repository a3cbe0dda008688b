// ---------------------------------------------------------------------------------------------------------
// fp16x2 pointwise convolution: the dense single-source 1x1 layers (bottleneck conv1, FPN laterals, plain conv3) on the
// arithmetic, tile shapes, weight image and epilogue of conv_h2_kernel (conv_h2.hip) -- bit-identical results -- with the
// ACTIVATION stream rebuilt around what bounds these layers: bytes in flight per CU.
//   conv_h2_kernel lands a stage's f32 activations in registers (ga[]: 16 VGPRs, one 32-KB stage per CU in flight for about
//   a stage's MFMA time) and its register file is full; at the ~2-3 us an HBM request takes under load that is ~3 TB/s over
//   256 CUs however the loop is scheduled (res4 conv1: 334 MB in 110 us with the matrix pipe and HBM both asking for ~45).
//   Here a stage's 32 channels x 256 pixel rows go global -> LDS by LDS-DMA as RAW f32 (32 KB, no registers), into a ring of
//   NSA = 3 (256-wide tiles) or 4 stages: two to three stages -- 64 to 96 KB per CU -- are in flight across two full stages
//   of MFMA time.  The split into f16 pieces moves to the fragment read: a lane reads the 8 f32 of its (pixel row, k-group)
//   and converts (x 2^s -> hi, lo) in registers; both waves of a row pair do it (twice the conversions of the store-side
//   split, a third of the VALU slots the MFMAs leave) and the ds_write traffic of the split stage disappears.
//   * vmcnt counts a wave's requests in order, so a deep activation ring behind shallow weight stages cannot be waited on
//     by one wave: the waves are specialised -- waves 4..7 issue the activation DMA (8 x 1 KB per wave and stage, waited
//     with vmcnt(8 (NSA - 2))), waves 0..3 the weight DMA (two stages, vmcnt(0)); all eight run the same MFMA schedule.
//   * requests past the last stage are issued out of range (no data moves): the counted waits stay compile-time constants.
//   * LDS image of a stage: pixel row r at r * 128, its eight 16-byte chunks XOR-swizzled by (r >> 1) & 7 -- the sixteen
//     lanes of a ds_read_b128 group (rows distinct mod 16) touch sixteen different 16-byte slots of the 256-byte bank row;
//     the DMA writes LDS lane-linearly, so the swizzle is applied to the lanes' GLOBAL chunk instead.
// K order, K-slice rotation, scales and epilogue: conv_h2_kernel's.  Reference ops: as conv_split.hip.
#include "conv_split_epilogue.hpp"

namespace odt {

namespace {

#define ODT_MFMA_F16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)

template <int TN>
struct H2pCfg {
  static constexpr int WM = 4, WN = 2, BM = 64 * WM, BN = 32 * TN * WN, NTHR = 512;
  static constexpr int STAGE_A = BM * 128;                                  // raw f32: 32 channels of BM pixel rows
  static constexpr int BKG = BN * 16, BPL = 4 * BKG, STAGE_B = 2 * BPL;     // B stage: the linear image the DMA writes (conv_h2.hip)
  static constexpr int NSA = (160 * 1024 - 2 * STAGE_B) / STAGE_A >= 4 ? 4 : 3;
  static constexpr int BOFF = NSA * STAGE_A;
  static constexpr int RING = BOFF + 2 * STAGE_B;
  static constexpr int CTILE = 128 * (BN + 4) * 4;                          // 128-row epilogue passes
  static constexpr int LDS = RING > CTILE ? RING : CTILE;
  static constexpr int NWA = STAGE_A / 4096, NWB = STAGE_B / 4096;          // DMA instructions per issuing wave and stage
  static_assert(LDS <= 160 * 1024 && NSA >= 3 && NWB >= 1, "LDS ring");
};

template <int TN>
__global__ void __launch_bounds__(512, 2) conv_h2p_kernel(const ConvParams* __restrict__ pp) {
  using G = H2pCfg<TN>;
  constexpr int WM = G::WM, WN = G::WN, BM = G::BM, BN = G::BN, STAGE_A = G::STAGE_A, BKG = G::BKG, BPL = G::BPL;
  constexpr int STAGE_B = G::STAGE_B, BOFF = G::BOFF, NSA = G::NSA, NWA = G::NWA, NWB = G::NWB;
  const ConvParams p = *pp;
  __shared__ __attribute__((aligned(16))) unsigned char lds[G::LDS];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const bool is_a = wave >= 4;               // this wave requests activations (else weights)
  const int wr = wave & 3;
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
  const int nsteps = p.Cin >> 5;             // 32-channel slices
  const int sexp = h2_in_scale_exp(p);
  const float a_scale = pow2f(sexp), h2_inv = pow2f(-sexp);
  const unsigned pix_bytes = (unsigned)p.in_ldc * 4u;

  // dense rows: pixel m at m * pix_bytes; rows >= M lie behind the descriptor's range (zeros)
  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, (int)((unsigned)M * pix_bytes), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_wt = __builtin_amdgcn_make_buffer_rsrc(
      (void*)p.wt_split, 0, (int)((unsigned)ntn * nsteps * (unsigned)STAGE_B), 0x00020000);

  // K-slice rotation (conv_h2_kernel): workgroup (mt, .) starts its reduction at slice mt mod nsteps and wraps
  const int rot = (p.debug & 0x100) == 0 ? mt % nsteps : 0;
  // ---- requests.  req: stages requested so far (both roles); a stage >= nsteps is requested out of range
  int req = 0, l_cs = rot;
  unsigned l_b = ((unsigned)nt * (unsigned)nsteps + (unsigned)rot) * (unsigned)STAGE_B;
  // activations: instruction i of wave wr writes the 1-KB piece d = 4 i + wr of the stage = rows 8 d + lane / 8, chunk
  // position lane % 8; the chunk it fetches there is position ^ ((row >> 1) & 7)
  const unsigned a_row = (unsigned)(wr * 8 + (lane >> 3));
  const unsigned a_voff = (unsigned)(m0 + (int)a_row) * pix_bytes + (unsigned)(((lane & 7) ^ (((wr & 1) << 2) + (lane >> 4))) << 4);
  const unsigned a_istep = 32u * pix_bytes;
  auto dma_a = [&](int abuf) {
    const unsigned v0 = req < nsteps ? a_voff : kOOB;
#pragma unroll
    for (int i = 0; i < NWA; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, ODT_LDS_PTR(lds + abuf * STAGE_A + (i * 4 + wr) * 1024), 16,
                                               (int)(v0 + (unsigned)i * a_istep), l_cs * 128, 0, 0);
    ++req;
    if (++l_cs == nsteps) l_cs = 0;
  };
  auto dma_b = [&](int boff) {
    const unsigned v0 = req < nsteps ? (unsigned)(lane * 16) : kOOB;
#pragma unroll
    for (int i = 0; i < NWB; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_wt, ODT_LDS_PTR(lds + boff + (i * 4 + wr) * 1024), 16,
                                               (int)(v0 + (unsigned)((i * 4 + wr) * 1024)), (int)l_b, 0, 0);
    ++req;
    l_b += (unsigned)STAGE_B;
    if (req + rot == nsteps) l_b -= (unsigned)nsteps * (unsigned)STAGE_B;        // the weight stream wraps to the first slice
  };

  f32x16 acc[2][TN];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int fr = lane & 31, fg = lane >> 5;
  // fragments: fa[k-step][piece][t] (converted), raw[t][half] (the next k-step's f32), fb[buffer][piece]
  f16x8 fa[2][2][2], fb[2][2];
  f32x4 raw[2][2];
  // lane (row fr of a 32-row block, k-group fg) reads chunks 4 kst + 2 fg + h of its row: position = chunk ^ ((row >> 1) & 7)
  const int a_rd = (wm * 64 + fr) * 128;
  const int a_sw = (fg << 5) ^ (((fr >> 1) & 7) << 4);
  const int b_rd = fg * BKG + (wn * TN * 32 + fr) * 16;
  auto rdRaw = [&](int abuf, int kst) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int h = 0; h < 2; ++h)
        raw[t][h] = *reinterpret_cast<const f32x4*>(lds + abuf * STAGE_A + a_rd + t * 4096 + (a_sw ^ ((4 * kst + h) << 4)));
  };
  auto cvt = [&](int kst, int t) {
    unsigned h0, h1, h2, h3, l0, l1, l2, l3;
    split2h(raw[t][0][0], raw[t][0][1], a_scale, h0, l0);
    split2h(raw[t][0][2], raw[t][0][3], a_scale, h1, l1);
    split2h(raw[t][1][0], raw[t][1][1], a_scale, h2, l2);
    split2h(raw[t][1][2], raw[t][1][3], a_scale, h3, l3);
    fa[kst][0][t] = (f16x8)u32x4{h0, h1, h2, h3};
    fa[kst][1][t] = (f16x8)u32x4{l0, l1, l2, l3};
  };
  auto rdB = [&](int bbuf, int kst, int j, int dst) {
#pragma unroll
    for (int q = 0; q < 2; ++q) fb[dst][q] = *reinterpret_cast<const f16x8*>(lds + bbuf + q * BPL + kst * 2 * BKG + b_rd + j * 512);
  };

  // ---- prologue: stages 0 .. NSA-2 of the activations and stage 0 of the weights requested; stage 0 complete behind the
  // barrier; then the ring's last buffer / the second weight stage
  if (is_a) {
#pragma unroll
    for (int s = 0; s < NSA - 1; ++s) dma_a(s);
    ODT_WAIT_VM_LGKM0(NWA * (NSA - 2));
  } else {
    dma_b(BOFF);
    ODT_WAIT_VM_LGKM0(0);
  }
  __builtin_amdgcn_s_barrier();
  if (is_a) dma_a(NSA - 1); else dma_b(BOFF + STAGE_B);
  rdRaw(0, 0);
  rdB(BOFF, 0, 0, 0);
  cvt(0, 0); cvt(0, 1);

#define ODT_MF(kst, qa, qb, j, bsel) { acc[0][j] = ODT_MFMA_F16(fa[kst][qa][0], fb[bsel][qb], acc[0][j]); \
                                        acc[1][j] = ODT_MFMA_F16(fa[kst][qa][1], fb[bsel][qb], acc[1][j]); }
#define ODT_FENCE() __builtin_amdgcn_sched_barrier(0)
  int a_cur = 0, a_nxt = 1, b_cur = BOFF, b_nxt = BOFF + STAGE_B;
  // One stage = 2 TN column groups (k-step, j).  The second k-step's raw activations are read behind the first group and
  // converted under the second; in front of the stage's last group: stage c+1 complete (counted wait of the issuing role,
  // barrier) -- which also releases stage c's buffers to the requests for stages c + NSA / c + 2 -- and the next stage's
  // first fragments, converted under the last group's MFMAs.  Every stage runs the same code (requests past the end move
  // nothing; the fragments read behind the last barrier are not used).
  constexpr int NG = 2 * TN;
  constexpr int GC = NG > 2 ? 1 : 0;         // the group whose slots convert the second k-step's fragments
  for (int c = 0; c < nsteps; ++c) {
    ODT_FENCE();
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int kst = g / TN, j = g % TN, bsel = g & 1;
      const bool last = g == NG - 1;
      if (last) {
        if (is_a) ODT_WAIT_VM_LGKM0(NWA * (NSA - 2)); else ODT_WAIT_VM_LGKM0(0);
        __builtin_amdgcn_s_barrier();
        ODT_FENCE();
        rdRaw(a_nxt, 0);
        rdB(b_nxt, 0, 0, bsel ^ 1);
        if (is_a) dma_a(a_cur); else dma_b(b_cur);
      } else {
        rdB(b_cur, (g + 1) / TN, (g + 1) % TN, bsel ^ 1);
        if (g == 0) rdRaw(a_cur, 1);
      }
      ODT_FENCE();
      ODT_MF(kst, 1, 0, j, bsel); ODT_FENCE();             // lo * hi
      if (last) cvt(0, 0);
      else if (g == GC) cvt(1, 0);
      ODT_FENCE();
      ODT_MF(kst, 0, 1, j, bsel); ODT_FENCE();             // hi * lo
      if (last) cvt(0, 1);
      else if (g == GC) cvt(1, 1);
      ODT_FENCE();
      ODT_MF(kst, 0, 0, j, bsel); ODT_FENCE();             // hi * hi
    }
    a_cur = a_nxt;
    a_nxt = a_nxt + 1 == NSA ? 0 : a_nxt + 1;
    { const int t = b_cur; b_cur = b_nxt; b_nxt = t; }
  }
#undef ODT_MF
#undef ODT_FENCE
  // the ring becomes the C tile: the out-of-range requests of the last stages and every fragment read are behind this
  ODT_WAIT_VM_LGKM0(0);
  __builtin_amdgcn_s_barrier();
  split3_epilogue<WM, WN, TN, G::LDS, false, G::NTHR>(p, acc, lds, m0, n0, M, HoWo, 0, 1, tid, wm, wn, fr, fg, h2_inv);
}

}  // namespace

// dense single-source 1x1 conv on 256-row fp16x2 tiles (launch_conv_h2 checked the family's requirements)
bool conv_h2p_fits(const ConvParams& p) {
  return p.wt_split_kind == 2 && p.wt_split_bm == 256 && !p.wt_split_kwr && p.kh == 1 && p.kw == 1 && p.stride == 1 && p.pad_t == 0 &&
         p.pad_l == 0 && p.H == p.in_Ha && p.W == p.in_Wa && p.Ho == p.H && p.Wo == p.W && p.in2 == nullptr && p.splitk <= 1 &&
         p.f_wt == nullptr && p.head_wt == nullptr && p.nlvl <= 1 && p.Cin % 32 == 0 && p.Cin >= 32 &&
         (p.wt_split_bn == 256 || p.wt_split_bn == 128 || p.wt_split_bn == 64) &&
         (double)p.B * p.Ho * p.Wo * p.in_ldc * 4.0 < 2147483648.0;
}

void launch_conv_h2p(const ConvParams& p, const ConvParams* dev, unsigned grid, hipStream_t stream) {
  const int bn = p.wt_split_bn;
  if (bn == 256) hipLaunchKernelGGL((conv_h2p_kernel<4>), dim3(grid), dim3(512), 0, stream, dev);
  else if (bn == 128) hipLaunchKernelGGL((conv_h2p_kernel<2>), dim3(grid), dim3(512), 0, stream, dev);
  else hipLaunchKernelGGL((conv_h2p_kernel<1>), dim3(grid), dim3(512), 0, stream, dev);
}

}  // namespace odt
