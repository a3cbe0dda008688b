// fp16x2 split convolution, the few-row 1x1 layers (b = 1: res4 / res5 conv1, conv3, the box-head FCs): conv_h2_kernel's
// two-wave 64 x 64 / 64 x 128 tiles with DOUBLE stages -- two BK = 32 sub-stages (64 channels) between barriers.
//
// At 8 160 output rows these layers run 33 - 45 us for 8 us of MFMA work: a two-wave workgroup spends a BK = 32 stage's
// time waiting for its own barrier and LDS-DMA round trip (~1 us) with 0.4 us of MFMAs to put under it.  A double stage puts
// twice the MFMAs under the same round trip and halves the number of barriers of the reduction.  Same arithmetic, same
// weight image (two consecutive stage images per double stage) and the same K-slice rotation scheme as conv_h2_kernel, in
// units of double stages -- so a tile may start its reduction at another slice than there: the same products summed in
// another (per tile fixed) order, i.e. results equal at f32 rounding level, deterministic run to run.
// ("d" for double stage -- not round 3's archived conv_h2d_kernel.inc experiment, which put the activations on the LDS-DMA path.)
// Scope: 1x1, stride 1, dense input rows, one source, no split-K, an even number of 32-channel slices; everything else
// stays on conv_h2_kernel (launch_conv_h2 decides).  Reference ops: as conv_split.hip (nn.py:337-381, :503-521).
#include "conv_split_epilogue.hpp"

namespace odt {

namespace {

#define ODT_MFMA_F16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)

template <int TN>
struct H2dCfg {
  static constexpr int WM = 1, WN = 2, BM = 64, BN = 32 * TN * WN, NWV = 2, NTHR = 128;
  // sub-stage images exactly as conv_h2_kernel's (H2Cfg): A [piece 2][k-group 4][row][8 f16] (+ 32-B pad per k-group), B the linear
  // image the DMA writes; a double stage = two of each, back to back
  static constexpr int AKG = BM * 16 + 32, APL = 4 * AKG, ASUB = 2 * APL, ASTG = 2 * ASUB;
  static constexpr int BKG = BN * 16, BPL = 4 * BKG, BSUB = 2 * BPL, BSTG = 2 * BSUB;
  static constexpr int BOFF = 2 * ASTG;
  static constexpr int RING = BOFF + 2 * BSTG;
  static constexpr int CTILE = BM * (BN + 4) * 4;
  static constexpr int LDS = RING > CTILE ? RING : CTILE;
  static constexpr int NW = BSUB / 1024 / NWV;                               // DMA instructions per wave and SUB-stage
  static constexpr int RA = BM * 8 / NTHR;                                   // A rows (16-byte loads) per thread and SUB-stage: 4
  static_assert(LDS <= 160 * 1024 && BSUB % (1024 * NWV) == 0 && RA == 4, "LDS ring");
};

template <int TN>
__global__ void __launch_bounds__(128, 2) conv_h2d_kernel(const ConvParams* __restrict__ pp) {
  using G = H2dCfg<TN>;
  constexpr int NWV = G::NWV, AR = G::NTHR / 8, WN = G::WN;
  constexpr int BM = G::BM, BN = G::BN, AKG = G::AKG, APL = G::APL, ASUB = G::ASUB, ASTG = G::ASTG, BKG = G::BKG, BPL = G::BPL;
  constexpr int BSUB = G::BSUB, BSTG = G::BSTG, BOFF = G::BOFF, NW = G::NW, RA = G::RA;
  const ConvParams p = *pp;
  __shared__ __attribute__((aligned(16))) unsigned char lds[G::LDS];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = 0, wn = wave;
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
  const int cpt = p.Cin >> 5;                                // 32-channel slices (even)
  const int nd = cpt >> 1;                                   // double stages
  const int sexp = h2_in_scale_exp(p);
  const float a_scale = pow2f(sexp), h2_inv = pow2f(-sexp);

  const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(
      (void*)p.in, 0, (int)((unsigned)p.B * p.in_Ha * p.in_Wa * p.in_ldc * 4u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_wt = __builtin_amdgcn_make_buffer_rsrc(
      (void*)p.wt_split, 0, (int)((unsigned)ntn * cpt * (unsigned)BSUB), 0x00020000);

  // K-slice rotation as in conv_h2_kernel (all workgroups of such a launch start together), in units of double stages
  const int rot = (p.debug & 0x100) == 0 ? (mt % nd) * 2 : 0;
  unsigned l_b = ((unsigned)nt * (unsigned)cpt + (unsigned)rot) * (unsigned)BSUB;
  int b_wrap = cpt - rot;                    // sub-stages until the weight stream wraps to the first slice
  auto dma_b = [&](int boff) {               // one SUB-stage image -> LDS
#pragma unroll
    for (int i = 0; i < NW; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_wt, ODT_LDS_PTR(lds + boff + (i * NWV + wave) * 1024), 16,
                                               lane * 16 + (i * NWV + wave) * 1024, (int)l_b, 0, 0);
    l_b += (unsigned)BSUB;
    if (--b_wrap == 0) l_b -= (unsigned)cpt * (unsigned)BSUB;
  };
  dma_b(BOFF); dma_b(BOFF + BSUB);           // double stage 0's weights

  // activations: thread -> (row (t >> 3) + 16 j, 16-byte column t & 7) of a sub-stage's 128 bytes per row
  const int a_c = tid & 7, a_r = tid >> 3;
  const unsigned pix_bytes = (unsigned)p.in_ldc * 4u;
  unsigned a_off[RA];
#pragma unroll
  for (int j = 0; j < RA; ++j) {
    const int m = m0 + a_r + AR * j;
    a_off[j] = m < M ? (unsigned)m * pix_bytes + a_c * 16u : kOOB;
  }
  int l_cs = rot;
  f32x4 ga[2][RA];
  const bool a_nt = (p.debug & 0x800) != 0 && ntn == 1;
  auto load_a = [&](int sub) {               // one sub-stage (32 channels) of the load stream -> registers
#pragma unroll
    for (int j = 0; j < RA; ++j)
      ga[sub][j] = a_nt ? (f32x4)__builtin_amdgcn_raw_buffer_load_b128(rs_in, (int)a_off[j], l_cs * 128, 2)
                        : (f32x4)__builtin_amdgcn_raw_buffer_load_b128(rs_in, (int)a_off[j], l_cs * 128, 0);
    if (++l_cs == cpt) l_cs = 0;
  };
  auto store_slot = [&](int abuf, int sub, int j) {
    unsigned h0, l0, h1, l1;
    split2h(ga[sub][j][0], ga[sub][j][1], a_scale, h0, l0);
    split2h(ga[sub][j][2], ga[sub][j][3], a_scale, h1, l1);
    unsigned char* d = lds + abuf + sub * ASUB + (a_c >> 1) * AKG + (a_r + AR * j) * 16 + (a_c & 1) * 8;
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
  // ---- prologue: double stage 0 complete, double stage 1's A in registers, its weights in flight behind the barrier
  load_a(0); load_a(1);
#pragma unroll
  for (int j = 0; j < RA; ++j) { store_slot(0, 0, j); store_slot(0, 1, j); }
  // (conv_h2d_fits: Cin >= 128, i.e. at least two double stages)
  load_a(0); load_a(1);
  ODT_WAIT_VM_LGKM0(2 * RA);
  __builtin_amdgcn_s_barrier();
  dma_b(BOFF + BSTG); dma_b(BOFF + BSTG + BSUB);

  // fragments: fa[k-step parity][piece][t], fb[buffer][piece]
  f16x8 fa[2][2][2], fb[2][2];
  const int a_rd = fg * AKG + (wm * 64 + fr) * 16;
  const int b_rd = fg * BKG + (wn * TN * 32 + fr) * 16;
  // k-step ks = 0..3 of a double stage: sub-stage ks >> 1, its k16 step ks & 1
  auto rdA = [&](int abuf, int ks, int q) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
      fa[ks & 1][q][t] = *reinterpret_cast<const f16x8*>(lds + abuf + (ks >> 1) * ASUB + q * APL + (ks & 1) * 2 * AKG + a_rd + t * 512);
  };
  auto rdB = [&](int bbuf, int ks, int j, int dst) {
#pragma unroll
    for (int q = 0; q < 2; ++q)
      fb[dst][q] = *reinterpret_cast<const f16x8*>(lds + bbuf + (ks >> 1) * BSUB + q * BPL + (ks & 1) * 2 * BKG + b_rd + j * 512);
  };
  rdA(0, 0, 1); rdA(0, 0, 0);
  rdB(BOFF, 0, 0, 0);

#define ODT_MF(ks, qa, qb, j, bsel) { acc[0][j] = ODT_MFMA_F16(fa[(ks) & 1][qa][0], fb[bsel][qb], acc[0][j]); \
                                       acc[1][j] = ODT_MFMA_F16(fa[(ks) & 1][qa][1], fb[bsel][qb], acc[1][j]); }
#define ODT_FENCE() __builtin_amdgcn_sched_barrier(0)
  int a_cur = 0, a_nxt = ASTG, b_cur = BOFF, b_nxt = BOFF + BSTG;
  // One double stage = 4 TN column groups (k-step, j).  NEXT: double stage c+1 exists (its A: registers -> LDS in the first
  // groups); PRE: double stage c+2 exists (fetch its A once the registers are free, start its weight DMA behind the barrier
  // into the buffers this stage is leaving).  Peeled so that no MFMA sits in a conditional arm.
  auto step = [&](auto NEXT, auto PRE) {
    constexpr bool next = decltype(NEXT)::value, pre = decltype(PRE)::value;
    constexpr int NG = 4 * TN;
    constexpr int SLOTS = 2 * RA;                          // 8 store slots of double stage c+1: slot s = (sub s / RA, row group s % RA)
    constexpr int SPG = (SLOTS + NG - 2) / (NG - 1);       // slots per group in front of the barrier
    ODT_FENCE();
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int ks = g / TN, j = g % TN, bsel = g & 1;
      const bool last = g == NG - 1;
      if (last) {
        if constexpr (pre) ODT_WAIT_VM_LGKM0(2 * RA); else ODT_WAIT_VM_LGKM0(0);
        __builtin_amdgcn_s_barrier();
        ODT_FENCE();
        if constexpr (pre) { dma_b(b_cur); dma_b(b_cur + BSUB); }
        if constexpr (next) rdB(b_nxt, 0, 0, bsel ^ 1);
      } else {
        rdB(b_cur, (g + 1) / TN, (g + 1) % TN, bsel ^ 1);
      }
      ODT_FENCE();
      ODT_MF(ks, 1, 0, j, bsel); ODT_FENCE();              // lo * hi
      if (last) {
        if constexpr (next) rdA(a_nxt, 0, 1);
      } else {
        if constexpr (next) {
#pragma unroll
          for (int q = 0; q < SPG; ++q)
            if (g * SPG + q < SLOTS) store_slot(a_nxt, (g * SPG + q) / RA, (g * SPG + q) % RA);
        }
      }
      ODT_FENCE();
      ODT_MF(ks, 0, 1, j, bsel); ODT_FENCE();              // hi * lo
      if (last) {
        if constexpr (next) rdA(a_nxt, 0, 0);
      } else {
        // the next k-step's A fragments go out under the current k-step's last column group
        if (j == TN - 1 && ks < 3) { rdA(a_cur, ks + 1, 1); rdA(a_cur, ks + 1, 0); }
        // the fetch of double stage c+2 reuses the registers: sub-stage s behind its last slot's store
        if constexpr (pre) {
          if (g == (RA - 1) / SPG) load_a(0);
          if (g == (SLOTS - 1) / SPG) load_a(1);
        }
      }
      ODT_FENCE();
      ODT_MF(ks, 0, 0, j, bsel); ODT_FENCE();              // hi * hi
    }
    { const int t = a_cur; a_cur = a_nxt; a_nxt = t; }
    { const int t = b_cur; b_cur = b_nxt; b_nxt = t; }
  };
  {
    int c = 0;
    for (; c + 2 < nd; ++c) step(std::true_type{}, std::true_type{});
    if (c + 1 < nd) { step(std::true_type{}, std::false_type{}); ++c; }
    step(std::false_type{}, std::false_type{});
  }
  split3_epilogue<G::WM, WN, TN, G::LDS, false, G::NTHR>(p, acc, lds, m0, n0, M, HoWo, 0, 1, tid, wm, wn, fr, fg, h2_inv);
}

#undef ODT_MF
#undef ODT_FENCE

}  // namespace

bool conv_h2d_fits(const ConvParams& p) {
  const int bn = p.wt_split_bn;
  return p.wt_split_kind == 2 && p.wt_split_bm == 64 && (bn == 64 || bn == 128) && !p.wt_split_kwr && p.kh == 1 && p.kw == 1 &&
         p.stride == 1 && p.pad_t == 0 && p.pad_l == 0 && p.H == p.in_Ha && p.W == p.in_Wa && p.Ho == p.H && p.Wo == p.W &&
         p.in2 == nullptr && p.splitk <= 1 && p.f_wt == nullptr && p.nlvl <= 1 && p.trace == nullptr && p.Cin % 64 == 0 && p.Cin >= 128;
}

void launch_conv_h2d(const ConvParams& p, const ConvParams* dev, unsigned grid, hipStream_t stream) {
  if (p.wt_split_bn == 64) hipLaunchKernelGGL((conv_h2d_kernel<1>), dim3(grid), dim3(128), 0, stream, dev);
  else hipLaunchKernelGGL((conv_h2d_kernel<2>), dim3(grid), dim3(128), 0, stream, dev);
}

}  // namespace odt
