// f32 implicit-GEMM convolution on the bf16 matrix pipe of gfx950 ("bf16x3 split"): weight images, the per-handle policy
// (which convs take the split kernels and which family) and the dispatch.  Kernels: conv_split3.hip (conv_split3_kernel /
// conv_split3k_kernel: 8 waves, LDS-DMA weight stages, three-stage ring; the default), conv_split1.hip (one-stage 4-wave
// loop: the 64-wide layers).  Same reference ops as conv_igemm.hip: nn.py:337-381 conv2d + :1771-1774 folded BN + ReLU,
// :503-521 residual, :949-1014 FPN lateral.  The arithmetic is described in conv_split_common.hpp.
// Scope: no residual, a same-shape residual or a nearest-2x upsampled one, an optional K-concatenated second A source
// (1x1), Cout % 64 == 0 (padded), Cin % 32 == 0, 16-byte-aligned output rows; everything else stays on the exact-f32 MFMA
// kernel (launch_conv decides).
#include "conv_split_common.hpp"

namespace odt {

namespace {

// f32 weights [Cout][K] -> per-stage image of bf16 pieces (one thread per 8 consecutive k of a row)
// (kscale != nullptr: the row is multiplied by kscale[k] first -- a per-input-channel gate folded into a 1x1 conv's weights)
__global__ void split_weights_kernel(const float* __restrict__ wt, int Cout, int K, int SBN, int bk, unsigned short* __restrict__ img,
                                     const float* __restrict__ kscale) {
  const int nsl = K / bk, kgs = bk >> 3;     // stages along K, k-groups of 8 per stage
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;       // (n, k8), n over the padded Cout
  const long total = (long)cout_padded(Cout) * (K >> 3);
  if (idx >= total) return;
  const int n = (int)(idx / (K >> 3)), k8 = (int)(idx - (long)n * (K >> 3));
  const int tn = n / SBN, nn = n - tn * SBN, sl = k8 / kgs, kg = k8 - sl * kgs;
  for (int e = 0; e < 8; e += 2) {
    unsigned piece[3];
    float w0 = n < Cout ? wt[(size_t)n * K + k8 * 8 + e] : 0.f, w1 = n < Cout ? wt[(size_t)n * K + k8 * 8 + e + 1] : 0.f;
    if (kscale != nullptr) { w0 = w0 * kscale[k8 * 8 + e]; w1 = w1 * kscale[k8 * 8 + e + 1]; }
    split2(w0, w1, piece[0], piece[1], piece[2]);
    for (int q = 0; q < 3; ++q) {
      const size_t at = ((((size_t)(tn * nsl + sl) * 3 + q) * kgs + kg) * SBN + nn) * 8 + e;
      img[at] = (unsigned short)(piece[q] & 0xffffu);
      img[at + 1] = (unsigned short)(piece[q] >> 16);
    }
  }
}

// conv_split3_kernel's image: [n-tile][stage][piece][k-group 2][BN n][8 k], stage order = (16-channel slice, tap)
// for the first source, then the second source's slices; wt is [Cout][tap][Cin] (+ [Cin2] behind it)
__global__ void split_weights3_kernel(const float* __restrict__ wt, int Cout, int K, int SBN, int ntaps, int Cin,
                                      unsigned short* __restrict__ img, const float* __restrict__ kscale) {
  const int nst = K >> 4, nst1 = ntaps * (Cin >> 4);
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;       // (n, stage, k-group)
  const long total = (long)cout_padded(Cout) * nst * 2;       // n over the padded Cout: zero rows behind the last channel
  if (idx >= total) return;
  const int n = (int)(idx / (nst * 2)), rem = (int)(idx - (long)n * (nst * 2)), st = rem >> 1, kg = rem & 1;
  int k0;
  if (st < nst1) { const int cs = st / ntaps, tap = st - cs * ntaps; k0 = tap * Cin + cs * 16; }
  else k0 = ntaps * Cin + (st - nst1) * 16;
  k0 += kg * 8;
  const int tn = n / SBN, nn = n - tn * SBN;
  for (int e = 0; e < 8; e += 2) {
    unsigned piece[3];
    float w0 = n < Cout ? wt[(size_t)n * K + k0 + e] : 0.f, w1 = n < Cout ? wt[(size_t)n * K + k0 + e + 1] : 0.f;
    if (kscale != nullptr) { w0 = w0 * kscale[k0 + e]; w1 = w1 * kscale[k0 + e + 1]; }
    split2(w0, w1, piece[0], piece[1], piece[2]);
    for (int q = 0; q < 3; ++q) {
      const size_t at = ((((size_t)(tn * nst + st) * 3 + q) * 2 + kg) * SBN + nn) * 8 + e;
      img[at] = (unsigned short)(piece[q] & 0xffffu);
      img[at + 1] = (unsigned short)(piece[q] >> 16);
    }
  }
}

}  // namespace

size_t conv_split_weight_bytes(int Cout, int K) { return (size_t)cout_padded(Cout) * K * 6; }

// n-tile width of the configuration that takes a layer with this Cout (0: none)
int conv_split_bn(int Cout) { const int n = cout_padded(Cout); return n % 256 == 0 ? 256 : (n % 128 == 0 ? 128 : 64); }
int conv_split_bm(int Cout) { return cout_padded(Cout) % 256 == 0 ? 128 : 256; }

bool conv_split_supported(const ConvParams& p) {
  const double wbytes = (double)p.Cout * (p.kh * p.kw * p.Cin + (p.in2 != nullptr ? p.Cin2 : 0)) * 6.0;
  const bool res_ok = p.res_mode == 0 || (p.res_mode == 1 && p.res_H == p.Ho && p.res_W == p.Wo) ||
                      (p.res_mode == 2 && 2 * p.res_H >= p.Ho && 2 * p.res_W >= p.Wo);
  const bool src2_ok = p.in2 == nullptr || (p.kh == 1 && p.kw == 1 && p.Cin2 % 32 == 0 && p.in2_ldc % 4 == 0);
  // a channel count that is not a multiple of 64 needs room for the padded n-tile in the output (and residual) rows
  const int np = cout_padded(p.Cout);
  const bool pad_ok = np == p.Cout || (p.Cout >= 16 && p.out_ldc >= np && (p.res_mode == 0 || p.res_ldc >= np));
  return pad_ok && p.Cin % 32 == 0 && src2_ok && res_ok && p.out_ldc % 4 == 0 &&
         p.in_ldc % 4 == 0 && wbytes < 2147483648.0;
}

// ---- policy: which convs take the split kernels, and which family.  A model's policy is fixed when its handle is
// created (odt_config.conv_arith / conv_split_family -> odt_create) and recorded with the handle (odt_describe); the
// ODT_CONV_* environment variables are debug / A-B overrides on top of it, read ONCE per handle -- and per call only by
// the stand-alone test entry points (odt_op_conv2d ...), which have no handle.
ConvPolicy conv_policy_default() {
  ConvPolicy q;
  q.arith = 1;            // bf16x3 split where it pays (same-box A/B at b=8 1080p: 116 -> 178 FPS, parity suite green)
  q.family = 2;           // fp16x2 kernels where a layer has 256-row tiles and a recorded input range (same-box A/B at b=8
                          // 1080p: 188 -> 259 FPS, profiles/r03_fp16x2_vs_bf16x3_ab.txt), conv_split3_kernel where its tiles fill the chip
  q.min_tiles = 256;      // one- / two-stage kernels: A/B at b=8 and b=1: 256 > 384 > 128 >> 64
  q.min_tiles3 = 200;
  q.min_k = 64;           // A/B at b=8: K >= 256: 155.0, >= 128: 156.2, >= 64: 156.6 FPS
  q.h2s_maxk = 0;         // fp16x2: reductions up to this K take the 128 x 128 two-per-CU tile (A/B knob; measured: no gain)
  q.h2_few_tiles = true;  // fp16x2: layers without enough 256-row tiles take 128 x 128 tiles instead of bf16x3 + split-K
  q.h2_n64 = true;        // fp16x2: the 64-wide layers too
  q.h2k_fewrows = 1;  // fp16x2 kw-reuse kernel on 256 x 128 tiles without split-K where those fill the chip (round-6 A/B knob)
  q.h2k_splitk = true;    // fp16x2 kw-reuse kernel with split-K for the stride-1 KH x 3 layers of few rows (ODT_CONV_H2K_SPLITK=0: A/B)
  q.fill_div = 6;         // split-K layers are taken when tiles x ranges reach min_tiles3 / fill_div workgroups (b = 1: fc6 / fc7 leave the
                          // exact-f32 kernel: 139.4 -> 144.9 FPS same box, profiles/r04_b1_filldiv_ab.txt; ODT_CONV_SPLIT3_FILLDIV: A/B)
  q.h2_bm64 = 3;          // fp16x2: 64 x 128 two-wave tiles instead of 128 x 128 + split-K where only those fill the chip: 0 off | 1 for
                          // reductions up to K = 1024 (no split-K at all) | 2 also the longer ones, cut in two | 3 = 1 on 64 x 64 tiles (ODT_CONV_H2_BM64: A/B)
  q.h2_n64_bm512 = 1;     // fp16x2 kw-reuse kernel on 64-wide layers: 512 x 64 tiles (eight waves stacked along M: 24 MFMAs per wave and
                          // stage instead of 12) where they fill the chip (res2 conv2 1.035 -> 0.897 ms, same box); 0 off, 2 wherever
                          // the shape allows, the generic kernel included (tests)
  q.min_bn = 0; q.force_bm3 = 0; q.splitk_max = 8; q.force_splitk = 0; q.kw_reuse = true; q.kwr_n64 = true; q.src2 = true; q.res2 = true;
  return q;
}

ConvPolicy conv_policy_from_env(ConvPolicy q) {
  auto geti = [&](Knob k, long* dst) {
    const KnobVal& e = env_knob(k);
    if (e.set) *dst = e.i;
  };
  long v;
  v = q.arith; geti(K_CONV_SPLIT, &v); q.arith = v != 0 ? 1 : 0;
  v = q.family; geti(K_CONV_SPLIT_PIPE, &v); q.family = v >= 3 ? 3 : (v == 2 ? 2 : 1);
  geti(K_CONV_SPLIT_MINTILES, &q.min_tiles);
  geti(K_CONV_SPLIT3_MINTILES, &q.min_tiles3);
  v = q.min_k; geti(K_CONV_SPLIT_MINK, &v); q.min_k = (int)v;
  v = q.min_bn; geti(K_CONV_SPLIT_MINBN, &v); q.min_bn = (int)v;
  v = q.h2s_maxk; geti(K_CONV_H2S_MAXK, &v); q.h2s_maxk = (int)v;
  v = q.h2_few_tiles; geti(K_CONV_H2_FEW_TILES, &v); q.h2_few_tiles = v != 0;
  v = q.h2_n64; geti(K_CONV_H2_N64, &v); q.h2_n64 = v != 0;
  v = q.h2_n64_bm512; geti(K_CONV_H2_N64_BM512, &v); q.h2_n64_bm512 = (int)v;
  v = q.h2_bm64; geti(K_CONV_H2_BM64, &v); q.h2_bm64 = (int)v;
  v = q.h2k_splitk; geti(K_CONV_H2K_SPLITK, &v); q.h2k_splitk = v != 0;
  v = q.h2k_fewrows; geti(K_CONV_H2K_FEWROWS, &v); q.h2k_fewrows = (int)v;
  v = q.fill_div; geti(K_CONV_SPLIT3_FILLDIV, &v); q.fill_div = v < 1 ? 1 : (int)v;
  v = q.force_bm3; geti(K_CONV_SPLIT3_BM, &v); q.force_bm3 = (int)v;
  v = q.splitk_max; geti(K_CONV_SPLIT3_SPLITK, &v); q.splitk_max = v < 1 ? 1 : (v > 16 ? 16 : (int)v);
  v = 1; geti(K_CONV_SPLIT3_KWR, &v); q.kw_reuse = v != 0;
  v = q.kwr_n64; geti(K_CONV_SPLIT3_KWR_N64, &v); q.kwr_n64 = v != 0;
  v = q.force_splitk; geti(K_CONV_SPLIT3_FORCE_SPLITK, &v); q.force_splitk = v < 0 ? 0 : (v > 16 ? 16 : (int)v);
  v = 1; geti(K_CONV_SPLIT_SRC2, &v); q.src2 = v != 0;      // 0 keeps the fused stage-entry convs on the f32 kernel
  v = 1; geti(K_CONV_SPLIT_RES2, &v); q.res2 = v != 0;      // 0 keeps the FPN laterals on the f32 kernel
  return q;
}

// conv_split3_kernel's way of filling the chip with this layer, if it has one: 256-row tiles, 128-row tiles, or 128-row
// tiles with the reduction cut into split-K ranges (layers of few output rows: everything at b=1 below res3, the box
// head's FC layers, the coarse pyramid levels)
static bool split3_fit(const ConvParams& p, const ConvPolicy& q, int* bm, int* bn, int* sk) {
  const int bn0 = conv_split_bn(p.Cout);
  const int K = p.kh * p.kw * p.Cin + (p.in2 != nullptr ? p.Cin2 : 0);
  if (q.family < 2 || bn0 == 0 || p.kh * p.kw > 32 || K < 32 || p.Cin % 16 != 0) return false;
  const long M = (long)p.B * p.Ho * p.Wo;
  const int nsteps = K >> 4;
  *bn = bn0; *sk = 1;
  auto with_forced_sk = [&]() {
    if (q.force_splitk > 1 && p.in2 == nullptr && nsteps >= q.force_splitk) *sk = q.force_splitk;
    return true;
  };
  if (q.force_bm3 == 256 || (q.force_bm3 == 128 && bn0 >= 128)) { *bm = q.force_bm3; return with_forced_sk(); }
  // (64-wide layers stay on the one-stage 256 x 64 tile: a 64 x 32 wave tile reads too many fragments per MFMA --
  // same-box A/B at b=8: res2 conv2 132 vs 118 TF, conv0 136 vs 112)
  // ... except where the kw-reuse kernel applies: with a third of the A-side work the 64-wide 3x3 layers (res2 conv2)
  // come out ahead on it (same-box A/B in profiles/r02_kw_reuse_n64_ab.txt)
  const bool kwr_ok = q.kw_reuse && q.kwr_n64 && p.kw == 3 && p.stride == 1 && p.in_Wa == p.Wo && p.in2 == nullptr && 2 * p.dil <= 4 &&
                      p.Ho * p.Wo >= 256;
  if (bn0 < 128 && !kwr_ok) return false;
  const long t256 = ((M + 255) / 256) * (cout_padded(p.Cout) / bn0), t128 = ((M + 127) / 128) * (cout_padded(p.Cout) / bn0);
  if (t256 >= q.min_tiles3) { *bm = 256; return with_forced_sk(); }
  if (bn0 < 128) return false;
  if (t128 >= q.min_tiles3) { *bm = 128; return with_forced_sk(); }
  if (q.splitk_max > 1 && p.in2 == nullptr) {
    int k = (int)((q.min_tiles3 + t128 - 1) / t128);
    if (k > q.splitk_max) k = q.splitk_max;
    while (k > 1 && nsteps / k < 8) --k;            // at least eight stages per range
    if (k > 1 && t128 * k >= q.min_tiles3 / q.fill_div) { *bm = 128; *sk = k; return true; }
  }
  return false;
}

bool conv_split_wanted(const ConvParams& p, const ConvPolicy& q) {
  if (q.arith == 0 || !conv_split_supported(p)) return false;
  if (p.kh * p.kw * p.Cin + (p.in2 != nullptr ? p.Cin2 : 0) < q.min_k) return false;
  if (p.in2 != nullptr && !q.src2) return false;
  if (p.res_mode == 2 && !q.res2) return false;
  const long M = (long)p.B * p.Ho * p.Wo;
  const int bm = conv_split_bm(p.Cout), bn = conv_split_bn(p.Cout);
  if (bn < q.min_bn) return false;
  int b3, n3, k3;
  if (split3_fit(p, q, &b3, &n3, &k3)) return true;
  // one- / two-stage kernels: below one workgroup per CU the exact-f32 kernel's smaller tiles fill the chip better
  return ((M + bm - 1) / bm) * (cout_padded(p.Cout) / bn) >= q.min_tiles;
}

// which kernel family takes a conv that conv_split_wanted() accepted: family 1 = one-stage BK = 32 kernel everywhere,
// 3 (default) = conv_split3_kernel where its tiles fill the chip (the one-stage kernel keeps the 64-wide layers)
void conv_split_choose(ConvParams& p, const ConvPolicy& q) {
  const int bn = conv_split_bn(p.Cout);
  const long M = (long)p.B * p.Ho * p.Wo;
  p.wt_split_kind = 1; p.wt_split_bm = conv_split_bm(p.Cout); p.wt_split_bn = bn; p.splitk = 1; p.wt_split_kwr = 0;
  const int K = p.kh * p.kw * p.Cin + (p.in2 != nullptr ? p.Cin2 : 0);
  int b3, n3, k3;
  if (split3_fit(p, q, &b3, &n3, &k3)) {
    p.wt_split_kind = 3; p.wt_split_bm = b3; p.wt_split_bn = n3; p.splitk = k3;
    // stride-1 KH x 3 convs over rows of the output's pitch: the kw taps share a staged run of pixels
    p.wt_split_kwr = (q.kw_reuse && b3 == 256 && k3 == 1 && (n3 >= 128 || q.kwr_n64) && p.kw == 3 && p.stride == 1 && p.in_Wa == p.Wo &&
                      p.in2 == nullptr && 2 * p.dil <= 4 && p.Ho * p.Wo >= 256 && p.kh * 3 <= 30) ? 1 : 0;
    // fp16x2 pieces: tiles at least 128 wide whose source tensor(s) come with a recorded |max|
    if (q.family == 2 && (n3 >= 128 || q.h2_n64) && p.in_amax != nullptr && (p.in2 == nullptr || (p.in2_amax != nullptr && p.Cin2 % 32 == 0)) &&
        p.Cin % 32 == 0 && p.nlvl <= 1 && p.lvl_scale == nullptr && p.head_wt == nullptr) {
      const long t128 = ((M + 127) / 128) * (cout_padded(p.Cout) / 128);
      if (n3 == 64 && !p.wt_split_kwr) {
        // (a forced tile height took a 64-wide layer past the size rule below: same choice as there)
        if (q.h2_n64 && p.in2 == nullptr) {
          p.wt_split_kind = 2; p.wt_split_bm = 128; p.wt_split_bn = 64; p.splitk = 1;
          if (cout_padded(p.Cout) == 64 && q.h2_n64_bm512 == 2) p.wt_split_bm = 512;      // (tests only: see below)
        }
      } else if (b3 == 256 && (K >> 5) >= k3) {
        p.wt_split_kind = 2;
        // 64-wide kw-reuse layers (res2 conv2): 512 x 64 tiles, a 64 x 64 wave tile (a tile may cross ONE image boundary)
        if (n3 == 64 && p.wt_split_kwr && k3 == 1 && p.Ho * p.Wo >= 512 && cout_padded(p.Cout) == 64 &&
            (q.h2_n64_bm512 == 2 || (q.h2_n64_bm512 == 1 && (M + 511) / 512 >= q.min_tiles3)))
          p.wt_split_bm = 512;
        // (A/B knob, off: short reductions on 128 x 128 tiles, two workgroups per CU in different phases -- measured no gain:
        // res4 conv3 4.28 -> 4.38 ms, res2 / res3 conv3 and the laterals 3-12 % slower, profiles/r03_h2_small_tile_ab.txt)
        if (!p.wt_split_kwr && k3 == 1 && K <= q.h2s_maxk && t128 >= q.min_tiles3) { p.wt_split_bm = 128; p.wt_split_bn = 128; }
      } else if (b3 == 128 && n3 >= 256 && q.h2k_splitk && q.kw_reuse && p.kw == 3 && p.stride == 1 && p.in_Wa == p.Wo && p.in2 == nullptr &&
                 2 * p.dil <= 4 && p.Ho * p.Wo >= 256 && p.kh * 3 <= 30 && q.splitk_max > 1 && q.force_splitk <= 1 &&
                 [&]() {
                   // stride-1 KH x 3 layers of few rows (res3 / res4 conv2 at b = 1, the P5 3x3s at b = 8): the kw-reuse kernel on
                   // 256 x 128 tiles with the (slice, kh) groups cut into ranges -- a third of the activation-side work of the
                   // generic kernel's 128 x 128 split-K tiles (same-box A/B at b = 1, profiles/r04_b1_h2k_splitk_ab.txt: res4 conv2
                   // 83 -> 78 us per layer; the 128-wide res3 conv2 lost, 65 -> 74 us, and keeps the generic kernel)
                   const long t256 = ((M + 255) / 256) * (cout_padded(p.Cout) / 128);
                   const int groups = p.kh * (p.Cin >> 5);
                   int k = (int)((q.min_tiles3 + t256 - 1) / t256);
                   if (k > q.splitk_max) k = q.splitk_max;
                   while (k > 1 && groups / k < 3) --k;
                   // (round 6: the 256 x 128 kw-reuse tiles WITHOUT split-K where they alone fill the chip -- res5 conv2 at b = 8:
                   // 64 x 4 = 256 tiles, one round, a third of the generic kernel's activation-side work -- instead of its 128 x 128
                   // tiles: 0.503 -> 0.376 ms for the two layers, 320.3 -> 321.8 FPS same box, profiles/r06_res5_conv2_kwr_tiles_ab.txt;
                   // ODT_CONV_H2K_FEWROWS=0: A/B)
                   if (k <= 1 && q.h2k_fewrows && t256 >= q.min_tiles3) {
                     p.wt_split_kind = 2; p.wt_split_bm = 256; p.wt_split_bn = 128; p.wt_split_kwr = 1; p.splitk = 1;
                     return true;
                   }
                   if (k <= 1 || t256 * k < q.min_tiles3 / 2) return false;
                   p.wt_split_kind = 2; p.wt_split_bm = 256; p.wt_split_bn = 128; p.wt_split_kwr = 1; p.splitk = k;
                   return true;
                 }()) {
      } else if (b3 == 128 && n3 >= 256 && q.h2k_fewrows >= 2 && p.kh * p.kw == 1 && p.in2 == nullptr &&
                 ((M + 255) / 256) * (cout_padded(p.Cout) / 128) >= q.min_tiles3) {
        // (A/B, ODT_CONV_H2K_FEWROWS=2: the same for the dense 1x1 layers of few rows -- res5 conv1 -- on conv_h2_kernel<2, 4>)
        p.wt_split_kind = 2; p.wt_split_bm = 256; p.wt_split_bn = 128; p.splitk = 1; p.wt_split_kwr = 0;
      } else if (b3 == 128 && n3 >= 128 && q.h2_few_tiles) {
        // too few 256-row tiles (res5, P5 at b=8; everything below res3 at b=1): 128 x 128 tiles on 4 waves -- without split-K
        // where they fill the chip (res5 conv2 0.690 -> 0.465 ms, conv1 0.338 -> 0.219, same file), else with the reduction
        // cut into ranges (at least four 32-channel stages each)
        // (two of these workgroups share a CU: the chip has twice min_tiles3 slots for them -- res4 at b=1 ran its 128 tiles
        // x 2 ranges one 4-wave workgroup per CU, a single wave per SIMD)
        int k = 1;
        // 64 x 128 tiles (two waves, three workgroups per CU) fill the chip without cutting the reduction: no partial slabs,
        // no combine pass (b = 1: res4 has 128 x 2 ... 8 such tiles)
        const long t64 = ((M + 63) / 64) * (cout_padded(p.Cout) / 128);
        // (same-box A/B at b = 1, profiles/r04_b1_bm64_ab.txt: res4 conv1, K = 1024: 62.8 -> 44.4 us per layer; the K = 2304
        // 3x3 layers lose without split-K -- 72 serial stages: 81.7 -> 105.8 us -- and take the tiles with the reduction cut in two)
        if (q.h2_bm64 > 0 && t128 < q.min_tiles3 && t64 >= q.min_tiles3 && q.force_splitk <= 1 && (K <= 1024 || (q.h2_bm64 == 2 && p.in2 == nullptr))) {
          p.wt_split_kind = 2; p.wt_split_bm = 64; p.wt_split_bn = 128; p.splitk = K <= 1024 ? 1 : 2; p.wt_split_kwr = 0;
          // 64 x 64 tiles (conv_h2_kernel<1, 1>: twice the workgroups once more) for the reductions that stay whole: same box, b = 1,
          // res4 conv1 44.0 -> 41.7 us per layer, 171.6 -> 174.3 FPS (profiles/r04_b1_tiles64_ab.txt; taking 64 x 128 tiles also
          // where 128 x 128 ones give fewer than four workgroups per CU -- res4 conv3 -- lost: 33 -> 46 us per layer)
          if (q.h2_bm64 == 3 && K <= 1024) p.wt_split_bn = 64;
          return;
        }
        if (t128 < q.min_tiles3 && q.splitk_max > 1 && p.in2 == nullptr) {
          k = (int)((2 * q.min_tiles3 + t128 - 1) / t128);
          if (k > q.splitk_max) k = q.splitk_max;
          while (k > 1 && (K >> 5) / k < 4) --k;
        }
        if (q.force_splitk > 1 && p.in2 == nullptr && (K >> 5) >= q.force_splitk) k = q.force_splitk;
        if (t128 * k >= q.min_tiles3 / 2) { p.wt_split_kind = 2; p.wt_split_bm = 128; p.wt_split_bn = 128; p.splitk = k; p.wt_split_kwr = 0; }
      }
    }
    return;
  }
  // the 64-wide layers outside the kw-reuse kernel (conv0, res2 conv1): fp16x2 pieces on 128 x 64 tiles of 4 waves (three
  // workgroups per CU) instead of the one-stage bf16x3 loop
  if (q.family == 2 && q.h2_n64 && bn == 64 && p.in_amax != nullptr && p.in2 == nullptr && p.Cin % 32 == 0 && p.nlvl <= 1 &&
      p.lvl_scale == nullptr && p.head_wt == nullptr && p.kh * p.kw <= 32 && ((M + 127) / 128) * (cout_padded(p.Cout) / 64) >= q.min_tiles) {
    p.wt_split_kind = 2; p.wt_split_bm = 128; p.wt_split_bn = 64;
    // 512 x 64 tiles on 8 waves stacked along M for these layers (conv0, res2 conv1): built, tested, and NOT the default --
    // same-box A/B at b=8 1080p: conv0 0.767 -> 0.888 ms, res2 conv1 0.555 -> 0.587 (three 4-wave workgroups per CU in different
    // phases hide these HBM-bound layers' loads and stores better than one 8-wave workgroup; profiles/r04_n64_bm512_ab.txt)
    if (cout_padded(p.Cout) == 64 && q.h2_n64_bm512 == 2) p.wt_split_bm = 512;
  }
}

size_t conv_split_partial_bytes(const ConvParams& p) {
  return (p.wt_split_kind == 3 || p.wt_split_kind == 2) && p.splitk > 1 ? (size_t)p.splitk * p.B * p.Ho * p.Wo * cout_padded(p.Cout) * sizeof(float) : 0;
}

// f32 weights [Cout][K] times a per-k gate (the exact-f32 kernel's form of the folded gate)
__global__ void __launch_bounds__(256) scale_weights_kernel(const float* __restrict__ wt, const float* __restrict__ kscale, long total, int K,
                                                            float* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < total) out[i] = wt[i] * kscale[(int)(i % K)];
}

int conv_scale_weights(const float* wt, const float* kscale, int Cout, int K, float* out, hipStream_t stream) {
  const long total = (long)Cout * K;
  hipLaunchKernelGGL(scale_weights_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, wt, kscale, total, K, out);
  ODT_HIP(hipGetLastError());
  return 0;
}

int conv_make_split_weights(const ConvParams& p, void* img_dev, hipStream_t stream, const float* wt_src, const float* kscale) {
  const float* src = wt_src != nullptr ? wt_src : p.wt;
  ODT_CHECK(kscale == nullptr || (p.kh == 1 && p.kw == 1 && p.in2 == nullptr), "conv_make_split_weights: a folded gate belongs to a single-source 1x1 conv");
  const int bn = p.wt_split_bn != 0 ? p.wt_split_bn : conv_split_bn(p.Cout);
  const int K = p.kh * p.kw * p.Cin + (p.in2 != nullptr ? p.Cin2 : 0);
  ODT_CHECK(bn != 0 && K % 32 == 0 && (p.wt_split_kind >= 1 && p.wt_split_kind <= 3),
            "conv_make_split_weights: Cout % 64 == 0, K % 32 == 0 and a chosen kernel family required");
  if (p.wt_split_kind == 2) {
    ODT_CHECK(kscale == nullptr && wt_src == nullptr, "conv_make_split_weights: the fp16x2 image takes the conv's own weights");
    return conv_make_h2_weights(p, img_dev, stream);
  }
  if (p.wt_split_kind == 3) {
    const long total = (long)cout_padded(p.Cout) * (K >> 4) * 2;
    hipLaunchKernelGGL(split_weights3_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, src, p.Cout, K,
                       bn, p.kh * p.kw, p.Cin, (unsigned short*)img_dev, kscale);
  } else {
    const long total = (long)cout_padded(p.Cout) * (K >> 3);
    hipLaunchKernelGGL(split_weights_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, src, p.Cout, K,
                       bn, 32, (unsigned short*)img_dev, kscale);
  }
  ODT_HIP(hipGetLastError());
  return 0;
}

int launch_conv_split(const ConvParams& p, const ConvParams* dev, hipStream_t stream) {
  if (p.wt_split_kind == 2) return launch_conv_h2(p, dev, stream);
  return p.wt_split_kind == 3 ? launch_conv_split3(p, dev, stream) : launch_conv_split1(p, dev, stream);
}

}  // namespace odt
