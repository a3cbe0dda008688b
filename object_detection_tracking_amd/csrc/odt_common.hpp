// Shared declarations for libodt_hip.so (gfx950).  Kernel launch wrappers are
// declared here and defined in the per-kernel .hip translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <vector>

#include <cstdint>
#include <cstdio>
#include <string>

#include "knobs.hpp"

namespace odt {

void set_error(const std::string& msg);

#define ODT_HIP(expr)                                                          \
  do {                                                                         \
    hipError_t odt_e_ = (expr);                                                \
    if (odt_e_ != hipSuccess) {                                                \
      ::odt::set_error(std::string(#expr) + ": " + hipGetErrorString(odt_e_)); \
      return 1;                                                                \
    }                                                                          \
  } while (0)

#define ODT_CHECK(cond, msg)                 \
  do {                                       \
    if (!(cond)) {                           \
      ::odt::set_error(std::string(msg));    \
      return 1;                              \
    }                                        \
  } while (0)

typedef float f32x4 __attribute__((vector_size(16)));
typedef float f32x16 __attribute__((vector_size(64)));

constexpr int kMaxTopK = 1024;   // rpn_test_post_nms_topk handled by the one-candidate-per-thread selection kernels
constexpr int kMaxTopKBig = 4096;// rpn_test_post_nms_topk upper bound (above kMaxTopK: the *_big kernels, paneled NMS)
#ifndef ODT_AMAX_WAYS
#define ODT_AMAX_WAYS 1
#endif
// words per tensor range slot (ConvParams::in_amax / out_amax; conv_split_common.hpp amax_read / amax_way).  1: one word per
// tensor.  Round 5 measured 16 (-DODT_AMAX_WAYS=16, tools/ab_build_all.sh: a sixteenth of the same-address atomics when a
// launch's workgroups finish together): b = 8 310.0 -> 308.7 FPS, b = 1 174.0 -> 164.0 (profiles/r05_amax_ways_ab.txt) -- the
// sixteen reads in front of every consumer's first load cost more than the atomics they spare.  Not kept.
constexpr int kAmaxWays = ODT_AMAX_WAYS;
// Round 6 (the continuous fp16x2 range guard): a plan's slot array is [live |max| : kRangeSlots][previous forward's |max| : kRangeSlots];
// amax_rotate_kernel (runtime.hip; the former memset at the start of a forward) moves a group's records to the second row and to
// host-visible memory, where odt_range_health compares them with the values the host last accepted -- no kernel code beyond what
// recorded the maxima all along (counters in the producers were built and cost 1 %: tools/experiments/r06_range_counters).
constexpr int kRangeSlots = 1024;
constexpr int kRoiOut = 7;       // ROIAlign output side (models.py:703)
constexpr int kRpnCh = 16;       // 3 logits + 12 deltas (+1 pad) per pixel
constexpr int kSelChunk = 32768; // logits per workgroup in stage 1 of the RPN top-k
int proposal_total_chunks(const struct ProposalParams& p);

// ---------------------------------------------------------------- conv (K2+K3)
// Implicit-GEMM convolution, NHWC fp32, exact-f32 MFMA (v_mfma_f32_32x32x2_f32).
//   out[n,ho+oy,wo+ox,co] = act( sum_{kh,kw,ci} in[n,ho*s+kh*d-pt, wo*s+kw*d-pl, ci]
//                                * wt[co,kh,kw,ci] + bias[co] (+ residual) )
struct ConvParams {
  const float* in;     // [B,H,W,in_ldc]  (first Cin channels read)
  const float* wt;     // [Cout][kh][kw][Cin]  (K contiguous), BN folded
  const float* bias;   // [Cout]
  const float* res;    // residual (see res_mode) or nullptr
  float* out;          // [B,out_H,out_W,out_ldc]
  int B, H, W, Cin, in_ldc;   // H,W: logical input extent (reads outside are zero)
  int in_Ha, in_Wa;           // allocation extent of `in` (row pitch); >= H,W (sliced views)
  int Ho, Wo, Cout;
  int kh, kw, stride, dil, pad_t, pad_l;
  int out_H, out_W, out_oy, out_ox, out_ldc;
  int res_mode;        // 0 none | 1 same shape | 2 nearest-2x upsample of [B,res_H,res_W,*]
  int res_H, res_W, res_ldc;
  int relu;            // epilogue activation: 0 none | 1 ReLU | 2 swish x*sigmoid(x) | 3 sigmoid
  // optional second A source, K-concatenated behind the first (1x1 convs only): the stage-entry
  // bottleneck computes conv3(t2) + convshortcut(x) as ONE GEMM over [t2 | x(::stride2)]
  const float* in2;    // [B,in2_Ha,in2_Wa,in2_ldc] or nullptr
  int Cin2, in2_ldc, in2_Ha, in2_Wa, in2_stride;
  unsigned div_howo_mul, div_howo_sh, div_wo_mul, div_wo_sh;   // exact m / (Ho*Wo), r / Wo by multiply-shift (conv_prepare)
  int debug;           // ablation bits for kernel tuning (0 in production; ODT_CONV_DEBUG)
  unsigned long long* trace;   // optional [grid][8] wall-clock phase stamps (tuning only)
  // optional bf16-piece image of `wt` (conv_make_split_weights): the layer runs on the bf16x3
  // split kernel (conv_split.hip: f32 result through six exact bf16 MFMA products per MAC)
  const void* wt_split;
  int wt_split_kind;   // which kernel family the image was laid out for: 1 one-stage BK = 32 (conv_split_kernel, the 64-wide
                       // layers) | 3 conv_split3_kernel / conv_split3k_kernel (8 waves, LDS-DMA weight stages) | 2 conv_h2_kernel /
                       // conv_h2k_kernel (fp16x2 pieces)
  int wt_split_bm;     // kind 3: rows of the block tile (256, or 128 when 256-row tiles would not fill the chip)
  int wt_split_bn;     // n-tile width the image was laid out for (kind 3 may use 128 on wider layers; 0: conv_split_bn(Cout))
  // split-K (conv_split3_kernel only): the reduction is cut into `splitk` contiguous ranges of stages, one workgroup per
  // (tile, range); each writes its raw f32 partial tile to partial[range][M][Cout] and split_reduce_kernel adds them in
  // range order (deterministic) and applies bias / residual / activation.  For layers whose tiles cannot fill the chip.
  int wt_split_kwr;    // kind 3: 1 = conv_split3k_kernel (the three kw taps of a (slice, kh) group share one staged run of pixels)
  int splitk;          // 0 / 1: off
  float* partial;      // scratch [splitk][M][Cout] (plan-owned, shared by the plan's split-K layers)
  // optional fused 1x1 head behind this conv (conv_split3 kernels whose n-tile is the whole Cout = 256: the RPN 3x3 conv +
  // ReLU, models.py:979-1009): head_out[m][j] = sum_c act(conv)[m][c] * head_wt[c][j] + head_bias[j], j < 16, evaluated on
  // the staged C tile in the epilogue with exact-f32 MFMA (v_mfma_f32_16x16x4_f32) -- the 256-channel tensor is never
  // written (out == nullptr) or read back, and the separate N = 15 launch disappears
  // optional per-row-range epilogue constants (conv_split3 kernels without split-K): rows [lvl_start[i], lvl_start[i+1])
  // of the GEMM take bias[i * lvl_stride + c] and, if given, the factor lvl_scale[i * lvl_stride + c] in front of it
  // (out = act(acc * scale + bias)).  The EfficientDet class / box nets share their conv weights between the five pyramid
  // levels but carry one BatchNorm per level (efficientdet_arch.py:227-393): with the levels' pixels concatenated (each
  // padded to whole 256-row tiles) a layer is ONE launch instead of five.
  int nlvl;            // 0 / 1: off
  int lvl_start[5];
  int lvl_stride;
  const float* lvl_scale;
  // fp16x2 pieces (wt_split_kind == 2, conv_h2.hip): the A operand is scaled by a power of two taken from the recorded |max|
  // of the source tensor(s) -- a u32 holding the f32 bit pattern, written by the producers' epilogues (out_amax: atomic max
  // over the stored values; the plan clears the slots at the start of a forward) -- and the weight image's column n by
  // 2^t_n; the epilogue multiplies the accumulators by h2_chinv[n] = 2^-t_n and by the inverse of the A scale.
  const unsigned* in_amax;   // |max| of `in` (required for kind 2)
  const unsigned* in2_amax;  // ... of `in2`
  unsigned* out_amax;        // where this conv records the |max| of what it stores (any split kernel), or nullptr
  const float* h2_chinv;     // [cout_padded(Cout)] (kind 2)
  const float* head_wt;    // [Cout][16] (k-major, column 15 zero) or nullptr
  const float* head_bias;  // [16]
  float* head_out;         // [M][head_ldc] dense rows (m = (n, ho, wo))
  int head_ldc;
  // optional fused 1x1 conv behind this conv (conv_h2k_kernel<.., FUSE>: the bottleneck's conv2 3x3 + BN + ReLU followed by
  // conv3 1x1 + BN (+ shortcut) + ReLU, nn.py:503-521): the tile's accumulators never leave the CU -- scaled, biased and
  // activated in registers, split into fp16x2 pieces with a power of two PER PIXEL ROW (not per tensor), they are the
  // second GEMM's operand as they sit (MFMA C layout == operand layout under a k permutation that the weight image
  // f_wt carries); f_out[m][n] = act(sum_k y[m][k] w[n][k] + f_bias[n] (+ f_res[m][n])), rows m dense over (n, ho, wo).
  // `out` is not written (nullptr) and the 1x1 conv's own launch disappears.
  const void* f_wt;        // conv_make_h2f_weights image of the 1x1 conv (K = this conv's Cout), or nullptr
  const float* f_chinv;    // [f_cout] inverse powers of two of the image's rows
  const float* f_bias;     // [f_cout]
  const float* f_res;      // [M][f_res_ldc] same-shape residual or nullptr
  float* f_out;            // [M][f_out_ldc]
  unsigned* f_out_amax;    // range slot of f_out or nullptr
  int f_cout, f_out_ldc, f_res_ldc, f_relu;
  // conv0 + pool0 in one kernel (conv_stem.hip; fuse_stem): `out` is the 3x3 / stride-2 max-pooled map [B, out_H, out_W, out_ldc]
  // of the conv's [B, Ho, Wo, Cout] result, which is not written
  int stem_pool;
};
// fills the derived fields (multiply-shift divisors); call before copying a record to the device
void conv_prepare(ConvParams& p);
// dev_params: device copy of `p` (plan-owned); nullptr = stage a temporary (stand-alone calls)
int launch_conv(const ConvParams& p, hipStream_t stream, const ConvParams* dev_params = nullptr);
double conv_flops(const ConvParams& p);   // algorithmic 2*M*N*K
// bf16x3 split path (conv_split.hip).  ConvPolicy: which convs take it and which kernel family -- per handle, fixed at
// odt_create from odt_config (+ ODT_CONV_* debug overrides, conv_policy_from_env); the stand-alone test entry points
// resolve it per call.
struct ConvPolicy {
  int arith;            // 0 exact-f32 MFMA everywhere | 1 bf16x3 split kernels where they pay
  int family;           // split kernel families allowed: 1 one-stage only | 3 + conv_split3_kernel where it fits | 2 + the
                        // fp16x2 kernels (conv_h2.hip) where a layer has 256-row tiles and a recorded input range
  long min_tiles;       // tiles a layer must offer the one- / two-stage kernels (256)
  long min_tiles3;      // ... conv_split3_kernel's 256- / 128-row tiles (200)
  int min_k, min_bn;    // shortest reduction / narrowest n-tile taken
  int h2s_maxk;         // fp16x2: longest reduction that takes the 128 x 128 4-wave tile (two workgroups per CU); 0: none
  bool h2_few_tiles;    // fp16x2: 128 x 128 tiles for the layers without enough 256-row tiles
  bool h2_n64;          // fp16x2: also the 64-wide layers (128 x 64 tiles on 4 waves; the kw-reuse kernel's 256 x 64 tile)
  bool h2k_splitk;      // fp16x2: kw-reuse kernel + split-K over (slice, kh) groups for the KH x 3 layers of few rows
  int h2k_fewrows;      // fp16x2: 1 = kw-reuse kernel on 256 x 128 tiles without split-K where those alone fill the chip; 2 (A/B): the dense 1x1 layers too
  int fill_div;         // a split-K layer is taken when tiles x ranges >= min_tiles3 / fill_div (6)
  int h2_bm64;          // fp16x2: 64 x 128 two-wave tiles for the layers of few rows: 0 off | 1 K <= 1024, no split-K | 2 + longer K cut in two | 3 = 1 on 64 x 64 tiles
  int h2_n64_bm512;     // fp16x2 kw-reuse kernel on 64-wide layers: 512 x 64 tiles -- 0 off | 1 where they fill the chip | 2 wherever valid
  int force_bm3;        // 0 auto | 128 | 256: force conv_split3_kernel with that tile height (tests)
  int splitk_max;       // conv_split3_kernel: largest split-K factor the policy may choose (1 = off)
  bool kw_reuse;        // conv_split3k_kernel for the stride-1 KH x 3 layers it fits (ODT_CONV_SPLIT3_KWR=0: off)
  bool kwr_n64;         // ... also for 64-wide layers (256 x 64 tile, wave tile 64 x 32)
  int force_splitk;     // 0 auto | k: force that split-K factor wherever conv_split3_kernel runs (tests)
  bool src2, res2;      // take the K-concatenated stage-entry convs / the 2x-upsampled-residual FPN laterals
};
ConvPolicy conv_policy_default();
ConvPolicy conv_policy_from_env(ConvPolicy q);
bool conv_split_supported(const ConvParams& p);
// the layer is supported AND large enough to fill the chip with the split tiles
bool conv_split_wanted(const ConvParams& p, const ConvPolicy& q);
size_t conv_split_weight_bytes(int Cout, int K);
int conv_split_bn(int Cout);   // n-tile width of the split configuration for this Cout (0: none)
int conv_split_bm(int Cout);
// picks the split kernel family / tile for a conv that conv_split_wanted() accepted (fills wt_split_kind / wt_split_bm)
void conv_split_choose(ConvParams& p, const ConvPolicy& q);
// builds the bf16-piece image of p.wt for p.wt_split_kind (conv_split_choose first)
// wt_src: source weights if not p.wt; kscale[K]: per-k factor folded into the weights first (single-source 1x1 convs: a
// per-input-channel gate, e.g. squeeze-excite, applied to the weights instead of the activations)
int conv_make_split_weights(const ConvParams& p, void* img_dev, hipStream_t stream, const float* wt_src = nullptr,
                            const float* kscale = nullptr);
int conv_scale_weights(const float* wt, const float* kscale, int Cout, int K, float* out, hipStream_t stream);
int launch_conv_split(const ConvParams& p, const ConvParams* dev, hipStream_t stream);
// fp16x2 pieces (conv_h2.hip; wt_split_kind == 2): the image (conv_make_split_weights builds it) carries the per-column
// inverse scales behind the pieces -- ConvParams::h2_chinv must point there; launch_tensor_amax records the |max| of a dense
// array for a source tensor no producer recorded one for (stand-alone calls)
const float* conv_h2_chinv(const void* img, int Cout, int K);
int conv_make_h2_weights(const ConvParams& p, void* img_dev, hipStream_t stream);
int launch_tensor_amax(const float* x, size_t n, unsigned* slot, hipStream_t stream);
// fused 1x1 conv behind a conv_h2k launch (ConvParams::f_wt): image of wt [Cout][K] (K = 128 or 256: the producer's
// Cout), chunks of 32 output columns, k in the order the producer's accumulator registers hold it; the rows' inverse
// powers of two sit behind the pieces (conv_h2f_chinv)
size_t conv_h2f_weight_bytes(int Cout, int K);
const float* conv_h2f_chinv(const void* img, int Cout, int K);
int conv_make_h2f_weights(const float* wt, int Cout, int K, void* img_dev, hipStream_t stream);
bool conv_stem_fits(const ConvParams& p);          // conv_stem.hip: the plan's conv0 on the fp16x2 family
int launch_conv_stem(const ConvParams& p, const ConvParams* dev, hipStream_t stream);
bool conv_h2f_fusable(const ConvParams& a, const ConvParams& b);   // a: the KH x 3 producer, b: the 1x1 conv reading a.out
size_t conv_split_partial_bytes(const ConvParams& p);   // scratch a split-K conv needs (0: none)

// ------------------------------------------------------------ elementwise (K1,K4)
// (amax: optional range slot of the output tensor -- the |max| of what is written goes there, see ConvParams::in_amax)
int launch_preprocess(const void* frames, int dtype, int B, int H, int W, int pad_t, int pad_l,
                      int Hp, int Wp, float* out, hipStream_t stream, unsigned* amax = nullptr);
int launch_preprocess_resize(const void* frames, int dtype, int B, int Hs, int Ws, int H, int W, int pad_t,
                             int pad_l, int Hp, int Wp, float* out, hipStream_t stream, unsigned* amax = nullptr);
int launch_maxpool3x3s2(const float* in, int B, int H, int W, int C, float* out, int Ho, int Wo,
                        hipStream_t stream);

// ------------------------------------------------------- EfficientNet blocks (effnet.hip)
struct DwConvParams {
  const float* in;     // [B,H,W,ldc]
  const float* wt;     // [k*k][ldc]  (BN scale folded, pad channels zero)
  const float* bias;   // [ldc]
  float* out;          // [B,Ho,Wo,ldc]
  int B, H, W, Ho, Wo, ldc, k, stride, pad_t, pad_l;
  int act;             // 0 none | 2 swish
  // optional fused squeeze (MBConv): per-workgroup sums of the OUTPUT per (image, channel) for the squeeze-excite gate,
  // sum_part[b][split][c] (fixed order); channel_mean_fold_kernel (launch_se_gate_from_parts) adds the splits
  float* sum_part;     // [B][dwconv_splits()][ldc] or nullptr
  int cqn, nsplit;     // filled by launch_dwconv: channel quads per workgroup (16), pixel splits
  int xcd_bands;       // filled by launch_dwconv: 1 = contiguous band of workgroups per XCD (halo rows shared in its L2)
  // optional: several independent [H,W] maps in one launch (batch 1, stride 1 'SAME': the five pyramid levels of a class /
  // box net layer): map i reads lin[i] and writes lout[i], both [lH[i], lW[i], ldc]; grid z = map
  int nlvl;            // 0: off
  const float* lin[5];
  float* lout[5];
  int lH[5], lW[5];
};
int launch_dwconv(const DwConvParams& p, hipStream_t stream);
int dwconv_splits(const DwConvParams& p);       // pixel splits launch_dwconv will use (sizes sum_part)
// MBConv front half in one kernel (effnet_mbconv.hip): expand 1x1 + BN + swish -> depthwise k x k + BN + swish (+ squeeze
// partial sums); the expanded tensor stays in LDS
struct MbExpandDwParams {
  const float* x;        // block input [B,H,W,in_ldc]; K of the expand GEMM = in_ldc (pad channels: zero weights)
  int B, H, W, in_ldc;
  const void* w_img;     // bf16x3 piece image of the expand weights [lmid][in_ldc] (conv_make_split_weights, kind 1, n-tile 64)
  const float* e_bias;   // [mid] folded BN shift of the expand conv
  int mid, lmid;         // expanded channels, their stride (multiple of 64)
  const float* dw_wt;    // [k*k][lmid]  (BN scale folded, pad channels zero)
  const float* dw_bias;  // [lmid]
  float* out;            // [B,Ho,Wo,lmid]
  int Ho, Wo, k, stride, pad_t, pad_l;
  float* sum_part;       // [B][nsplit][lmid] per-workgroup sums of the output (squeeze-excite), or nullptr
  int nsplit;            // tile ranges per image (0: launch_mbconv_expand_dw picks mbconv_expand_dw_splits())
  int tiles_y, tiles_x;  // filled by the launcher
};
int launch_mbconv_expand_dw(const MbExpandDwParams& p, hipStream_t stream);
int mbconv_expand_dw_splits(const MbExpandDwParams& p);
size_t mbconv_expand_weight_bytes(int lmid, int in_ldc);
struct FuseParams {
  const float* in[3];
  int ih[3], iw[3], mode[3];     // mode 0 same size | 1 nearest resize | 2 max-pool 3x3 s2 'SAME'
  float sy[3], sx[3];            // mode 1: in / out size ratios
  int pt[3], pl[3];              // mode 2: 'SAME' pads before
  float wgt[3], denom;           // weighted: relu(WSM_i), sum + 1e-4
  int n, weighted, act;          // act 0 none | 2 swish
  int B, h, w, ldc;
  float* out;
};
int launch_bifpn_fuse(const FuseParams& p, hipStream_t stream);
int launch_preprocess_rgb(const void* frames, int dtype, int B, int H, int W, int pad_t, int pad_l, int Hp, int Wp,
                          float* out, hipStream_t stream);
int channel_mean_splits(int HW, int ldc, int B);
int launch_preprocess_rgb_resize(const void* frames, int dtype, int B, int Hs, int Ws, int Hr, int Wr, int pad_t,
                                 int pad_l, int Hp, int Wp, float* out, hipStream_t stream);
int launch_channel_mean(const float* in, int B, int HW, int ldc, float* scratch, float* out, hipStream_t stream);
int launch_channel_scale(float* x, const float* s, int B, int HW, int ldc, hipStream_t stream);
struct SeGateParams {
  const float* part; int nsplit;      // filled by launch_se_gate (launch_se_gate_from_parts: by the caller)
  int HW, ldc, mid, se;
  const float* w1;   // [se][ldc]   reduce weights (pad channels zero)
  const float* b1;   // [se]
  const float* w2t;  // [se][ldc]   expand weights, transposed
  const float* b2;   // [mid]
  float* gate;       // [B][ldc]    (pad channels stay 0)
  float* r;          // [B][256]    reduced vector (scratch)
  float* mean;       // [B][ldc]    spatial mean (scratch)
};
int launch_se_gate(const float* in, const SeGateParams& p, int B, float* scratch, hipStream_t stream);
// the gate from the depthwise kernel's fused partial sums (p.part / p.nsplit set by the caller): fold + reduce + expand
int launch_se_gate_from_parts(const SeGateParams& p, int B, hipStream_t stream);

// ------------------------------------------------------- EfficientDet tail (effdet_post.hip)
struct EffPostParams {
  const float* cls[5];     // per level [B, npix, ldc_cls]  (9 * ncls valid channels: anchor-major)
  const float* box[5];     // per level [B, npix, ldc_box]  (36 valid)
  int npix[5], anchor_off[6];   // anchors before level l; anchor_off[5] = total
  int ldc_cls, ldc_box, ncls, B;
  const float* anchors;    // [total, 4] y1,x1,y2,x2
  int k, max_out;
  float score_thresh, iou_thresh, image_scale;
  unsigned* keys;          // scratch [total * ncls]
  unsigned* hist;          // scratch [256]
  unsigned* state;         // scratch [4]
  unsigned long long* sel; // scratch [B, k]
  float* cand_boxes; float* cand_scores; int* cand_cls; int* cand_lvl;   // [B, k(,4)]
  float* out_boxes; float* out_scores; int* out_labels; int* out_levels; int* out_valid;   // [B, max_out(,4)], [B]
};
int launch_effdet_post(const EffPostParams& p, hipStream_t stream);

// ------------------------------------------------------- proposals (K6,K7,K8)
struct RpnLevel {
  const float* rpn;      // [B,h,w,kRpnCh]: ch 0..2 logits, 3+a*4+c deltas
  const float* anchors;  // [field,field,3,4]
  int h, w, field;
};
struct ProposalParams {
  RpnLevel lvl[5];
  int nlevels;
  int graph;             // ODT_GRAPH_SINGLE / MULTI semantics
  int B, K;
  int img_h, img_w;
  float nms_thresh, decode_clip;
  // workspace (device): per (b,level) sorted candidates and survivors
  float* cand_boxes;     // [B,L,K,4]
  float* cand_scores;    // [B,L,K]
  int* cand_count;       // [B,L]
  float* lvl_boxes;      // [B,L,K,4]
  float* lvl_scores;     // [B,L,K]
  int* lvl_count;        // [B,L]
  unsigned long long* chunk_keys;   // [B, total_chunks, K] per-chunk top-K keys (stage 1 of the select)
  // outputs
  float* props;          // [B,K,4]
  int* nprops;           // [B]
};
size_t proposal_workspace_bytes(int B, int L, int K);
int launch_proposals(const ProposalParams& p, hipStream_t stream);
int launch_topk(const float* scores, int n, int k, int* idx_out, hipStream_t stream);
int launch_nms(const float* boxes, const float* scores, int n, int max_out, float thresh,
               int* idx_out, int* n_out, hipStream_t stream);

// --------------------------------------------------------------- ROIAlign (K9,K14)
struct RoiAlignParams {
  const float* feat[5];  // NHWC [B,h,w,C] (sliced dims h,w ; pixel stride ldc); 5th level: EfficientDet P7
  int h[5], w[5], ldc[5], alloc_h[5], alloc_w[5];
  float inv_stride[5];
  const int* levels;     // optional [R_cap]: pyramid level of each box (value - level0 indexes feat[]); nullptr: FPN rule
  int level0;
  int C;
  const float* boxes;    // [R_cap,4] image coords
  const int* box_ind;    // [R_cap] or nullptr (then box r belongs to image r / per_image)
  int per_image;         // rows per image when box_ind == nullptr
  const int* count;      // device: number of valid rows per image [B] (nullptr: all R_cap)
  int R_cap;
  float* out_nhwc;       // [R_cap,7,7,C] or nullptr   (box-head input order h,w,c)
  float* out_nchw;       // [R_cap,C,7,7] or nullptr   (fpn_box_feat)
  float* pooled;         // [R_cap,C] or nullptr
  int out_size;          // 0 / 7: box head + features; 14: mask head (models.py:935-936)
  int pack_rows;         // with count: 1 = output rows packed over the valid rows of all images (final features, masks);
                         // 0 = row r stays row r, rows past count[b] untouched (box head: its consumers index b * per_image + j)
  unsigned* amax;        // optional range slot of out_nhwc (round 6: fc6 reads the RoI features on the fp16x2 kernels, which scale by the
                         // tensor's |max|).  With it the rows past count[b] are written as zeros instead of left stale: the recorded
                         // maximum must cover every row the consumer's GEMM reads
};
int launch_roi_align(const RoiAlignParams& p, hipStream_t stream);

// ------------------------------------------------------- mask head tail (models.py:951-962)
struct MaskSelectParams {
  const float* logits;   // [R_cap,14,14,4,ld]: deconv sub-pixel (dy,dx) major, class minor
  int ld;                // padded class stride
  const int* labels;     // [R_cap] 1-based foreground labels (device)
  const int* valid;      // [B] rows per image
  int B, per_image;
  float* masks;          // [R_cap,28,28]
};
int launch_mask_select(const MaskSelectParams& p, hipStream_t stream);

// ------------------------------------------------------- detection tail (K11,K12,K13)
struct DetectParams {
  int graph, B, K, C;          // C = num_class incl. BG
  const float* head_out;       // [B*K, ld] : cols [0,C) class logits, [C, C+4C) box logits
  int ld;
  const float* props;          // [B,K,4]
  const int* nprops;           // [B]
  int img_h, img_w;
  float reg_w[4];
  float decode_clip, score_thresh, nms_thresh;
  int per_im;
  // workspace
  float* dec_boxes;            // [B*K, C-1, 4]
  float* probs;                // [B*K, C]
  int* cls_keep;               // [B, C-1, per_im]
  int* cls_count;              // [B, C-1]
  // outputs (device)
  float* out_boxes;            // [B, per_im, 4]
  float* out_probs;            // [B, per_im]
  int* out_labels;             // [B, per_im]
  int* out_valid;              // [B]
};
int launch_detections(const DetectParams& p, hipStream_t stream);
int launch_class_nms(const DetectParams& p, hipStream_t stream);

// ----------------------------------------------------------------- tracker (K15)
// blocks: nn_cosine_blocks' table of <= 32-row gallery blocks (device copy); any_part: a track spans several blocks
bool nn_cosine_blocks(const int* seg, int T, std::vector<int>* blocks);
int launch_nn_cosine(const float* gallery, const int* seg, const int* blocks, int nblocks, bool any_part, int T, const float* dets, int N,
                     int D, double* cost, hipStream_t stream);
// host-to-host cosine nearest-neighbour call with persistent scratch and a stream of its own (tracker.hip)
struct CosineCtx {
  int device = -1;
  hipStream_t stream = nullptr;
  hipEvent_t done = nullptr;
  float* h_in = nullptr; float* d_in = nullptr; size_t cap_in = 0;        // packed [seg | gallery | detections]
  double* h_cost = nullptr; double* d_cost = nullptr; size_t cap_cost = 0;
  std::vector<int> blocks_scratch;                   // the call's block table (host)
  std::vector<void*> retired_host, retired_dev;      // outgrown buffers: released with the context (hipFree waits for the device)
  ~CosineCtx();
  // gal_rows[G] / det_rows[N]: pointers to the D-float rows (gathered into the pinned record); seg[T+1]; cost[T*N]
  int run(int dev, const float* const* gal_rows, int G, const int* seg, int T, const float* const* det_rows, int N,
          int D, double* cost);
};

}  // namespace odt
