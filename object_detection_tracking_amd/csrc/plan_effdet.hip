// EfficientDet plan builder (plan helpers: plan_common.hip -- Tensor, Op, odt_model, make_tensor, find_w, upload_raw,
// add_conv).
#include "odt_model.hpp"

#define g_err (::odt::last_error())

namespace odt {

// ------------------------------------------------------------------------------------------------
// EfficientNet backbone plan (EfficientDet path, SURVEY.md 8f rank 3 -- detector half in progress).
// reference efficientdet/backbone/efficientnet_builder.py:37-53,162-168, efficientnet_model.py:137-159,
// :162-330, :520-650.  Tensors are NHWC with the channel stride padded to a multiple of 32 (zero pad
// channels) so that every 1x1 conv feeds conv_igemm_kernel; BN (epsilon 1e-3) is folded on the host.
// channel stride of a tensor: padded with zero channels to a multiple of 32 -- or of 64 when the bf16x3 split kernels
// take the 1x1 convs (their n-tile granule; conv_split.hip cout_padded)
static bool eff_split_on() {        // (read per plan build: tests and A/B runs flip it inside one process)
  return !env_knob_off(K_EFFDET_SPLIT);
}
int r32(int c) { return eff_split_on() ? (c + 63) / 64 * 64 : (c + 31) / 32 * 32; }
// squeeze-excite gate folded into the projection's weights at batch 1 (ODT_EFFDET_WSCALE=0: a pass over the activations)
// MBConv front half (expand 1x1 -> depthwise) as one kernel, the expanded tensor kept in LDS (effnet_mbconv.hip):
// ODT_EFFDET_FUSE_MB = 0 off | 1 (default) blocks whose depthwise output is at least 64 pixels on its short side (smaller
// maps do not fill 16 x 16 patches: the halo / partial-tile recompute would cost more than the launch it saves) | 2 every block
static int eff_fuse_mb_mode() {
  return (int)env_knob_long(K_EFFDET_FUSE_MB, 1);
}
static int eff_fuse_mb_min() {           // A/B knob: smallest short side of the depthwise output that is fused in mode 1
  const long v = env_knob_long(K_EFFDET_FUSE_MB_MIN, 64);
  return v > 0 ? (int)v : 64;
}
static bool eff_wscale_on() {
  return !env_knob_off(K_EFFDET_WSCALE);
}

int eff_round_filters(int f, double width) {
  const double x = f * width;
  int nf = std::max(8, (int)(x + 4) / 8 * 8);
  if (nf < 0.9 * x) nf += 8;
  return nf;
}

struct EffBlock { int idx, kernel, stride, expand, cin, cout, se, reduction; };

std::vector<EffBlock> eff_blocks(int variant, int* stem) {
  static const double WD[8][2] = {{1.0, 1.0}, {1.0, 1.1}, {1.1, 1.2}, {1.2, 1.4}, {1.4, 1.8}, {1.6, 2.2}, {1.8, 2.6}, {2.0, 3.1}};
  static const int BL[7][6] = {{1, 3, 1, 1, 32, 16}, {2, 3, 2, 6, 16, 24}, {2, 5, 2, 6, 24, 40}, {3, 3, 2, 6, 40, 80},
                               {3, 5, 1, 6, 80, 112}, {4, 5, 2, 6, 112, 192}, {1, 3, 1, 6, 192, 320}};
  const double w = WD[variant][0], d = WD[variant][1];
  *stem = eff_round_filters(32, w);
  std::vector<EffBlock> v;
  for (const auto& b : BL) {
    const int cin = eff_round_filters(b[4], w), cout = eff_round_filters(b[5], w);
    const int reps = (int)std::ceil(d * b[0] - 1e-9);
    for (int r = 0; r < reps; ++r) {
      EffBlock e;
      e.idx = (int)v.size(); e.kernel = b[1]; e.stride = r == 0 ? b[2] : 1; e.expand = b[3];
      e.cin = r == 0 ? cin : cout; e.cout = cout; e.se = std::max(1, (int)(e.cin * 0.25)); e.reduction = 0;
      v.push_back(e);
    }
  }
  int red = 0;
  for (size_t i = 0; i < v.size(); ++i)
    if (i + 1 == v.size() || v[i + 1].stride > 1) v[i].reduction = ++red;
  return v;
}

// BN fold factors of a Keras BatchNormalization scope (gamma, beta, moving_mean, moving_variance)
int eff_bn(odt_model* m, const std::string& scope, int c, std::vector<double>* scale, std::vector<double>* shift) {
  const HostTensor* g = find_w(m, scope + "/gamma"); const HostTensor* b = find_w(m, scope + "/beta");
  const HostTensor* mu = find_w(m, scope + "/moving_mean"); const HostTensor* var = find_w(m, scope + "/moving_variance");
  ODT_CHECK(g && b && mu && var, "missing BN variables for " + scope);
  ODT_CHECK((int)g->data.size() == c && (int)var->data.size() == c, "bad BN shape for " + scope);
  scale->resize(c); shift->resize(c);
  for (int o = 0; o < c; ++o) {
    const double inv = (double)g->data[o] / std::sqrt((double)var->data[o] + 1e-3);
    (*scale)[o] = inv; (*shift)[o] = (double)b->data[o] - (double)mu->data[o] * inv;
  }
  return 0;
}

// 1x1 conv weights [1,1,cin,cout] (+ BN scope or bias) -> device [cout][cin_pad] + bias[cout]
int eff_upload_pw(odt_model* m, const std::string& conv, const std::string& bn_scope, bool has_bias, int cin,
                  int cin_pad, int cout, const float** wt_out, const float** bias_out,
                  const char* kernel_name = "kernel") {
  const HostTensor* W = find_w(m, conv + "/" + kernel_name);
  ODT_CHECK(W != nullptr && W->data.size() == (size_t)cin * cout, "missing / bad " + conv + "/" + kernel_name);
  std::vector<double> scale(cout, 1.0), shift(cout, 0.0);
  if (!bn_scope.empty() && eff_bn(m, bn_scope, cout, &scale, &shift)) return 1;
  if (has_bias) {
    const HostTensor* b = find_w(m, conv + "/bias");
    ODT_CHECK(b != nullptr && (int)b->data.size() == cout, "missing / bad " + conv + "/bias");
    for (int o = 0; o < cout; ++o) shift[o] += (double)b->data[o] * scale[o];
  }
  std::vector<float> wt((size_t)cout * cin_pad, 0.f), bias(cout);
  for (int o = 0; o < cout; ++o) {
    for (int i = 0; i < cin; ++i) wt[(size_t)o * cin_pad + i] = (float)((double)W->data[(size_t)i * cout + o] * scale[o]);
    bias[o] = (float)shift[o];
  }
  if (upload_raw(m, wt, wt_out) || upload_raw(m, bias, bias_out)) return 1;
  return 0;
}


// ---- EfficientDet feature network (BiFPN) + class / box nets
// reference efficientdet_arch.py:105-200 (resample), :440-505 (P6/P7 + cells), :594-682 (nodes),
// :227-393 (nets); efficientdet_wrapper.py:511-587 (D0..D7 table).
struct EffDetCfg { int filters, cells, repeats; bool fastattn; };
EffDetCfg effdet_cfg(int d) {
  static const int T[8][3] = {{64, 3, 3}, {88, 4, 3}, {112, 5, 3}, {160, 6, 4}, {224, 7, 4}, {288, 7, 4}, {384, 8, 5}, {384, 8, 5}};
  return EffDetCfg{T[d][0], T[d][1], T[d][2], d != 7};
}

int eff_same(int n, int k, int s, int* before) {
  const int out = (n + s - 1) / s;
  const int tot = std::max((out - 1) * s + k - n, 0);
  *before = tot / 2;
  return out;
}

// depthwise 3x3 'same' (no BN, no activation) + pointwise 1x1 (+bias, optional BN fold, activation)
int eff_sepconv(odt_model* m, const std::string& scope, const std::string& bn_scope, const Tensor& in, int cin,
                int cout, int act, const std::string& tap, Tensor* out) {
  const int B = in.B, ldc = in.C;
  const HostTensor* Wd = find_w(m, scope + "/depthwise_kernel");
  ODT_CHECK(Wd && Wd->data.size() == (size_t)9 * cin, "missing / bad " + scope + "/depthwise_kernel");
  std::vector<float> v((size_t)9 * ldc, 0.f), bv(ldc, 0.f);
  for (int t = 0; t < 9; ++t) for (int c = 0; c < cin; ++c) v[(size_t)t * ldc + c] = Wd->data[(size_t)t * cin + c];
  const float *dwt, *dbias;
  if (upload_raw(m, v, &dwt) || upload_raw(m, bv, &dbias)) return 1;
  Tensor t1{};
  if (make_tensor(m, "", B, in.h, in.w, ldc, &t1, false)) return 1;     // (the depthwise kernel writes every channel of the stride)
  {
    Op op; op.kind = OP_DW;
    op.dw.in = in.d; op.dw.wt = dwt; op.dw.bias = dbias; op.dw.out = t1.d;
    op.dw.B = B; op.dw.H = in.h; op.dw.W = in.w; op.dw.Ho = in.h; op.dw.Wo = in.w; op.dw.ldc = ldc;
    op.dw.k = 3; op.dw.stride = 1; op.dw.pad_t = 1; op.dw.pad_l = 1; op.dw.act = 0;
    m->ops.push_back(op);
  }
  const float *wt, *bias;
  if (eff_upload_pw(m, scope, bn_scope, true, cin, ldc, cout, &wt, &bias, "pointwise_kernel")) return 1;
  // the pointwise bias variable of separable_conv2d is "<scope>/bias"
  if (add_conv(m, scope, t1, ldc, wt, bias, 1, 1, cout, 1, 1, 0, 0, in.h, in.w, 0, 0, nullptr, 0, false, r32(cout), out, tap)) return 1;
  m->convs.back().p.relu = act;
  return 0;
}

int build_effdet_heads(odt_model* m, const Tensor* red, const int* red_ch) {
  const odt_config& cfg = m->cfg;
  const EffDetCfg dc = effdet_cfg(cfg.eff_det);
  const int B = cfg.batch, F = dc.filters, LF = r32(F);
  const int ncls = cfg.num_class > 0 ? cfg.num_class : 90;
  // node sizes: utils.get_feat_sizes
  int fh[8], fw[8];
  fh[0] = cfg.height; fw[0] = cfg.width;
  for (int l = 1; l < 8; ++l) { fh[l] = (fh[l - 1] - 1) / 2 + 1; fw[l] = (fw[l - 1] - 1) / 2 + 1; }
  struct Feat { Tensor t; int ch; };
  std::vector<Feat> feats;
  for (int l = 3; l <= 5; ++l) {
    ODT_CHECK(red[l].h == fh[l] && red[l].w == fw[l], "backbone / feature-pyramid size mismatch");
    feats.push_back(Feat{red[l], red_ch[l]});
  }
  const float *wt, *bias;
  // optional 1x1 conv + BN when the channel count differs (resample_feature_map._maybe_apply_1x1)
  auto maybe_1x1 = [&](const std::string& scope, const Feat& f, Tensor* out) {
    if (f.ch == F) { *out = f.t; return 0; }
    if (eff_upload_pw(m, scope + "/conv2d", scope + "/bn", true, f.ch, f.t.C, F, &wt, &bias)) return 1;
    *out = Tensor{};
    return add_conv(m, scope + "/conv2d", f.t, f.t.C, wt, bias, 1, 1, F, 1, 1, 0, 0, f.t.h, f.t.w, 0, 0, nullptr, 0, false, LF, out, "");
  };
  auto fuse_input = [&](FuseParams& fp, int k, const Tensor& t, int th, int tw) {
    fp.in[k] = t.d; fp.ih[k] = t.h; fp.iw[k] = t.w; fp.sy[k] = fp.sx[k] = 1.f; fp.pt[k] = fp.pl[k] = 0;
    if (t.h > th && t.w > tw) {
      int pb; ODT_CHECK(eff_same(t.h, 3, 2, &pb) == th, "BiFPN: unsupported down-sampling ratio"); fp.pt[k] = pb;
      ODT_CHECK(eff_same(t.w, 3, 2, &pb) == tw, "BiFPN: unsupported down-sampling ratio"); fp.pl[k] = pb;
      fp.mode[k] = 2;
    } else if (t.h > th || t.w > tw) {
      ODT_CHECK(false, "BiFPN: incompatible feature map sizes (efficientdet_arch.py:196-199): every pyramid level "
                       "must shrink in both dimensions; use a larger input");
    } else if (t.h < th || t.w < tw) {
      fp.mode[k] = 1; fp.sy[k] = (float)t.h / (float)th; fp.sx[k] = (float)t.w / (float)tw;
    } else {
      fp.mode[k] = 0;
    }
    return 0;
  };
  // P6, P7 (efficientdet_arch.py:452-475)
  for (int l = 6; l <= 7; ++l) {
    Tensor src{};
    if (maybe_1x1("resample_p" + std::to_string(l), feats.back(), &src)) return 1;
    Tensor t{};
    if (make_tensor(m, "", B, fh[l], fw[l], LF, &t, true)) return 1;
    Op op; op.kind = OP_FUSE; std::memset(&op.fuse, 0, sizeof(op.fuse));
    if (fuse_input(op.fuse, 0, src, fh[l], fw[l])) return 1;
    op.fuse.n = 1; op.fuse.B = B; op.fuse.h = fh[l]; op.fuse.w = fw[l]; op.fuse.ldc = LF; op.fuse.out = t.d;
    m->ops.push_back(op);
    m->taps["fpn_in_" + std::to_string(l)] = t;
    feats.push_back(Feat{t, F});
  }
  static const int NODE_LVL[8] = {6, 5, 4, 3, 4, 5, 6, 7};
  static const int NODE_IN[8][3] = {{3, 4, -1}, {2, 5, -1}, {1, 6, -1}, {0, 7, -1}, {1, 7, 8}, {2, 6, 9}, {3, 5, 10}, {4, 11, -1}};
  for (int rep = 0; rep < dc.cells; ++rep) {
    for (int i = 0; i < 8; ++i) {
      const std::string p = "fpn_cells/cell_" + std::to_string(rep) + "/fnode" + std::to_string(i) + "/";
      const int lvl = NODE_LVL[i], th = fh[lvl], tw = fw[lvl];
      Op op; op.kind = OP_FUSE; std::memset(&op.fuse, 0, sizeof(op.fuse));
      int n = 0;
      double wsum = 0.0;
      for (int k = 0; k < 3 && NODE_IN[i][k] >= 0; ++k, ++n) {
        const int off = NODE_IN[i][k];
        Tensor src{};
        if (maybe_1x1(p + "resample_" + std::to_string(k) + "_" + std::to_string(off) + "_" + std::to_string(feats.size()),
                      feats[off], &src)) return 1;
        if (fuse_input(op.fuse, k, src, th, tw)) return 1;
        if (dc.fastattn) {
          const HostTensor* ws = find_w(m, p + (k == 0 ? std::string("WSM") : "WSM_" + std::to_string(k)));
          ODT_CHECK(ws != nullptr && ws->data.size() == 1, "missing " + p + "WSM");
          op.fuse.wgt[k] = std::max(ws->data[0], 0.f);
        }
      }
      if (dc.fastattn) {      // tf.add_n of float32 scalars, left to right, + 0.0001
        float tot = op.fuse.wgt[0];
        for (int k = 1; k < n; ++k) tot = tot + op.fuse.wgt[k];
        op.fuse.denom = tot + 0.0001f; op.fuse.weighted = 1;
      }
      (void)wsum;
      op.fuse.n = n; op.fuse.act = 2; op.fuse.B = B; op.fuse.h = th; op.fuse.w = tw; op.fuse.ldc = LF;
      const std::string q = p + "op_after_combine" + std::to_string(feats.size()) + "/";
      Tensor node{};
      Tensor fused{};
      if (make_tensor(m, "", B, th, tw, LF, &fused, false)) return 1;     // (the fusion kernel writes every channel of the stride)
      op.fuse.out = fused.d;
      m->ops.push_back(op);
      if (eff_sepconv(m, q + "conv", q + "bn", fused, F, F, 0, "cell" + std::to_string(rep) + "_fnode" + std::to_string(i), &node)) return 1;
      feats.push_back(Feat{node, F});
    }
    // next cell's inputs: the last node of every level (efficientdet_arch.py:676-682)
    std::vector<Feat> nxt(5);
    for (int l = 3; l <= 7; ++l)
      for (int i = 7; i >= 0; --i)
        if (NODE_LVL[i] == l) { nxt[l - 3] = feats[feats.size() - 8 + i]; break; }
    feats = nxt;
  }
  for (int l = 3; l <= 7; ++l) m->taps["fpn_" + std::to_string(l)] = feats[l - 3].t;
  // class / box nets: shared separable convs, per-level BN, swish (efficientdet_arch.py:227-393)
  Tensor cls_out[5], box_out[5];
  // Batch 1 and layers wide enough for the conv_split3 kernels: the five levels of a layer run as ONE depthwise launch and
  // ONE pointwise launch over the levels' pixels concatenated (each level padded to whole 256-row tiles so that a tile never
  // straddles two levels; the per-level BatchNorm becomes per-row-range scale / bias in the epilogue).  D7: 120 launches
  // instead of 600 for the two nets.  ODT_EFFDET_MERGE_LEVELS=0: one launch per level and layer (A/B; smaller models and
  // batch > 1 always).
  int off[6] = {0, 0, 0, 0, 0, 0};
  for (int l = 0; l < 5; ++l) off[l + 1] = off[l] + (fh[l + 3] * fw[l + 3] + 255) / 256 * 256;
  const int Mtot = off[5];
  bool merge = B == 1 && eff_split_on();
  merge = merge && !env_knob_off(K_EFFDET_MERGE_LEVELS);
  if (merge) {      // would the merged pointwise conv run on a conv_split3 kernel without split-K?
    ConvParams q; std::memset(&q, 0, sizeof(q));
    q.B = 1; q.H = Mtot / 256; q.W = 256; q.in_Ha = q.H; q.in_Wa = 256; q.Cin = LF; q.in_ldc = LF; q.Ho = q.H; q.Wo = 256; q.Cout = F;
    q.kh = q.kw = 1; q.stride = 1; q.dil = 1; q.out_H = q.H; q.out_W = 256; q.out_ldc = LF;
    const ConvPolicy pol = resolve_conv_policy(m);
    merge = conv_split_wanted(q, pol);
    if (merge) { conv_split_choose(q, pol); merge = q.wt_split_kind == 3 && q.splitk <= 1; }
  }
  if (merge) {
    auto level_view = [&](const Tensor& cat, int l) {
      Tensor v = cat;
      v.d = cat.d + (size_t)off[l] * cat.C; v.B = 1; v.H = v.h = fh[l + 3]; v.W = v.w = fw[l + 3];
      return v;
    };
    // one separable layer over all levels: depthwise (levels as maps of one launch) -> pointwise over Mtot rows
    auto merged_sepconv = [&](const std::string& scope, const std::string& bn_prefix, const Tensor* lvl_in, const Tensor* cat_in,
                              int cout, int act, Tensor* out) {
      const HostTensor* Wd = find_w(m, scope + "/depthwise_kernel");
      ODT_CHECK(Wd && Wd->data.size() == (size_t)9 * F, "missing / bad " + scope + "/depthwise_kernel");
      std::vector<float> v((size_t)9 * LF, 0.f), bv(LF, 0.f);
      for (int t = 0; t < 9; ++t) for (int c = 0; c < F; ++c) v[(size_t)t * LF + c] = Wd->data[(size_t)t * F + c];
      const float *dwt, *dbias;
      if (upload_raw(m, v, &dwt) || upload_raw(m, bv, &dbias)) return 1;
      Tensor t1{};
      if (make_tensor(m, "", 1, Mtot / 256, 256, LF, &t1, false)) return 1;      // (rows as [Mtot / 256, 256]: tap coordinates are 16-bit)
      {
        Op op; op.kind = OP_DW;
        op.dw.wt = dwt; op.dw.bias = dbias; op.dw.B = 1; op.dw.ldc = LF; op.dw.k = 3; op.dw.stride = 1; op.dw.pad_t = 1; op.dw.pad_l = 1;
        op.dw.act = 0; op.dw.nlvl = 5;
        for (int l = 0; l < 5; ++l) {
          op.dw.lin[l] = lvl_in != nullptr ? lvl_in[l].d : cat_in->d + (size_t)off[l] * LF;
          op.dw.lout[l] = t1.d + (size_t)off[l] * LF;
          op.dw.lH[l] = fh[l + 3]; op.dw.lW[l] = fw[l + 3];
        }
        op.dw.in = op.dw.lin[0]; op.dw.out = op.dw.lout[0];
        op.dw.H = op.dw.Ho = fh[3]; op.dw.W = op.dw.Wo = fw[3];
        m->ops.push_back(op);
      }
      const float *wt, *bias;
      if (eff_upload_pw(m, scope, "", true, F, LF, cout, &wt, &bias, "pointwise_kernel")) return 1;
      const int lco = r32(cout);
      if (add_conv(m, scope, t1, LF, wt, bias, 1, 1, cout, 1, 1, 0, 0, Mtot / 256, 256, 0, 0, nullptr, 0, false, lco, out, "")) return 1;
      ConvParams& cp = m->convs.back().p;
      cp.relu = act;
      if (!bn_prefix.empty()) {
        // per-level BatchNorm in the epilogue: out = (W x) * inv_l + (beta_l - mean_l * inv_l + b * inv_l)
        const HostTensor* pb = find_w(m, scope + "/bias");
        ODT_CHECK(pb != nullptr && (int)pb->data.size() == cout, "missing / bad " + scope + "/bias");
        std::vector<float> sc((size_t)5 * lco, 1.f), sh((size_t)5 * lco, 0.f);
        for (int l = 0; l < 5; ++l) {
          std::vector<double> scale, shift;
          if (eff_bn(m, bn_prefix + std::to_string(l + 3), cout, &scale, &shift)) return 1;
          for (int o = 0; o < cout; ++o) {
            sc[(size_t)l * lco + o] = (float)scale[o];
            sh[(size_t)l * lco + o] = (float)(shift[o] + (double)pb->data[o] * scale[o]);
          }
        }
        const float *dsc, *dsh;
        if (upload_raw(m, sc, &dsc) || upload_raw(m, sh, &dsh)) return 1;
        cp.bias = dsh; cp.lvl_scale = dsc; cp.nlvl = 5; cp.lvl_stride = lco;
        for (int l = 0; l < 5; ++l) cp.lvl_start[l] = off[l];      // (all five in use; fewer ranges: fill the rest with INT_MAX)
      }
      return 0;
    };
    Tensor lvl_feats[5];
    for (int l = 0; l < 5; ++l) lvl_feats[l] = feats[l].t;
    for (int net = 0; net < 2; ++net) {
      const std::string nn = net == 0 ? "class" : "box";
      Tensor x{};
      for (int r = 0; r < dc.repeats; ++r) {
        Tensor y{};
        if (merged_sepconv(nn + "_net/" + nn + "-" + std::to_string(r), nn + "_net/" + nn + "-" + std::to_string(r) + "-bn-",
                           r == 0 ? lvl_feats : nullptr, r == 0 ? nullptr : &x, F, 2, &y)) return 1;
        x = y;
      }
      Tensor o{};
      const int nout = net == 0 ? ncls * 9 : 36;
      if (merged_sepconv(nn + "_net/" + nn + "-predict", "", dc.repeats == 0 ? lvl_feats : nullptr, dc.repeats == 0 ? nullptr : &x, nout, 0, &o))
        return 1;
      for (int l = 0; l < 5; ++l) {
        (net == 0 ? cls_out : box_out)[l] = level_view(o, l);
        m->taps[nn + "_" + std::to_string(l + 3)] = (net == 0 ? cls_out : box_out)[l];
      }
    }
  } else {
  for (int l = 3; l <= 7; ++l) {
    for (int net = 0; net < 2; ++net) {
      const std::string nn = net == 0 ? "class" : "box";
      Tensor x = feats[l - 3].t;
      for (int r = 0; r < dc.repeats; ++r) {
        Tensor y{};
        if (eff_sepconv(m, nn + "_net/" + nn + "-" + std::to_string(r),
                        nn + "_net/" + nn + "-" + std::to_string(r) + "-bn-" + std::to_string(l), x, F, F, 2, "", &y)) return 1;
        x = y;
      }
      Tensor o{};
      const int nout = net == 0 ? ncls * 9 : 36;
      if (eff_sepconv(m, nn + "_net/" + nn + "-predict", "", x, F, nout, 0, nn + "_" + std::to_string(l), &o)) return 1;
      (net == 0 ? cls_out : box_out)[l - 3] = o;
    }
  }
  }
  // ---- detection tail (efficientdet_wrapper.py:363-480, anchors.py:369-489) + per-level ROIAlign mean
  EffPostParams& ep = m->eff_post;
  std::memset(&ep, 0, sizeof(ep));
  int tot = 0;
  for (int l = 0; l < 5; ++l) {
    ep.cls[l] = cls_out[l].d; ep.box[l] = box_out[l].d; ep.npix[l] = fh[l + 3] * fw[l + 3];
    ep.anchor_off[l] = tot; tot += ep.npix[l] * 9;
  }
  ep.anchor_off[5] = tot;
  ep.ldc_cls = cls_out[0].C; ep.ldc_box = box_out[0].C; ep.ncls = ncls; ep.B = B;
  const HostTensor* an = find_w(m, "effdet/anchors");
  ODT_CHECK(an != nullptr && an->data.size() == (size_t)tot * 4, "missing / bad effdet/anchors (expected [sum(h*w*9), 4])");
  if (upload_raw(m, an->data, &ep.anchors)) return 1;
  const long nlog = (long)tot * ncls;
  ep.k = (int)std::min<long>(cfg.eff_topk > 0 ? cfg.eff_topk : 5000, nlog);
  ep.max_out = cfg.result_per_im > 0 ? cfg.result_per_im : 100;
  ep.score_thresh = cfg.result_score_thresh; ep.iou_thresh = 0.5f;
  ep.image_scale = cfg.eff_image_scale > 0.f ? cfg.eff_image_scale : 1.f;
  ep.keys = (unsigned*)m->alloc_f((size_t)nlog, false);
  ep.hist = (unsigned*)m->alloc_f(256, true);
  ep.state = (unsigned*)m->alloc_f(8, true);
  ep.sel = (unsigned long long*)m->alloc_f((size_t)B * ep.k * 2, true);
  ep.cand_boxes = m->alloc_f((size_t)B * ep.k * 4, true); ep.cand_scores = m->alloc_f((size_t)B * ep.k, true);
  ep.cand_cls = (int*)m->alloc_f((size_t)B * ep.k, true); ep.cand_lvl = (int*)m->alloc_f((size_t)B * ep.k, true);
  ep.out_boxes = m->alloc_f((size_t)B * ep.max_out * 4, true); ep.out_scores = m->alloc_f((size_t)B * ep.max_out, true);
  ep.out_labels = (int*)m->alloc_f((size_t)B * ep.max_out, true); ep.out_levels = (int*)m->alloc_f((size_t)B * ep.max_out, true);
  ep.out_valid = (int*)m->alloc_f(B, true);
  ODT_CHECK(ep.keys && ep.hist && ep.state && ep.sel && ep.cand_boxes && ep.cand_scores && ep.cand_cls && ep.cand_lvl &&
            ep.out_boxes && ep.out_scores && ep.out_labels && ep.out_levels && ep.out_valid, "device allocation failed (effdet tail)");
  { Op op; op.kind = OP_EFF_POST; m->ops.push_back(op); }
  // fpn_box_feat [R, F]: crop_and_resize 14x14 + 2x2 average on the box's own level, mean over 7x7
  // (efficientdet_wrapper.py:244-296; the boxes are the SCALED output boxes, as in the reference)
  RoiAlignParams& rf = m->roi_eff;
  std::memset(&rf, 0, sizeof(rf));
  for (int l = 0; l < 5; ++l) {
    const Tensor& t = feats[l].t;
    rf.feat[l] = t.d; rf.h[l] = t.h; rf.w[l] = t.w; rf.ldc[l] = t.C; rf.alloc_h[l] = t.H; rf.alloc_w[l] = t.W;
    rf.inv_stride[l] = 1.0f / (float)(1 << (l + 3));
  }
  rf.C = F; rf.boxes = ep.out_boxes; rf.box_ind = nullptr; rf.per_image = ep.max_out; rf.count = ep.out_valid;
  rf.R_cap = B * ep.max_out; rf.levels = ep.out_levels; rf.level0 = 3;
  m->final_feat = m->alloc_f((size_t)B * ep.max_out * F * 49, true);
  m->final_pooled = m->alloc_f((size_t)B * ep.max_out * F, true);
  ODT_CHECK(m->final_feat && m->final_pooled, "device allocation failed (effdet features)");
  rf.out_nhwc = nullptr; rf.out_nchw = m->final_feat; rf.pooled = m->final_pooled; rf.pack_rows = 1;
  m->eff_filters = F;
  { Op op; op.kind = OP_ROI_EFF; m->ops.push_back(op); }
  return 0;
}

int build_plan_effnet(odt_model* m) {
  const odt_config& cfg = m->cfg;
  const int B = cfg.batch, H = cfg.height, W = cfg.width;
  ODT_CHECK(cfg.eff_backbone >= 0 && cfg.eff_backbone <= 7, "odt_create: eff_backbone must be 0..7");
  const std::string name = "efficientnet-b" + std::to_string(cfg.eff_backbone);
  int stemC = 0;
  const std::vector<EffBlock> blocks = eff_blocks(cfg.eff_backbone, &stemC);
  auto same = [](int n, int k, int s, int* out, int* before) {
    *out = (n + s - 1) / s;
    const int tot = std::max((*out - 1) * s + k - n, 0);
    *before = tot / 2;
    return tot;
  };
  // ---- input: normalised RGB, HWC4, physically padded with the stem's 'SAME' pads
  int Ho, Wo, pt, pl;
  const int ph = same(H, 3, 2, &Ho, &pt), pw = same(W, 3, 2, &Wo, &pl);
  m->Hp = H + ph; m->Wp = std::max(W + pw, 2 * (Wo - 1) + 8);
  if (make_tensor(m, "image_pad", B, m->Hp, m->Wp, 4, &m->image_pad, true)) return 1;
  m->frames_bytes = (size_t)B * H * W * 3 * sizeof(float);
  m->src_h = H; m->src_w = W;
  { m->bufs.emplace_back(new DevBuf()); if (m->bufs.back()->alloc(m->frames_bytes)) return 1;
    m->frames_dev.d = (float*)m->bufs.back()->p; }
  { Op op; op.kind = OP_PRE_RGB; op.pad_t = pt; op.pad_l = pl; m->ops.push_back(op); }
  // ---- stem: 3x3 s2 conv as a 3x1 conv over 8-pixel x 4-channel rows (K = 3 * 32), BN, swish
  const float *wt = nullptr, *bias = nullptr;
  {
    const HostTensor* W0 = find_w(m, name + "/stem/conv2d/kernel");
    ODT_CHECK(W0 && W0->data.size() == (size_t)3 * 3 * 3 * stemC, "missing / bad " + name + "/stem/conv2d/kernel");
    std::vector<double> scale, shift;
    if (eff_bn(m, name + "/stem/tpu_batch_normalization", stemC, &scale, &shift)) return 1;
    std::vector<float> v((size_t)stemC * 3 * 32, 0.f), bv(stemC);
    for (int o = 0; o < stemC; ++o) {
      for (int y = 0; y < 3; ++y) for (int x = 0; x < 3; ++x) for (int c = 0; c < 3; ++c)
        v[((size_t)o * 3 + y) * 32 + x * 4 + c] = (float)((double)W0->data[(((size_t)y * 3 + x) * 3 + c) * stemC + o] * scale[o]);
      bv[o] = (float)shift[o];
    }
    if (upload_raw(m, v, &wt) || upload_raw(m, bv, &bias)) return 1;
  }
  Tensor x{};
  if (add_conv(m, "stem", m->image_pad, 32, wt, bias, 3, 1, stemC, 2, 1, 0, 0, Ho, Wo, 0, 0, nullptr, 0, false,
               r32(stemC), &x, "stem")) return 1;
  m->convs.back().p.relu = 2;
  // ---- MBConv blocks
  Tensor red_feats[6]; int red_ch[6] = {0, 0, 0, 0, 0, 0};
  for (const EffBlock& b : blocks) {
    const std::string p = name + "/blocks_" + std::to_string(b.idx) + "/";
    const int mid = b.cin * b.expand, lmid = r32(mid);
    int nconv = 0, nbn = 0;
    auto cname = [&]() { const std::string n = nconv == 0 ? "conv2d" : "conv2d_" + std::to_string(nconv); ++nconv; return p + n; };
    auto bname = [&]() { const std::string n = nbn == 0 ? "tpu_batch_normalization" : "tpu_batch_normalization_" + std::to_string(nbn); ++nbn; return p + n; };
    const Tensor inp = x;
    Tensor t1 = x;
    int ho, wo, dpt, dpl;
    same(x.h, b.kernel, b.stride, &ho, &dpt); same(x.w, b.kernel, b.stride, &wo, &dpl);
    const int fmode = eff_fuse_mb_mode();
    const bool fuse_mb = b.expand != 1 && eff_split_on() && x.C % 32 == 0 && lmid % 64 == 0 && x.h == x.H && x.w == x.W &&
                         (fmode >= 2 || (fmode == 1 && std::min(ho, wo) >= eff_fuse_mb_min()));
    const float *ewt = nullptr, *ebias = nullptr;
    if (b.expand != 1) {
      const std::string cn = cname(), bn = bname();
      if (eff_upload_pw(m, cn, bn, false, b.cin, x.C, mid, &ewt, &ebias)) return 1;
      if (!fuse_mb) {
        t1 = Tensor{};
        if (add_conv(m, cn, x, x.C, ewt, ebias, 1, 1, mid, 1, 1, 0, 0, x.h, x.w, 0, 0, nullptr, 0, false, lmid, &t1, "")) return 1;
        m->convs.back().p.relu = 2;
      }
    }
    // depthwise + BN + swish
    Tensor t2{};
    if (make_tensor(m, "", B, ho, wo, lmid, &t2, true)) return 1;
    float* se_part = nullptr; int se_nsplit = 0;
    {
      const HostTensor* Wd = find_w(m, p + "depthwise_conv2d/depthwise_kernel");
      ODT_CHECK(Wd && Wd->data.size() == (size_t)b.kernel * b.kernel * mid, "missing / bad " + p + "depthwise_conv2d/depthwise_kernel");
      std::vector<double> scale, shift;
      if (eff_bn(m, bname(), mid, &scale, &shift)) return 1;
      std::vector<float> v((size_t)b.kernel * b.kernel * lmid, 0.f), bv(lmid, 0.f);
      for (int t = 0; t < b.kernel * b.kernel; ++t)
        for (int c = 0; c < mid; ++c) v[(size_t)t * lmid + c] = (float)((double)Wd->data[(size_t)t * mid + c] * scale[c]);
      for (int c = 0; c < mid; ++c) bv[c] = (float)shift[c];
      const float *dwt, *dbias;
      if (upload_raw(m, v, &dwt) || upload_raw(m, bv, &dbias)) return 1;
      if (fuse_mb) {
        // expand 1x1 + BN + swish -> depthwise + BN + swish in one kernel: the [h, w, mid] tensor is never written
        Op op; op.kind = OP_MB_EXPAND_DW;
        MbExpandDwParams& q = op.mb;
        q.x = x.d; q.B = B; q.H = x.h; q.W = x.w; q.in_ldc = x.C;
        q.e_bias = ebias; q.mid = mid; q.lmid = lmid; q.dw_wt = dwt; q.dw_bias = dbias; q.out = t2.d;
        q.Ho = ho; q.Wo = wo; q.k = b.kernel; q.stride = b.stride; q.pad_t = dpt; q.pad_l = dpl;
        {
          // the expand weights as the bf16x3 piece image of the one-stage 256 x 64 kernel (built once, here)
          float* img = m->alloc_f((mbconv_expand_weight_bytes(lmid, x.C) + 3) / 4, false);
          ODT_CHECK(img != nullptr, "device allocation failed (expand weight image of " + p + ")");
          ConvParams cp{};
          cp.wt = ewt; cp.Cout = mid; cp.Cin = x.C; cp.kh = 1; cp.kw = 1; cp.wt_split_kind = 1; cp.wt_split_bn = 64;
          if (conv_make_split_weights(cp, img, nullptr)) return 1;
          ODT_HIP(hipDeviceSynchronize());
          q.w_img = img;
        }
        q.nsplit = 0;
        se_nsplit = mbconv_expand_dw_splits(q);
        q.nsplit = se_nsplit;
        se_part = m->alloc_f((size_t)B * se_nsplit * lmid, false);
        ODT_CHECK(se_part, "device allocation failed (SE)");
        q.sum_part = se_part;
        m->ops.push_back(op);
        ++m->mb_fused;
      } else {
        Op op; op.kind = OP_DW;
        op.dw.in = t1.d; op.dw.wt = dwt; op.dw.bias = dbias; op.dw.out = t2.d;
        op.dw.B = B; op.dw.H = t1.h; op.dw.W = t1.w; op.dw.Ho = ho; op.dw.Wo = wo; op.dw.ldc = lmid;
        op.dw.k = b.kernel; op.dw.stride = b.stride; op.dw.pad_t = dpt; op.dw.pad_l = dpl; op.dw.act = 2;
        // fused squeeze: the depthwise kernel also delivers per-workgroup sums of its output
        // (sized by the split count the launcher will use: a placeholder makes dwconv_splits() take the fused-squeeze path)
        op.dw.sum_part = reinterpret_cast<float*>(sizeof(float));
        se_nsplit = dwconv_splits(op.dw);
        ODT_CHECK(se_nsplit >= 1 && se_nsplit <= 1024, "dwconv_splits: bad number of partial sums");
        se_part = m->alloc_f((size_t)B * se_nsplit * lmid, false);
        ODT_CHECK(se_part, "device allocation failed (SE)");
        op.dw.sum_part = se_part;
        m->ops.push_back(op);
      }
    }
    // squeeze-excite gate from that mean (1x1 reduce + swish -> 1x1 expand + sigmoid); the channel scale itself is folded
    // into the projection's weights at batch 1 (W diag(g) instead of a read-modify-write pass over the widest tensor of
    // the block), and is a pass over the activations otherwise
    float* gate = nullptr;
    {
      gate = m->alloc_f((size_t)B * lmid, true);
      ODT_CHECK(gate, "device allocation failed (SE)");
      const HostTensor* W1 = find_w(m, p + "se/conv2d/kernel"); const HostTensor* B1 = find_w(m, p + "se/conv2d/bias");
      const HostTensor* W2 = find_w(m, p + "se/conv2d_1/kernel"); const HostTensor* B2 = find_w(m, p + "se/conv2d_1/bias");
      ODT_CHECK(W1 && B1 && W2 && B2, "missing squeeze-excite variables of " + p);
      ODT_CHECK(W1->data.size() == (size_t)mid * b.se && W2->data.size() == (size_t)b.se * mid &&
                (int)B1->data.size() == b.se && (int)B2->data.size() == mid, "bad squeeze-excite shapes in " + p);
      std::vector<float> w1((size_t)b.se * lmid, 0.f), w2t((size_t)b.se * lmid, 0.f);
      for (int j = 0; j < b.se; ++j)
        for (int c = 0; c < mid; ++c) {
          w1[(size_t)j * lmid + c] = W1->data[(size_t)c * b.se + j];      // kernel [1,1,mid,se]
          w2t[(size_t)j * lmid + c] = W2->data[(size_t)j * mid + c];      // kernel [1,1,se,mid]
        }
      Op op; op.kind = OP_SE_GATE_MEAN; op.in = t2;
      op.se.HW = ho * wo; op.se.ldc = lmid; op.se.mid = mid; op.se.se = b.se; op.se.gate = gate;
      op.se.r = m->alloc_f((size_t)B * 256, true);
      op.se.mean = m->alloc_f((size_t)B * lmid, true);
      op.se.part = se_part; op.se.nsplit = se_nsplit;
      ODT_CHECK(op.se.r != nullptr && op.se.mean != nullptr, "device allocation failed (SE)");
      if (upload_raw(m, w1, &op.se.w1) || upload_raw(m, B1->data, &op.se.b1) || upload_raw(m, w2t, &op.se.w2t) ||
          upload_raw(m, B2->data, &op.se.b2)) return 1;
      m->ops.push_back(op);
      if (B != 1 || !eff_wscale_on()) { Op sc; sc.kind = OP_CSCALE; sc.in = t2; sc.aux = gate; m->ops.push_back(sc); }
    }
    // projection + BN (+ identity skip)
    {
      const std::string cn = cname(), bn = bname();
      if (eff_upload_pw(m, cn, bn, false, mid, lmid, b.cout, &wt, &bias)) return 1;
      const bool skip = b.stride == 1 && b.cin == b.cout;
      Tensor y{};
      std::string tap = "block_" + std::to_string(b.idx);
      const float* wt_run = wt;
      if (B == 1 && eff_wscale_on()) {       // the conv runs on gate-scaled weights, rebuilt per forward right behind the gate
        float* ws = m->alloc_f((size_t)b.cout * lmid, true);
        ODT_CHECK(ws, "device allocation failed (SE)");
        Op wsop; wsop.kind = OP_WSCALE; wsop.conv = (int)m->convs.size(); wsop.wt0 = wt; wsop.aux = gate;
        m->ops.push_back(wsop);
        wt_run = ws;
      }
      if (add_conv(m, cn, t2, lmid, wt_run, bias, 1, 1, b.cout, 1, 1, 0, 0, ho, wo, 0, 0, skip ? &inp : nullptr, 1, false,
                   r32(b.cout), &y, tap)) return 1;
      if (b.reduction) { m->taps["reduction_" + std::to_string(b.reduction)] = y; red_feats[b.reduction] = y; red_ch[b.reduction] = b.cout; }
      x = y;
    }
  }
  if (cfg.eff_det >= 0 && build_effdet_heads(m, red_feats, red_ch)) return 1;
  // the EfficientNet / BiFPN 1x1 convs: bf16x3 split kernels where their tiles (with split-K) fill the chip, as for the
  // FPN detector (conv_arith / ODT_CONV_* apply; ODT_EFFDET_SPLIT=0: exact-f32 MFMA everywhere and 32-channel strides)
  if (eff_split_on() && attach_split_weights(m)) return 1;
  if (plan_arena(m)) return 1;
  if (upload_conv_records(m)) return 1;
  return 0;
}

}  // namespace odt
