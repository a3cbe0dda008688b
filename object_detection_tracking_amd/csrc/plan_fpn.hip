// Static plan of the reference's Mask_RCNN_FPN / Mask_RCNN_FPN_multi inference graphs:
// preprocess -> ResNet-101-dilated + FPN -> RPN -> proposals -> ROIAlign -> box head -> detections -> appearance features
// (+ mask head).  reference nn.py:843-1014, models.py:979-1108, :2058-2408.
#include "odt_model.hpp"

#define g_err (::odt::last_error())

namespace odt {

// Build the whole static plan (called from odt_finalize_weights).
int build_plan(odt_model* m) {
  const odt_config& cfg = m->cfg;
  m->arena_on = cfg.keep_taps == 0;
  if (cfg.graph == ODT_GRAPH_EFFNET) return build_plan_effnet(m);
  const int B = cfg.batch, H = cfg.height, W = cfg.width;
  const int FC = cfg.fpn_channels;
  // ---- front end geometry (nn.py:860-896; tf_pad_reverse => pad [3, 2 + pad_to_32])
  const int ph = ceil_div(H, 32) * 32 - H, pw = ceil_div(W, 32) * 32 - W;
  const int Hp = 3 + H + 2 + ph, Wpl = 3 + W + 2 + pw;
  const int Ho0 = (Hp - 7) / 2 + 1, Wo0 = (Wpl - 7) / 2 + 1;
  const int Wp = 2 * Wo0 + 8;          // room for the 8th (zero-weight) tap of the last window
  m->Hp = Hp; m->Wp = Wp;
  if (make_tensor(m, "image_pad", B, Hp, Wp, 4, &m->image_pad)) return 1;
  m->frames_bytes = (size_t)B * H * W * 3 * sizeof(float);
  m->src_h = H; m->src_w = W;
  { m->bufs.emplace_back(new DevBuf()); if (m->bufs.back()->alloc(m->frames_bytes)) return 1;
    m->frames_dev.d = (float*)m->bufs.back()->p; }
  { Op op; op.kind = OP_PRE; m->ops.push_back(op); }

  // ---- conv0: 7x7 s2 VALID as a 7x1 conv over 8-tap x 4-channel rows (K = 7*32)
  const float *wt = nullptr, *bias = nullptr;
  {
    const HostTensor* W0 = find_w(m, "conv0/W");
    ODT_CHECK(W0 && W0->data.size() == (size_t)7 * 7 * 3 * 64, "missing/bad conv0/W");
    HostTensor v;   // virtual HWIO [7,1,32,64]
    v.data.assign((size_t)7 * 32 * 64, 0.f);
    for (int y = 0; y < 7; ++y)
      for (int x = 0; x < 7; ++x)
        for (int c = 0; c < 3; ++c)
          for (int o = 0; o < 64; ++o)
            v.data[((size_t)y * 32 + x * 4 + c) * 64 + o] = W0->data[(((size_t)y * 7 + x) * 3 + c) * 64 + o];
    m->host_w["__conv0v/W"] = v;
    for (const char* s : {"gamma", "beta", "mean/EMA", "variance/EMA"}) {
      const HostTensor* t = find_w(m, std::string("conv0/bn/") + s);
      ODT_CHECK(t != nullptr, std::string("missing conv0/bn/") + s);
      m->host_w[std::string("__conv0v/bn/") + s] = *t;
    }
    if (upload_conv(m, "__conv0v", 7, 1, 32, 64, true, &wt, &bias)) return 1;
  }
  Tensor x{};
  if (add_conv(m, "conv0", m->image_pad, 32, wt, bias, 7, 1, 64, 2, 1, 0, 0, Ho0, Wo0, 0, 0, nullptr, 0,
               true, 64, &x, "conv0")) return 1;
  // ---- pool0: pad top/left 1, 3x3 s2 VALID max
  Tensor pool{};
  const int Hq = (Ho0 + 1 - 3) / 2 + 1, Wq = (Wo0 + 1 - 3) / 2 + 1;
  if (make_tensor(m, "pool0", B, Hq, Wq, 64, &pool)) return 1;
  { Op op; op.kind = OP_POOL; op.in = x; op.out = pool; m->ops.push_back(op); }
  x = pool;

  // ---- ResNet groups (nn.py:459-588, 898-936)
  Tensor cfeat[4];
  const int feats[4] = {64, 128, 256, 512};
  int cin = 64;
  for (int g = 0; g < 4; ++g) {
    const int cnt = cfg.num_blocks[g], ch = feats[g];
    for (int i = 0; i < cnt; ++i) {
      const std::string pre = "group" + std::to_string(g) + "/block" + std::to_string(i);
      const int stride = (i == 0 && g > 0) ? 2 : 1;
      const int dil = (g == 3 && cfg.use_dilations && i >= cnt - 3) ? 2 : 1;
      Tensor t1{}, t2{}, sc = x, y{};
      if (upload_conv(m, pre + "/conv1", 1, 1, cin, ch, true, &wt, &bias)) return 1;
      if (add_conv(m, pre + "/conv1", x, cin, wt, bias, 1, 1, ch, 1, 1, 0, 0, x.h, x.w, 0, 0, nullptr, 0,
                   true, ch, &t1, "")) return 1;
      if (upload_conv(m, pre + "/conv2", 3, 3, ch, ch, true, &wt, &bias)) return 1;
      int Ho, Wo;
      if (stride == 2) {
        const int keff = 2 * dil + 1;
        const int h2 = (x.h + 1 - keff) / 2 + 1, w2 = (x.w + 1 - keff) / 2 + 1;
        const int off = dil != 1 ? 1 : 0;       // nn.py:493-497 second pad, after BN+ReLU
        if (add_conv(m, pre + "/conv2", t1, ch, wt, bias, 3, 3, ch, 2, dil, 1, 1, h2, w2, off, off,
                     nullptr, 0, true, ch, &t2, "")) return 1;
        Ho = h2 + off; Wo = w2 + off;
      } else {
        if (add_conv(m, pre + "/conv2", t1, ch, wt, bias, 3, 3, ch, 1, dil, dil, dil, x.h, x.w, 0, 0,
                     nullptr, 0, true, ch, &t2, "")) return 1;
        Ho = x.h; Wo = x.w;
      }
      const std::string tap = (i == cnt - 1) ? "c" + std::to_string(g + 2) : (i == 0 ? pre : "");
      const bool fuse_shortcut = !env_knob_off(K_FUSE_SHORTCUT);
      if (cin != ch * 4 && fuse_shortcut) {
        // stage entry: conv3(t2) + convshortcut(x[::stride]) as one K-concatenated GEMM -- saves the
        // shortcut tensor's write + read and one launch (shortcut[:, :, :-1, :-1] of nn.py:555-556
        // never matters: the stride-2 samples stop at 2 * (Ho - 1) <= h - 2)
        if (stride == 2) {
          const int hs = (x.h - 2) / 2 + 1, ws = (x.w - 2) / 2 + 1;
          ODT_CHECK(hs == Ho && ws == Wo, "shortcut / conv2 geometry mismatch in " + pre);
        }
        if (upload_conv_cat(m, pre + "/conv3", ch, pre + "/convshortcut", cin, ch * 4, &wt, &bias)) return 1;
        if (add_conv(m, pre + "/conv3+shortcut", t2, ch, wt, bias, 1, 1, ch * 4, 1, 1, 0, 0, Ho, Wo, 0, 0, nullptr, 0,
                     true, ch * 4, &y, tap)) return 1;
        ConvParams& cp = m->convs.back().p;
        cp.in2 = x.d; cp.Cin2 = cin; cp.in2_ldc = x.C; cp.in2_Ha = x.H; cp.in2_Wa = x.W; cp.in2_stride = stride;
        x = y;
        cin = ch * 4;
        continue;
      }
      if (cin != ch * 4) {
        if (upload_conv(m, pre + "/convshortcut", 1, 1, cin, ch * 4, true, &wt, &bias)) return 1;
        Tensor s{};
        if (stride == 2) {
          Tensor xc = x;              // shortcut[:, :, :-1, :-1] (nn.py:555-556)
          xc.h = x.h - 1; xc.w = x.w - 1;
          const int hs = (xc.h - 1) / 2 + 1, ws = (xc.w - 1) / 2 + 1;
          ODT_CHECK(hs == Ho && ws == Wo, "shortcut / conv2 geometry mismatch in " + pre);
          if (add_conv(m, pre + "/convshortcut", xc, cin, wt, bias, 1, 1, ch * 4, 2, 1, 0, 0, hs, ws, 0, 0,
                       nullptr, 0, false, ch * 4, &s, "")) return 1;
        } else {
          if (add_conv(m, pre + "/convshortcut", x, cin, wt, bias, 1, 1, ch * 4, 1, 1, 0, 0, x.h, x.w, 0,
                       0, nullptr, 0, false, ch * 4, &s, "")) return 1;
        }
        sc = s;
      }
      if (upload_conv(m, pre + "/conv3", 1, 1, ch, ch * 4, true, &wt, &bias)) return 1;
      if (add_conv(m, pre + "/conv3", t2, ch, wt, bias, 1, 1, ch * 4, 1, 1, 0, 0, Ho, Wo, 0, 0, &sc, 1,
                   true, ch * 4, &y, tap)) return 1;
      x = y;
      cin = ch * 4;
    }
    cfeat[g] = x;
  }

  // ---- FPN (nn.py:947-1014): lateral 1x1 (+ nearest-2x top-down add), posthoc 3x3, P6
  Tensor lat[4], P[5];
  for (int l = 3; l >= 0; --l) {
    const std::string sc = "fpn/lateral_1x1_c" + std::to_string(l + 2);
    if (upload_conv(m, sc, 1, 1, cfeat[l].c, FC, false, &wt, &bias)) return 1;
    const Tensor* res = l < 3 ? &lat[l + 1] : nullptr;
    if (res) ODT_CHECK(res->H * 2 == cfeat[l].h && res->W * 2 == cfeat[l].w, "FPN levels are not exact 2x");
    lat[l] = Tensor{};
    if (add_conv(m, sc, cfeat[l], cfeat[l].c, wt, bias, 1, 1, FC, 1, 1, 0, 0, cfeat[l].h, cfeat[l].w, 0, 0,
                 res, 2, false, FC, &lat[l], "")) return 1;
  }
  for (int l = 0; l < 4; ++l) {
    const std::string sc = "fpn/posthoc_3x3_p" + std::to_string(l + 2);
    if (upload_conv(m, sc, 3, 3, FC, FC, false, &wt, &bias)) return 1;
    P[l] = Tensor{};
    if (add_conv(m, sc, lat[l], FC, wt, bias, 3, 3, FC, 1, 1, 1, 1, lat[l].h, lat[l].w, 0, 0, nullptr, 0,
                 false, FC, &P[l], "")) return 1;
  }
  {
    const int h6 = (P[3].h - 1) / 2 + 1, w6 = (P[3].w - 1) / 2 + 1;
    if (make_tensor(m, "p6", B, h6, w6, FC, &P[4])) return 1;
    Op op; op.kind = OP_SUB2; op.in = P[3]; op.out = P[4]; m->ops.push_back(op);
  }
  // slice_feature_and_anchors (models.py:372-400): P2..P4 cropped to ceil(H / stride)
  const int strides[5] = {4, 8, 16, 32, 64};
  for (int l = 0; l < 3; ++l) {
    const int th = (int)std::ceil((float)H * (float)(1.0 / strides[l]));
    const int tw = (int)std::ceil((float)W * (float)(1.0 / strides[l]));
    ODT_CHECK(th <= P[l].H && tw <= P[l].W, "sliced feature larger than the feature map");
    P[l].h = th; P[l].w = tw;
  }
  for (int l = 0; l < 5; ++l) m->taps["p" + std::to_string(l + 2)] = P[l];

  // ---- RPN head (models.py:979-1009), class(3) + box(12) merged into one 15-channel 1x1
  const float *w_r0, *b_r0, *w_r1, *b_r1;
  if (upload_conv(m, "rpn/conv0", 3, 3, FC, FC, false, &w_r0, &b_r0)) return 1;
  {
    const HostTensor* wc = find_w(m, "rpn/class/W"); const HostTensor* bc = find_w(m, "rpn/class/b");
    const HostTensor* wb = find_w(m, "rpn/box/W"); const HostTensor* bb = find_w(m, "rpn/box/b");
    ODT_CHECK(wc && bc && wb && bb, "missing rpn/class or rpn/box variables");
    ODT_CHECK(wc->data.size() == (size_t)FC * 3 && wb->data.size() == (size_t)FC * 12, "bad rpn head shapes");
    HostTensor v, vb;
    v.data.resize((size_t)FC * 15); vb.data.resize(15);
    for (int i = 0; i < FC; ++i) {
      for (int a = 0; a < 3; ++a) v.data[(size_t)i * 15 + a] = wc->data[(size_t)i * 3 + a];
      for (int j = 0; j < 12; ++j) v.data[(size_t)i * 15 + 3 + j] = wb->data[(size_t)i * 12 + j];
    }
    for (int a = 0; a < 3; ++a) vb.data[a] = bc->data[a];
    for (int j = 0; j < 12; ++j) vb.data[3 + j] = bb->data[j];
    m->host_w["__rpnhead/W"] = v; m->host_w["__rpnhead/b"] = vb;
    if (upload_conv(m, "__rpnhead", 1, 1, FC, 15, false, &w_r1, &b_r1)) return 1;
  }
  Tensor rpn_out[5];
  for (int l = 0; l < 5; ++l) {
    Tensor t{};
    if (add_conv(m, "rpn/conv0@p" + std::to_string(l + 2), P[l], FC, w_r0, b_r0, 3, 3, FC, 1, 1, 1, 1,
                 P[l].h, P[l].w, 0, 0, nullptr, 0, true, FC, &t, "")) return 1;
    rpn_out[l] = Tensor{};
    if (add_conv(m, "rpn/head@p" + std::to_string(l + 2), t, FC, w_r1, b_r1, 1, 1, 15, 1, 1, 0, 0, P[l].h,
                 P[l].w, 0, 0, nullptr, 0, false, kRpnCh, &rpn_out[l], "rpn" + std::to_string(l + 2)))
      return 1;
  }

  // ---- proposals
  const int K = cfg.rpn_topk, L = 5;
  ProposalParams& pp = m->prop;
  pp.nlevels = L; pp.graph = cfg.graph; pp.B = B; pp.K = K; pp.img_h = H; pp.img_w = W;
  pp.nms_thresh = cfg.rpn_nms_thresh; pp.decode_clip = cfg.rpn_decode_clip;
  for (int l = 0; l < L; ++l) {
    const HostTensor* a = find_w(m, "anchors/lvl" + std::to_string(l));
    ODT_CHECK(a != nullptr && a->shape.size() == 4 && a->shape[2] == 3 && a->shape[3] == 4 &&
              a->shape[0] == a->shape[1], "missing/bad anchors/lvl" + std::to_string(l));
    ODT_CHECK(a->shape[0] >= rpn_out[l].h && a->shape[0] >= rpn_out[l].w, "anchor field smaller than feature map");
    const float* d;
    if (upload_raw(m, a->data, &d)) return 1;
    pp.lvl[l].rpn = rpn_out[l].d; pp.lvl[l].anchors = d; pp.lvl[l].h = rpn_out[l].h; pp.lvl[l].w = rpn_out[l].w;
    pp.lvl[l].field = (int)a->shape[0];
    ODT_CHECK(rpn_out[l].H == rpn_out[l].h && rpn_out[l].W == rpn_out[l].w, "rpn output must be dense");
  }
  const size_t per = (size_t)B * L * K;
  pp.cand_boxes = m->alloc_f(per * 4, true); pp.cand_scores = m->alloc_f(per, true);
  pp.lvl_boxes = m->alloc_f(per * 4, true); pp.lvl_scores = m->alloc_f(per, true);
  pp.cand_count = (int*)m->alloc_f((size_t)B * L, true); pp.lvl_count = (int*)m->alloc_f((size_t)B * L, true);
  Tensor props{}; if (make_tensor(m, "proposals", 1, B, K, 4, &props, true)) return 1;
  pp.props = props.d;
  pp.nprops = (int*)m->alloc_f(B, true);
  pp.chunk_keys = (unsigned long long*)m->alloc_f((size_t)B * proposal_total_chunks(pp) * K * 2, true);
  ODT_CHECK(pp.cand_boxes && pp.cand_scores && pp.lvl_boxes && pp.lvl_scores && pp.cand_count &&
            pp.lvl_count && pp.nprops && pp.chunk_keys, "device allocation failed (proposals)");
  { Op op; op.kind = OP_PROPOSALS; m->ops.push_back(op); }

  // ---- ROIAlign over P2..P5 -> box head (models.py:465-485, 1030-1108)
  Tensor roi{}; if (make_tensor(m, "roi_feat", 1, 1, B * K, 49 * FC, &roi, true)) return 1;
  RoiAlignParams& rh = m->roi_head;
  std::memset(&rh, 0, sizeof(rh));
  for (int l = 0; l < 4; ++l) {
    rh.feat[l] = P[l].d; rh.h[l] = P[l].h; rh.w[l] = P[l].w; rh.ldc[l] = P[l].C;
    rh.alloc_h[l] = P[l].H; rh.alloc_w[l] = P[l].W; rh.inv_stride[l] = (float)(1.0 / strides[l]);
  }
  rh.C = FC; rh.boxes = props.d; rh.box_ind = nullptr; rh.per_image = K; rh.count = pp.nprops;
  rh.R_cap = B * K; rh.out_nhwc = roi.d;
  m->roi_final = rh;
  { Op op; op.kind = OP_ROI_HEAD; m->ops.push_back(op); }

  const int D = cfg.head_dim, C = cfg.num_class;
  {   // fc6: rows of W are flattened NCHW (c*49 + h*7 + w, nn.py:736-738); our RoI rows are (h,w,c)
    const HostTensor* w6 = find_w(m, "fastrcnn/fc6/W");
    ODT_CHECK(w6 && w6->data.size() == (size_t)FC * 49 * D, "missing/bad fastrcnn/fc6/W");
    HostTensor v; v.data.resize(w6->data.size());
    for (int c = 0; c < FC; ++c)
      for (int s = 0; s < 49; ++s)
        std::memcpy(&v.data[((size_t)s * FC + c) * D], &w6->data[((size_t)c * 49 + s) * D], sizeof(float) * D);
    m->host_w["__fc6/W"] = v;
    const HostTensor* b6 = find_w(m, "fastrcnn/fc6/b"); ODT_CHECK(b6 != nullptr, "missing fastrcnn/fc6/b");
    m->host_w["__fc6/b"] = *b6;
  }
  Tensor h6{}, h7{}, hout{};
  if (upload_conv(m, "__fc6", 1, 1, 49 * FC, D, false, &wt, &bias)) return 1;
  // compacted rows: only the first nprops[b] rows of each image are meaningful
  if (add_conv(m, "fastrcnn/fc6", roi, 49 * FC, wt, bias, 1, 1, D, 1, 1, 0, 0, 1, B * K, 0, 0, nullptr, 0, true,
               D, &h6, "fc6")) return 1;
  if (upload_conv(m, "fastrcnn/fc7", 1, 1, D, D, false, &wt, &bias)) return 1;
  if (add_conv(m, "fastrcnn/fc7", h6, D, wt, bias, 1, 1, D, 1, 1, 0, 0, 1, B * K, 0, 0, nullptr, 0, true, D,
               &h7, "fc7")) return 1;
  {
    const HostTensor* wc = find_w(m, "fastrcnn/outputs/class/W"); const HostTensor* bc = find_w(m, "fastrcnn/outputs/class/b");
    const HostTensor* wb = find_w(m, "fastrcnn/outputs/box/W"); const HostTensor* bb = find_w(m, "fastrcnn/outputs/box/b");
    ODT_CHECK(wc && bc && wb && bb, "missing fastrcnn/outputs variables");
    ODT_CHECK(wc->data.size() == (size_t)D * C && wb->data.size() == (size_t)D * C * 4, "bad fastrcnn/outputs shapes");
    HostTensor v, vb; v.data.resize((size_t)D * C * 5); vb.data.resize((size_t)C * 5);
    for (int i = 0; i < D; ++i) {
      for (int c = 0; c < C; ++c) v.data[(size_t)i * C * 5 + c] = wc->data[(size_t)i * C + c];
      for (int j = 0; j < 4 * C; ++j) v.data[(size_t)i * C * 5 + C + j] = wb->data[(size_t)i * 4 * C + j];
    }
    for (int c = 0; c < C; ++c) vb.data[c] = bc->data[c];
    for (int j = 0; j < 4 * C; ++j) vb.data[C + j] = bb->data[j];
    m->host_w["__headout/W"] = v; m->host_w["__headout/b"] = vb;
    if (upload_conv(m, "__headout", 1, 1, D, C * 5, false, &wt, &bias)) return 1;
  }
  const int ld = (C * 5 + 3) / 4 * 4;
  if (add_conv(m, "fastrcnn/outputs", h7, D, wt, bias, 1, 1, C * 5, 1, 1, 0, 0, 1, B * K, 0, 0, nullptr, 0,
               false, ld, &hout, "head_out")) return 1;

  // ---- detection tail
  DetectParams& dp = m->det;
  std::memset(&dp, 0, sizeof(dp));
  dp.graph = cfg.graph; dp.B = B; dp.K = K; dp.C = C; dp.head_out = hout.d; dp.ld = hout.C;
  dp.props = props.d; dp.nprops = pp.nprops; dp.img_h = H; dp.img_w = W;
  for (int i = 0; i < 4; ++i) dp.reg_w[i] = cfg.bbox_reg_weights[i];
  dp.decode_clip = cfg.head_decode_clip; dp.score_thresh = cfg.result_score_thresh;
  dp.nms_thresh = cfg.head_nms_thresh; dp.per_im = cfg.result_per_im;
  const int per_im = cfg.result_per_im;
  Tensor dec{}, prb{};
  if (make_tensor(m, "decoded_boxes", 1, B * K, C - 1, 4, &dec, true)) return 1;
  if (make_tensor(m, "label_probs", 1, 1, B * K, C, &prb, true)) return 1;
  dp.dec_boxes = dec.d; dp.probs = prb.d;
  dp.cls_keep = (int*)m->alloc_f((size_t)B * (C - 1) * per_im, true);
  dp.cls_count = (int*)m->alloc_f((size_t)B * (C - 1), true);
  dp.out_boxes = m->alloc_f((size_t)B * per_im * 4, true);
  dp.out_probs = m->alloc_f((size_t)B * per_im, true);
  dp.out_labels = (int*)m->alloc_f((size_t)B * per_im, true);
  dp.out_valid = (int*)m->alloc_f(B, true);
  ODT_CHECK(dp.cls_keep && dp.cls_count && dp.out_boxes && dp.out_probs && dp.out_labels && dp.out_valid,
            "device allocation failed (detections)");
  { Op op; op.kind = OP_DETECT; m->ops.push_back(op); }

  // ---- appearance features: ROIAlign of the final boxes (models.py:971-973) + 7x7 mean
  m->final_feat = m->alloc_f((size_t)B * per_im * FC * 49, true);
  m->final_pooled = m->alloc_f((size_t)B * per_im * FC, true);
  ODT_CHECK(m->final_feat && m->final_pooled, "device allocation failed (features)");
  RoiAlignParams& rf = m->roi_final;
  rf.boxes = dp.out_boxes; rf.per_image = per_im; rf.count = dp.out_valid; rf.R_cap = B * per_im;
  rf.out_nhwc = nullptr; rf.out_nchw = m->final_feat; rf.pooled = m->final_pooled;
  rf.pack_rows = 1;                      // fpn_box_feat is [M,...] over the valid detections of all images
  { Op op; op.kind = OP_ROI_FINAL; m->ops.push_back(op); }

  // ---- Mask R-CNN head on the final boxes (--add_mask; models.py:932-962, 1173-1199): 14x14
  // ROIAlign, 4 x (3x3 conv + ReLU), 2x2 stride-2 transposed conv + ReLU, 1x1 conv to the
  // foreground classes, sigmoid of each detection's own class.  The transposed conv has no
  // overlap (kernel == stride), so it runs as ONE 1x1 conv to 4 * dim sub-pixel channels
  // (dy, dx, co); the following 1x1 conv treats [R,14,14,4*dim] as [R,14,56,dim] pixels and the
  // pixel shuffle to 28x28 happens in mask_select_kernel.
  if (cfg.add_mask) {
    ODT_CHECK(cfg.graph == ODT_GRAPH_SINGLE, "add_mask: built for the single-image graph only");
    const int MD = cfg.mask_dim > 0 ? cfg.mask_dim : 256;
    ODT_CHECK(MD % 32 == 0, "add_mask: mrcnn_head_dim must be a multiple of 32");
    const int R = B * per_im;
    Tensor mroi{};
    if (make_tensor(m, "mask_roi", R, 14, 14, FC, &mroi, true)) return 1;
    m->roi_mask = m->roi_final;
    m->roi_mask.out_nhwc = mroi.d; m->roi_mask.out_nchw = nullptr; m->roi_mask.pooled = nullptr;
    m->roi_mask.out_size = 14;
    { Op op; op.kind = OP_ROI_MASK; m->ops.push_back(op); }
    Tensor cur = mroi;
    int cin = FC;
    for (int k = 0; k < 4; ++k) {
      const std::string sc = "maskrcnn/fcn" + std::to_string(k);
      if (upload_conv(m, sc, 3, 3, cin, MD, false, &wt, &bias)) return 1;
      Tensor nx{};
      if (add_conv(m, sc, cur, cin, wt, bias, 3, 3, MD, 1, 1, 1, 1, 14, 14, 0, 0, nullptr, 0, true, MD, &nx,
                   "mask_fcn" + std::to_string(k))) return 1;
      cur = nx; cin = MD;
    }
    {   // Conv2DTranspose kernel [2,2,out,in] (nn.py:383-413) -> 1x1 conv [in][(dy,dx,out)]
      const HostTensor* wd = find_w(m, "maskrcnn/deconv/W"); const HostTensor* bd = find_w(m, "maskrcnn/deconv/b");
      ODT_CHECK(wd && bd, "missing maskrcnn/deconv variables");
      ODT_CHECK(wd->data.size() == (size_t)4 * MD * MD && bd->data.size() == (size_t)MD, "bad maskrcnn/deconv shapes");
      HostTensor v, vb; v.data.resize((size_t)MD * 4 * MD); vb.data.resize((size_t)4 * MD);
      for (int q = 0; q < 4; ++q)
        for (int co = 0; co < MD; ++co) {
          vb.data[(size_t)q * MD + co] = bd->data[co];
          for (int ci = 0; ci < MD; ++ci)
            v.data[(size_t)ci * 4 * MD + (size_t)q * MD + co] = wd->data[((size_t)q * MD + co) * MD + ci];
        }
      m->host_w["__maskdeconv/W"] = v; m->host_w["__maskdeconv/b"] = vb;
      if (upload_conv(m, "__maskdeconv", 1, 1, MD, 4 * MD, false, &wt, &bias)) return 1;
    }
    Tensor dc{};
    if (add_conv(m, "maskrcnn/deconv", cur, MD, wt, bias, 1, 1, 4 * MD, 1, 1, 0, 0, 14, 14, 0, 0, nullptr, 0, true,
                 4 * MD, &dc, "mask_deconv")) return 1;
    Tensor dv = dc;                       // [R,14,14,4*MD] viewed as [R,14,56,MD]
    dv.W = dv.w = 56; dv.C = MD; dv.c = MD;
    if (upload_conv(m, "maskrcnn/conv", 1, 1, MD, C - 1, false, &wt, &bias)) return 1;
    Tensor ml{};
    const int mld = (C - 1 + 3) / 4 * 4;
    if (add_conv(m, "maskrcnn/conv", dv, MD, wt, bias, 1, 1, C - 1, 1, 1, 0, 0, 14, 56, 0, 0, nullptr, 0, false, mld,
                 &ml, "mask_logits")) return 1;
    m->final_masks = m->alloc_f((size_t)R * 784, true);
    ODT_CHECK(m->final_masks != nullptr, "device allocation failed (masks)");
    MaskSelectParams& ms = m->mask_sel;
    ms.logits = ml.d; ms.ld = ml.C; ms.labels = dp.out_labels; ms.valid = dp.out_valid; ms.B = B;
    ms.per_image = per_im; ms.masks = m->final_masks;
    { Op op; op.kind = OP_MASK_SELECT; m->ops.push_back(op); }
  }
  if (attach_split_weights(m)) return 1;
  if (fuse_rpn_heads(m)) return 1;
  if (fuse_bottleneck_tails(m)) return 1;
  if (fuse_stem(m)) return 1;
  if (plan_arena(m)) return 1;
  if (upload_conv_records(m)) return 1;
  return 0;
}

}  // namespace odt
