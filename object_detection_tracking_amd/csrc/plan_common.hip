// Plan-building helpers shared by the FPN detector plan (plan_fpn.hip) and the EfficientDet plan (plan_effdet.hip):
// tensors (dedicated or arena-planned), weight upload with BN folding, conv ops, bf16x3 weight images, the RPN-head
// fusion and the activation arena.
#include "odt_model.hpp"

#define g_err (::odt::last_error())

namespace odt {

int ceil_div(int a, int b) { return (a + b - 1) / b; }

int make_tensor(odt_model* m, const std::string& name, int B, int H, int W, int C, Tensor* t, bool zero) {
  t->B = B; t->H = H; t->W = W; t->C = C; t->h = H; t->w = W; t->c = C;
  if (m->arena_on && !zero) {
    // (zero == true means "regions the kernels never write must read as zero": such a tensor cannot share memory)
    odt_model::VTensor v;
    v.bytes = (t->elems() * sizeof(float) + 255) & ~(size_t)255;
    v.voff = m->vnext;
    m->vnext += v.bytes;
    m->vt.push_back(v);
    m->virtual_tensor_bytes += v.bytes;
    t->d = reinterpret_cast<float*>(odt_model::kVirtBase + v.voff);
    if (!name.empty()) m->taps[name] = *t;
    return 0;
  }
  t->d = m->alloc_f(t->elems(), zero);
  ODT_CHECK(t->d != nullptr, "device allocation failed for " + name + ": " + g_err);
  m->dedicated_tensor_bytes += t->elems() * sizeof(float);
  if (!name.empty()) m->taps[name] = *t;
  return 0;
}

const HostTensor* find_w(odt_model* m, const std::string& name) {
  auto it = m->host_w.find(name);
  return it == m->host_w.end() ? nullptr : &it->second;
}

// Upload conv weights in [Cout][kh][kw][Cin] with optional folded BN; returns device ptrs.
int upload_conv(odt_model* m, const std::string& scope, int kh, int kw, int cin, int cout,
                bool has_bn, const float** wt_out, const float** bias_out) {
  const HostTensor* W = find_w(m, scope + "/W");
  ODT_CHECK(W != nullptr, "missing weight " + scope + "/W");
  ODT_CHECK(W->data.size() == (size_t)kh * kw * cin * cout,
            "bad shape for " + scope + "/W");
  std::vector<double> scale(cout, 1.0), shift(cout, 0.0);
  if (has_bn) {
    const HostTensor* g = find_w(m, scope + "/bn/gamma");
    const HostTensor* b = find_w(m, scope + "/bn/beta");
    const HostTensor* mu = find_w(m, scope + "/bn/mean/EMA");
    const HostTensor* var = find_w(m, scope + "/bn/variance/EMA");
    ODT_CHECK(g && b && mu && var, "missing BN variables for " + scope);
    for (int o = 0; o < cout; ++o) {   // tf.nn.batch_normalization, eps 1e-5 (nn.py:1771-1774)
      const double inv = (double)g->data[o] / std::sqrt((double)var->data[o] + 1e-5);
      scale[o] = inv;
      shift[o] = (double)b->data[o] - (double)mu->data[o] * inv;
    }
  } else {
    const HostTensor* b = find_w(m, scope + "/b");
    ODT_CHECK(b != nullptr, "missing bias " + scope + "/b");
    for (int o = 0; o < cout; ++o) shift[o] = b->data[o];
  }
  std::vector<float> wt((size_t)cout * kh * kw * cin), bias(cout);
  for (int y = 0; y < kh; ++y)
    for (int x = 0; x < kw; ++x)
      for (int i = 0; i < cin; ++i)
        for (int o = 0; o < cout; ++o)
          wt[(((size_t)o * kh + y) * kw + x) * cin + i] =
              (float)((double)W->data[(((size_t)y * kw + x) * cin + i) * cout + o] * scale[o]);
  for (int o = 0; o < cout; ++o) bias[o] = (float)shift[o];
  float* dw = m->alloc_f(wt.size(), false);
  float* db = m->alloc_f(bias.size(), false);
  ODT_CHECK(dw && db, "device allocation failed for weights of " + scope);
  ODT_HIP(hipMemcpy(dw, wt.data(), wt.size() * sizeof(float), hipMemcpyHostToDevice));
  ODT_HIP(hipMemcpy(db, bias.data(), bias.size() * sizeof(float), hipMemcpyHostToDevice));
  *wt_out = dw; *bias_out = db;
  return 0;
}

// conv3 + convshortcut of a stage-entry bottleneck as ONE 1x1 conv over the K-concatenated input
// [t2 | x]: weights [cout][cin_a + cin_b] with each part's BN folded in, bias = shift_a + shift_b
// (reference nn.py:503-521: conv3 -> BN, shortcut conv -> BN, add, ReLU).
int upload_conv_cat(odt_model* m, const std::string& sa, int cin_a, const std::string& sb, int cin_b, int cout,
                    const float** wt_out, const float** bias_out) {
  std::vector<float> wt((size_t)cout * (cin_a + cin_b));
  std::vector<double> shift(cout, 0.0);
  const std::string scopes[2] = {sa, sb};
  const int cins[2] = {cin_a, cin_b};
  int koff = 0;
  for (int part = 0; part < 2; ++part) {
    const std::string& scope = scopes[part];
    const int cin = cins[part];
    const HostTensor* W = find_w(m, scope + "/W");
    const HostTensor* g = find_w(m, scope + "/bn/gamma");
    const HostTensor* b = find_w(m, scope + "/bn/beta");
    const HostTensor* mu = find_w(m, scope + "/bn/mean/EMA");
    const HostTensor* var = find_w(m, scope + "/bn/variance/EMA");
    ODT_CHECK(W && g && b && mu && var, "missing variables for " + scope);
    ODT_CHECK(W->data.size() == (size_t)cin * cout, "bad shape for " + scope + "/W");
    for (int o = 0; o < cout; ++o) {
      const double inv = (double)g->data[o] / std::sqrt((double)var->data[o] + 1e-5);
      shift[o] += (double)b->data[o] - (double)mu->data[o] * inv;
      for (int i = 0; i < cin; ++i)
        wt[(size_t)o * (cin_a + cin_b) + koff + i] = (float)((double)W->data[(size_t)i * cout + o] * inv);
    }
    koff += cin;
  }
  std::vector<float> bias(cout);
  for (int o = 0; o < cout; ++o) bias[o] = (float)shift[o];
  float* dw = m->alloc_f(wt.size(), false);
  float* db = m->alloc_f(bias.size(), false);
  ODT_CHECK(dw && db, "device allocation failed for weights of " + sa);
  ODT_HIP(hipMemcpy(dw, wt.data(), wt.size() * sizeof(float), hipMemcpyHostToDevice));
  ODT_HIP(hipMemcpy(db, bias.data(), bias.size() * sizeof(float), hipMemcpyHostToDevice));
  *wt_out = dw; *bias_out = db;
  return 0;
}

int upload_raw(odt_model* m, const std::vector<float>& v, const float** out) {
  float* d = m->alloc_f(v.size(), false);
  ODT_CHECK(d != nullptr, "device allocation failed");
  ODT_HIP(hipMemcpy(d, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
  *out = d;
  return 0;
}

// Append a conv op.  `in` logical dims (h,w) bound the reads; output tensor is created here
// unless `out_existing` is given.
int add_conv(odt_model* m, const std::string& name, const Tensor& in, int cin, const float* wt,
             const float* bias, int kh, int kw, int cout, int stride, int dil, int pad_t, int pad_l,
             int Ho, int Wo, int oy, int ox, const Tensor* res, int res_mode, bool relu,
             int out_ldc, Tensor* out, const std::string& tap) {
  ConvOp c;
  c.name = name;
  ConvParams& p = c.p;
  std::memset(&p, 0, sizeof(p));
  if (out->d == nullptr) {
    if (make_tensor(m, tap, in.B, Ho + oy, Wo + ox, out_ldc, out, oy != 0 || ox != 0 || out_ldc != cout))
      return 1;
    out->c = cout;
  }
  p.in = in.d; p.wt = wt; p.bias = bias; p.out = out->d;
  p.res = res ? res->d : nullptr;
  p.B = in.B; p.H = in.h; p.W = in.w; p.Cin = cin; p.in_ldc = in.C;
  p.in_Ha = in.H; p.in_Wa = in.W;   // sliced views keep the allocation pitch
  p.Ho = Ho; p.Wo = Wo; p.Cout = cout;
  p.kh = kh; p.kw = kw; p.stride = stride; p.dil = dil; p.pad_t = pad_t; p.pad_l = pad_l;
  p.out_H = out->H; p.out_W = out->W; p.out_oy = oy; p.out_ox = ox; p.out_ldc = out->C;
  p.res_mode = res ? res_mode : 0;
  if (res) { p.res_H = res->H; p.res_W = res->W; p.res_ldc = res->C; }
  p.relu = relu ? 1 : 0;
  conv_prepare(p);
  m->convs.push_back(c);
  Op op;
  op.kind = OP_CONV;
  op.conv = (int)m->convs.size() - 1;
  m->ops.push_back(op);
  return 0;
}


// A handle's side streams (tail, H2D, D2H) are created with the highest stream priority.  Not for the arbitration: the
// runtime multiplexes all streams of ONE priority onto a few hardware queues in creation order, and a side stream that
// lands on the main stream's queue is serialised behind the forward it is meant to overlap (seen with the tracker's
// stream: tools/experiments/track_stream_collision.py); streams of another priority get queues of their own.
// ODT_SIDE_STREAM_PRIORITY=0: plain streams (A/B).
int create_side_stream(hipStream_t* s) {
  const bool flat = env_knob_off(K_SIDE_STREAM_PRIORITY);
  if (flat) { ODT_HIP(hipStreamCreateWithFlags(s, hipStreamNonBlocking)); return 0; }
  int least = 0, greatest = 0;
  ODT_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
  ODT_HIP(hipStreamCreateWithPriority(s, hipStreamNonBlocking, greatest));
  return 0;
}

// bf16-piece weight images (conv_split.hip) for the plan's convs that the split kernel takes
// the handle's conv policy: odt_config first, ODT_CONV_* debug overrides on top
ConvPolicy resolve_conv_policy(const odt_model* m) {
  ConvPolicy pol = conv_policy_default();
  if (m->cfg.conv_arith == ODT_ARITH_F32) pol.arith = 0;
  else if (m->cfg.conv_arith == ODT_ARITH_BF16X3) pol.arith = 1;
  if (m->cfg.conv_split_family >= 1 && m->cfg.conv_split_family <= 3) pol.family = m->cfg.conv_split_family;
  pol = conv_policy_from_env(pol);
  // (the EfficientDet graph keeps the bf16x3 kernels: its convs mostly lack 256-row x 128-column tiles, and its in-place
  // gates would need their own range bookkeeping)
  if (m->cfg.graph == ODT_GRAPH_EFFNET && pol.family == 2) pol.family = 3;
  return pol;
}

int attach_split_weights(odt_model* m) {
  const ConvPolicy pol = resolve_conv_policy(m);
  m->policy = pol;
  if (pol.arith == 0) return 0;
  find_overlap_points(m);
  // [live |max|][previous forward's |max|], kRangeSlots words each (odt_common.hpp)
  static_assert(2 * odt_model::kAmaxSlots * kAmaxWays == kRangeSlots, "the range records are laid out for one word per slot");
  m->amax_dev = reinterpret_cast<unsigned*>(m->alloc_f((size_t)2 * kRangeSlots, true));
  ODT_CHECK(m->amax_dev != nullptr, "device allocation failed (range slots)");
  if (m->range_host == nullptr) {
    // what the previous forward recorded, where the host can read it without a copy or a synchronisation (odt_range_health)
    ODT_HIP(hipHostMalloc((void**)&m->range_host, (size_t)kRangeSlots * sizeof(unsigned), hipHostMallocMapped));
    std::memset(m->range_host, 0, (size_t)kRangeSlots * sizeof(unsigned));
    ODT_HIP(hipHostGetDevicePointer((void**)&m->range_host_dev, m->range_host, 0));
  }
  m->range_slot_name.assign(kRangeSlots, std::string());
  m->range_baseline.assign(kRangeSlots, 0.f);
  // |max| slots: a tensor written by a split conv kernel gets one; a pooled / subsampled tensor shares its source's (its
  // values are a subset); anything else has none, and a conv reading it stays off the fp16x2 kernels.  Tail convs only
  // see slots filled in the tail (the trunk group is cleared by the next forward while the tail may still be running).
  std::map<const float*, int> slot_of;
  auto slot_ptr = [&](const float* t, bool tail_reader) -> const unsigned* {
    auto it = t != nullptr ? slot_of.find(t) : slot_of.end();
    if (it == slot_of.end()) return nullptr;
    if (tail_reader != (it->second >= odt_model::kAmaxSlots)) return nullptr;
    return m->amax_dev + (size_t)it->second * kAmaxWays;
  };
  std::map<std::pair<const float*, int>, const void*> made;      // the RPN conv is shared by the five levels: one image per layout
  size_t need_partial = 0;
  for (size_t oi = 0; oi < m->ops.size(); ++oi) {
    const Op& op = m->ops[oi];
    const bool tail = m->op_tail > 0 && oi >= m->op_tail;
    if (op.kind == OP_PRE && pol.family == 2) {        // the preprocess kernel records the range of the padded frames
      ODT_CHECK(m->amax_used[0] < odt_model::kAmaxSlots, "too many conv outputs for the range slots");
      slot_of[m->image_pad.d] = m->amax_used[0];
      m->range_slot_name[m->amax_used[0]] = "preprocessed frames";
      m->pre_amax = m->amax_dev + (size_t)(m->amax_used[0]++) * kAmaxWays;
      continue;
    }
    if (op.kind == OP_ROI_HEAD && pol.family == 2 && m->roi_head.out_nhwc != nullptr && !env_knob_off(K_ROI_AMAX)) {
      // the box head's RoI features: ROIAlign records their range, so that fc6 (K = 12544, a third of the box head's time on the
      // bf16x3 kernels) can take the fp16x2 kernels like every other layer (round 6; ODT_ROI_AMAX=0: A/B)
      const int g = tail ? 1 : 0;
      ODT_CHECK(m->amax_used[g] < odt_model::kAmaxSlots, "too many conv outputs for the range slots");
      const int slot = (tail ? odt_model::kAmaxSlots : 0) + m->amax_used[g]++;
      slot_of[m->roi_head.out_nhwc] = slot;
      m->roi_head.amax = m->amax_dev + (size_t)slot * kAmaxWays;
      m->range_slot_name[slot] = "roi_feat";
      continue;
    }
    if (op.kind == OP_POOL || op.kind == OP_SUB2) {
      auto it = slot_of.find(op.in.d);
      if (it != slot_of.end()) slot_of[op.out.d] = it->second; else slot_of.erase(op.out.d);
      continue;
    }
    if (op.kind != OP_CONV) continue;
    ConvOp& c = m->convs[op.conv];
    c.p.in_amax = slot_ptr(c.p.in, tail);
    c.p.in2_amax = slot_ptr(c.p.in2, tail);
    if (!conv_split_wanted(c.p, pol)) { slot_of.erase(c.p.out); continue; }      // exact-f32 kernel: records no range
    conv_split_choose(c.p, pol);
    const int K = c.p.kh * c.p.kw * c.p.Cin + (c.p.in2 != nullptr ? c.p.Cin2 : 0);
    const auto key = std::make_pair(c.p.wt, c.p.wt_split_kind * 1024 + c.p.wt_split_bn);
    auto it = made.find(key);
    if (it == made.end()) {
      float* img = m->alloc_f((conv_split_weight_bytes(c.p.Cout, K) + 3) / 4, false);
      ODT_CHECK(img != nullptr, "device allocation failed (split weights of " + c.name + ")");
      if (conv_make_split_weights(c.p, img, 0)) return 1;
      it = made.emplace(key, img).first;
    }
    c.p.wt_split = it->second;
    {
      // non-temporal hints on the loads of data that one workgroup reads once -- residual chunks (1), the activations of a 1x1
      // layer with a single n-tile (2) -- keep them from pushing the weights and the re-read tensors out of L2 / MALL: same-box
      // A/B at b=8 1080p 30.8 -> 29.8 ms of conv time, res4 conv1 275 -> 317 TF (profiles/r03_h2_nt_ab.txt; 4: large-output
      // stores, measured no gain).  ODT_CONV_NT overrides the mask (A/B).
      int nt = 3;
      nt = (int)env_knob_long(K_CONV_NT, nt);
      c.p.debug |= (nt & 7) << 10;
    }
    {
      const bool per_wave = env_knob(K_AMAX_PER_WAVE).c0 == '1';
      if (per_wave) c.p.debug |= 0x4000;      // A/B: range record per wave instead of per workgroup
    }
    if (c.p.wt_split_kind == 2) {
      c.p.h2_chinv = conv_h2_chinv(c.p.wt_split, c.p.Cout, K); ++m->convs_h2;
      const bool norot = env_knob_off(K_CONV_H2_ROT);
      if (norot) c.p.debug |= 0x100;          // A/B: every workgroup walks the K slices in the same order
    }
    need_partial = std::max(need_partial, conv_split_partial_bytes(c.p));
    // the output's slot (split-K layers: the combine pass records it, one atomic per block -- one per WAVE cost that short
    // memory-bound kernel 2x); only where the fp16x2 kernels may read it
    if (c.p.out != nullptr && pol.family == 2) {
      auto so = slot_of.find(c.p.out);
      int slot;
      if (so != slot_of.end() && (so->second >= odt_model::kAmaxSlots) == tail) slot = so->second;
      else {
        ODT_CHECK(m->amax_used[tail] < odt_model::kAmaxSlots, "too many conv outputs for the range slots");
        slot = (tail ? odt_model::kAmaxSlots : 0) + m->amax_used[tail]++;
        slot_of[c.p.out] = slot;
      }
      c.p.out_amax = m->amax_dev + (size_t)slot * kAmaxWays;
      if (m->range_slot_name[slot].empty()) m->range_slot_name[slot] = c.name;
    }
  }
  if (need_partial > 0) {
    // split-K scratch: the layers of one stream run one after another and share a buffer -- but the tail ops (box-head
    // FCs ...) of forward i run on the side stream UNDER the trunk of forward i+1 (tail overlap), so the two groups get
    // a buffer each
    std::vector<char> in_tail(m->convs.size(), 0);
    bool tail = false;
    for (const Op& op : m->ops) {
      if (op.kind == OP_PROPOSALS) tail = true;
      if (op.kind == OP_CONV && tail) in_tail[op.conv] = 1;
    }
    size_t need[2] = {0, 0};
    for (size_t i = 0; i < m->convs.size(); ++i)
      need[in_tail[i]] = std::max(need[in_tail[i]], conv_split_partial_bytes(m->convs[i].p));
    float* part[2] = {nullptr, nullptr};
    for (int g = 0; g < 2; ++g) {
      if (need[g] == 0) continue;
      part[g] = m->alloc_f((need[g] + 3) / 4, false);
      ODT_CHECK(part[g] != nullptr, "device allocation failed (split-K partial sums)");
    }
    for (size_t i = 0; i < m->convs.size(); ++i)
      if (conv_split_partial_bytes(m->convs[i].p) > 0) m->convs[i].p.partial = part[in_tail[i]];
  }
  ODT_HIP(hipDeviceSynchronize());
  return 0;
}


// ---- RPN head folded into the RPN conv's epilogue --------------------------------------------------------------------
// rpn/conv0@pL (3x3, 256 -> 256, ReLU) is followed by rpn/head@pL (1x1, 256 -> 3 logits || 12 deltas) and nothing else
// reads its output (models.py:979-1009).  Where the 3x3 conv runs on a conv_split3 kernel whose n-tile is the whole
// Cout (256) and has no split-K, the head is evaluated on the staged C tile in that kernel's epilogue (exact-f32 MFMA):
// the [M,256] tensor is neither written nor read back (2.1 GB at P2, b = 8) and the N = 15 launch disappears.
// ODT_FUSE_RPN_HEAD=0 keeps the two launches (A/B).  Called after attach_split_weights, before plan_arena.
int fuse_rpn_heads(odt_model* m) {
  m->conv_fused.assign(m->convs.size(), 0);
  if (env_knob_off(K_FUSE_RPN_HEAD)) return 0;
  const HostTensor* W = find_w(m, "__rpnhead/W");
  const HostTensor* Bv = find_w(m, "__rpnhead/b");
  if (W == nullptr || Bv == nullptr) return 0;
  const float *hw = nullptr, *hb = nullptr;
  for (size_t oi = 0; oi + 1 < m->ops.size(); ++oi) {
    Op& oa = m->ops[oi]; Op& ob = m->ops[oi + 1];
    if (oa.kind != OP_CONV || ob.kind != OP_CONV) continue;
    ConvOp& a = m->convs[oa.conv]; ConvOp& b = m->convs[ob.conv];
    if (a.name.compare(0, 10, "rpn/conv0@") != 0 || b.name.compare(0, 9, "rpn/head@") != 0) continue;
    const ConvParams& bp = b.p;
    ConvParams& ap = a.p;
    const bool ok = ap.wt_split != nullptr && (ap.wt_split_kind == 3 || ap.wt_split_kind == 2) && ap.wt_split_bn == 256 && ap.Cout == 256 && ap.splitk <= 1 &&
                    ap.res_mode == 0 && ap.in2 == nullptr && ap.relu <= 1 && bp.in == ap.out && bp.kh == 1 && bp.kw == 1 &&
                    bp.Cin == 256 && bp.Cout == 15 && bp.out_ldc == 16 && bp.stride == 1 && bp.res_mode == 0 && bp.relu == 0 &&
                    bp.out_oy == 0 && bp.out_ox == 0 && bp.out_H == bp.Ho && bp.out_W == bp.Wo && bp.Ho == ap.Ho && bp.Wo == ap.Wo &&
                    ap.out_oy == 0 && ap.out_ox == 0 && ap.out_H == ap.Ho && ap.out_W == ap.Wo && (int)W->data.size() == 256 * 15;
    if (!ok) continue;
    if (hw == nullptr) {
      std::vector<float> v((size_t)256 * 16, 0.f), vb(16, 0.f);
      for (int c = 0; c < 256; ++c)
        for (int j = 0; j < 15; ++j) v[(size_t)c * 16 + j] = W->data[(size_t)c * 15 + j];
      for (int j = 0; j < 15; ++j) vb[j] = Bv->data[j];
      if (upload_raw(m, v, &hw) || upload_raw(m, vb, &hb)) return 1;
    }
    ap.head_wt = hw; ap.head_bias = hb; ap.head_out = bp.out; ap.head_ldc = bp.out_ldc;
    ap.out = nullptr;                      // nothing else reads the 256-channel tensor
    ap.out_amax = nullptr;
    ob.skip = true;
    m->conv_fused[ob.conv] = 1;
  }
  return 0;
}

// ---- bottleneck tail folded into the 3x3 conv's kernel ----------------------------------------------------------------
// block/conv2 (3x3, ch -> ch, BN, ReLU) is read by block/conv3 (1x1, ch -> 4 ch, BN, + shortcut, ReLU) and by nothing else
// (nn.py:503-521).  Where conv2 runs on conv_h2k_kernel with its whole Cout in one 256-wide n-tile (res4 at b >= 4 @1080p)
// and conv3 is a plain dense 1x1 (conv_h2f_fusable), conv3 is evaluated from conv2's accumulators inside that kernel
// (conv_h2k.hip: h2f_tail): conv2's [M,256] tensor is neither written nor read back, conv3's launch -- prologue, A stream,
// its own lock-step store phase behind an idle matrix pipe -- disappears, and conv3's operand gets its power of two per
// pixel row instead of per tensor.  ODT_FUSE_BOTTLENECK=0 keeps the two launches (A/B).  Called after
// attach_split_weights, before plan_arena.
int fuse_bottleneck_tails(odt_model* m) {
  if (m->conv_fused.size() < m->convs.size()) m->conv_fused.resize(m->convs.size(), 0);
  const char e0 = env_knob(K_FUSE_BOTTLENECK).c0;      // A/B: 0 off | 1 only the 256-wide blocks (res4) | 2 + the 128-wide (res3) | otherwise every fusable block
  if (e0 == '0') return 0;
  const int min_cout = e0 == '1' ? 256 : (e0 == '2' ? 128 : 64);
  if (m->policy.arith == 0 || m->policy.family != 2) return 0;
  std::map<const float*, const void*> made;
  for (size_t oi = 0; oi + 1 < m->ops.size(); ++oi) {
    Op& oa = m->ops[oi]; Op& ob = m->ops[oi + 1];
    if (oa.kind != OP_CONV || ob.kind != OP_CONV || oa.skip || ob.skip) continue;
    ConvOp& a = m->convs[oa.conv]; ConvOp& b = m->convs[ob.conv];
    if (a.p.Cout < min_cout) continue;
    // (a 64-wide conv2 on the kw-reuse kernel's 512 x 64 tiles: the fused tail works on 256-row tiles -- same weight image)
    const int bm0 = a.p.wt_split_bm;
    if (a.p.Cout == 64 && a.p.wt_split_kind == 2 && a.p.wt_split_kwr == 1 && bm0 == 512) a.p.wt_split_bm = 256;
    if (!conv_h2f_fusable(a.p, b.p)) { a.p.wt_split_bm = bm0; continue; }
    // nothing else may read conv2's output (taps: a keep_taps handle exposes no stage tensor under this name, see add_conv)
    bool other = false;
    for (size_t k = 0; k < m->ops.size() && !other; ++k) {
      if (k == oi || k == oi + 1) continue;
      visit_op_ptrs(m, k, [&](auto& ptr) { if ((const void*)ptr == (const void*)a.p.out) other = true; });
    }
    for (const auto& kv : m->taps) if (kv.second.d == a.p.out) other = true;
    if (other) { a.p.wt_split_bm = bm0; continue; }
    const int K = b.p.Cin;
    auto it = made.find(b.p.wt);
    if (it == made.end()) {
      float* img = m->alloc_f((conv_h2f_weight_bytes(b.p.Cout, K) + 3) / 4, false);
      ODT_CHECK(img != nullptr, "device allocation failed (fused 1x1 weights of " + b.name + ")");
      if (conv_make_h2f_weights(b.p.wt, b.p.Cout, K, img, 0)) return 1;
      it = made.emplace(b.p.wt, img).first;
    }
    ConvParams& ap = a.p;
    ap.f_wt = it->second; ap.f_chinv = conv_h2f_chinv(it->second, b.p.Cout, K); ap.f_bias = b.p.bias;
    ap.f_res = b.p.res_mode != 0 ? b.p.res : nullptr; ap.f_res_ldc = b.p.res_ldc;
    ap.f_out = b.p.out; ap.f_out_ldc = b.p.out_ldc; ap.f_cout = b.p.Cout; ap.f_relu = b.p.relu; ap.f_out_amax = b.p.out_amax;
    ap.debug |= b.p.debug & 0x400;           // the residual's non-temporal hint travels with it
    if (env_knob_off(K_FUSE_ROT)) ap.debug |= 0x100;     // A/B: 0 = every workgroup walks the output column chunks in the same order
    ap.out = nullptr; ap.out_amax = nullptr;
    ob.skip = true;
    m->conv_fused[ob.conv] = 2;
    ++m->convs_h2f;
  }
  ODT_HIP(hipDeviceSynchronize());
  return 0;
}

// ---- conv0 + pool0 in one kernel ---------------------------------------------------------------------------------------
// pool0 reads conv0's map and nothing else does (nn.py:860-896): where conv0 runs on the fp16x2 family the pair becomes one
// launch of conv_stem_kernel (a 39 x 36 patch of the frame per 8 x 7 pooled pixels, split once into LDS; the [B,544,960,64] map
// -- 1.07 GB written and read back at b=8 -- never exists).  Bit-identical.  ODT_FUSE_STEM=0 keeps the two launches (A/B).
// Called after attach_split_weights, before plan_arena.
int fuse_stem(odt_model* m) {
  if (env_knob_off(K_FUSE_STEM)) return 0;
  if (m->policy.arith == 0 || m->policy.family != 2) return 0;
  for (size_t oi = 0; oi + 1 < m->ops.size(); ++oi) {
    Op& oa = m->ops[oi]; Op& ob = m->ops[oi + 1];
    if (oa.kind != OP_CONV || ob.kind != OP_POOL || oa.skip || ob.skip) continue;
    ConvParams& ap = m->convs[oa.conv].p;
    if (ob.in.d != ap.out || ap.out == nullptr || !conv_stem_fits(ap) || ap.out_oy != 0 || ap.out_ox != 0 || ap.out_H != ap.Ho || ap.out_W != ap.Wo ||
        ob.in.h != ap.Ho || ob.in.w != ap.Wo || ob.in.C != 64 || ob.out.H != (ap.Ho + 1 - 3) / 2 + 1 || ob.out.W != (ap.Wo + 1 - 3) / 2 + 1 ||
        ob.out.C < 64 || (double)ap.B * ob.out.H * ob.out.W * ob.out.C * 4.0 >= 2147483648.0) continue;
    bool other = false;                      // nothing else may read the conv map (a keep_taps handle exposes it as "conv0")
    for (size_t k = 0; k < m->ops.size() && !other; ++k) {
      if (k == oi || k == oi + 1) continue;
      visit_op_ptrs(m, k, [&](auto& ptr) { if ((const void*)ptr == (const void*)ap.out) other = true; });
    }
    // (arena handles keep no stage tensor: odt_tap refuses the transient ones)
    if (!m->arena_on) for (const auto& kv : m->taps) if (kv.second.d == ap.out) other = true;
    if (other) continue;
    // the conv map no longer exists: an arena handle must neither reserve memory for its stage name nor hand it out
    for (auto it = m->taps.begin(); it != m->taps.end();) { if (it->second.d == ap.out) it = m->taps.erase(it); else ++it; }
    ap.out = ob.out.d; ap.out_H = ob.out.H; ap.out_W = ob.out.W; ap.out_ldc = ob.out.C; ap.stem_pool = 1;
    if (env_knob(K_STEM_GRID).set) ap.debug |= ((int)env_knob(K_STEM_GRID).i & 0x3ff) << 20;   // test knob: workgroups of the launch
    ob.skip = true;
    m->stem_fused = 1;
  }
  return 0;
}

// ---- activation arena -----------------------------------------------------------------------------------------------
// ops [op_tail, end) of forward i (selection / ROIAlign / box head / NMS / features) may run on the side stream under ops
// [0, op_first_fpn) of forward i+1 (run_plan: tail overlap); 0 / 0 when the graph has no such split
void find_overlap_points(odt_model* m) {
  m->op_first_fpn = m->op_tail = 0;
  if (m->cfg.graph == ODT_GRAPH_EFFNET) return;
  for (size_t i = 0; i < m->ops.size(); ++i) {
    if (m->ops[i].kind == OP_PROPOSALS) { m->op_tail = i; break; }
    if (m->op_first_fpn == 0 && m->ops[i].kind == OP_CONV && m->convs[m->ops[i].conv].name.compare(0, 4, "fpn/") == 0)
      m->op_first_fpn = i;
  }
  if (m->op_tail == 0 || m->op_first_fpn == 0 || m->op_first_fpn >= m->op_tail) m->op_first_fpn = m->op_tail = 0;
}

// Lay the virtual stage tensors out in (at most) two arenas by live range and rewrite the plan's pointers.
// Region 1 holds what the tail ops touch (so that forward i's tail and forward i+1's early trunk never share memory);
// within a region a tensor takes the lowest offset not occupied by a tensor whose [first, last] range intersects its own
// (largest tensors first).  Called once, before the conv parameter records go to the device.
int plan_arena(odt_model* m) {
  if (!m->arena_on || m->vt.empty()) return 0;
  find_overlap_points(m);
  const int nops = (int)m->ops.size();
  for (int oi = 0; oi < nops; ++oi)
    visit_op_ptrs(m, (size_t)oi, [&](auto& p) {
      if (p == nullptr || !m->is_virtual((const void*)p)) return;
      odt_model::VTensor& v = m->vt[m->vt_index((const void*)p)];
      v.first = std::min(v.first, oi); v.last = std::max(v.last, oi);
    });
  const bool split = m->op_tail > 0;
  std::vector<char> tapped(m->vt.size(), 0);
  for (const auto& kv : m->taps)
    if (kv.second.d != nullptr && m->is_virtual(kv.second.d)) tapped[m->vt_index(kv.second.d)] = 1;
  for (size_t i = 0; i < m->vt.size(); ++i) {
    auto& v = m->vt[i];
    if (v.last < 0 && !tapped[i]) { v.bytes = 0; v.first = v.last = 0; }     // no op touches it (its producer was fused away)
    if (v.last < 0) { v.first = 0; v.last = nops - 1; }          // never referenced by an op (tap only): keep it apart
    v.region = split && v.last >= (int)m->op_tail ? 1 : 0;
    // forward i + 1 waits for forward i's tail only in front of op_first_fpn: what the tail reads must not be (re)written earlier
    ODT_CHECK(v.region == 0 || v.bytes == 0 || v.first >= (int)m->op_first_fpn || tapped[i],
              "activation arena: a tensor the tail ops read is produced before the FPN stage (tail overlap would race)");
    if (v.region == 1) v.last = nops - 1;                        // readable after the forward (appearance features / taps of the pyramid)
  }
  std::vector<int> order(m->vt.size());
  for (size_t i = 0; i < order.size(); ++i) order[i] = (int)i;
  std::sort(order.begin(), order.end(), [&](int a, int b) {
    return m->vt[a].bytes != m->vt[b].bytes ? m->vt[a].bytes > m->vt[b].bytes : a < b;
  });
  std::vector<int> placed;
  for (int id : order) {
    odt_model::VTensor& v = m->vt[id];
    // candidates: offset 0 and the end of every conflicting placed tensor; take the lowest that fits
    std::vector<std::pair<size_t, size_t>> busy;     // [begin, end) of placed tensors of this region alive at the same time
    for (int q : placed) {
      const odt_model::VTensor& u = m->vt[q];
      if (u.region == v.region && u.first <= v.last && v.first <= u.last) busy.emplace_back(u.off, u.off + u.bytes);
    }
    std::sort(busy.begin(), busy.end());
    size_t off = 0;
    for (const auto& b : busy) {
      if (off + v.bytes <= b.first) break;
      off = std::max(off, b.second);
    }
    v.off = off;
    m->arena_bytes[v.region] = std::max(m->arena_bytes[v.region], off + v.bytes);
    placed.push_back(id);
  }
  for (int r = 0; r < 2; ++r) {
    if (m->arena_bytes[r] == 0) continue;
    m->arena[r] = m->alloc_f(m->arena_bytes[r] / sizeof(float), false);
    ODT_CHECK(m->arena[r] != nullptr, "device allocation failed (activation arena): " + g_err);
  }
  auto fix = [&](auto& p) {
    if (p == nullptr || !m->is_virtual((const void*)p)) return;
    const odt_model::VTensor& v = m->vt[m->vt_index((const void*)p)];
    const size_t within = (size_t)((uintptr_t)p - odt_model::kVirtBase) - v.voff;
    p = reinterpret_cast<std::remove_reference_t<decltype(p)>>(reinterpret_cast<char*>(m->arena[v.region]) + v.off + within);
  };
  // taps first (they still hold virtual addresses: remember which of them do not outlive a forward)
  for (auto& kv : m->taps) {
    if (kv.second.d != nullptr && m->is_virtual(kv.second.d)) {
      const odt_model::VTensor& v = m->vt[m->vt_index(kv.second.d)];
      if (v.region == 0 && !(v.first == 0 && v.last == nops - 1)) m->transient_taps.insert(kv.first);
    }
    fix(kv.second.d);
  }
  for (size_t oi = 0; oi < m->ops.size(); ++oi) visit_op_ptrs(m, oi, fix);
  fix(m->image_pad.d);
  // every pointer a launched conv dereferences must be real memory by now (a field visit_op_ptrs does not enumerate would
  // still hold its virtual address)
  for (size_t oi = 0; oi < m->ops.size(); ++oi) {
    const Op& op = m->ops[oi];
    if (op.kind != OP_CONV || op.skip) continue;
    const ConvParams& c = m->convs[op.conv].p;
    for (const void* q : {(const void*)c.in, (const void*)c.res, (const void*)c.out, (const void*)c.in2, (const void*)c.head_out,
                          (const void*)c.f_res, (const void*)c.f_out})
      ODT_CHECK(q == nullptr || !m->is_virtual(q), "activation arena: unmapped pointer left in " + m->convs[op.conv].name);
  }
  return 0;
}

// conv parameter records in device memory (the kernels read their ConvParams from there).  The conv kernels address
// their tensors through buffer descriptors with 32-bit offsets: a launch whose input / output / residual tensor would
// reach 2 GiB (b = 16 @1080p: conv0's output, the res2 tensors, P2) is cut into equal batch ranges, each with its own
// record whose pointers start at that range's first image -- every image's arithmetic is unchanged.
// (ODT_CONV_CHUNK_BYTES: test knob that lowers the limit so that small plans exercise the chunked path.)
static int conv_batch_chunks(const ConvParams& p, double limit) {
  auto fits = [&](int n) {
    const double b = (double)(p.B / n);
    return b * p.in_Ha * p.in_Wa * p.in_ldc * 4.0 < limit && b * p.out_H * p.out_W * p.out_ldc * 4.0 < limit &&
           (p.res_mode == 0 || b * p.res_H * p.res_W * p.res_ldc * 4.0 < limit) &&
           (p.in2 == nullptr || b * p.in2_Ha * p.in2_Wa * p.in2_ldc * 4.0 < limit) &&
           (p.f_wt == nullptr || (b * p.Ho * p.Wo * p.f_out_ldc * 4.0 < limit && (p.f_res == nullptr || b * p.Ho * p.Wo * p.f_res_ldc * 4.0 < limit)));
  };
  for (int n = 1; n <= p.B; ++n)
    if (p.B % n == 0 && fits(n)) return n;
  return p.B;           // (one image per launch; launch_conv reports it if even that is too large)
}

int upload_conv_records(odt_model* m) {
  double limit = 2147483648.0;
  if (env_knob(K_CONV_CHUNK_BYTES).d > 0) limit = env_knob(K_CONV_CHUNK_BYTES).d;
  m->conv_recs.clear(); m->conv_rec0.clear(); m->conv_nrec.clear();
  for (const ConvOp& c : m->convs) {
    const ConvParams& p = c.p;
    const int n = conv_batch_chunks(p, limit), bc = p.B / n;
    m->conv_rec0.push_back((int)m->conv_recs.size());
    m->conv_nrec.push_back(n);
    for (int k = 0; k < n; ++k) {
      ConvParams q = p;
      const size_t b0 = (size_t)k * bc;
      q.B = bc;
      q.in = p.in + b0 * p.in_Ha * p.in_Wa * p.in_ldc;
      if (p.out != nullptr) q.out = p.out + b0 * p.out_H * p.out_W * p.out_ldc;
      if (p.res != nullptr) q.res = p.res + b0 * p.res_H * p.res_W * p.res_ldc;
      if (p.in2 != nullptr) q.in2 = p.in2 + b0 * p.in2_Ha * p.in2_Wa * p.in2_ldc;
      if (p.head_out != nullptr) q.head_out = p.head_out + b0 * p.Ho * p.Wo * p.head_ldc;
      if (p.f_out != nullptr) q.f_out = p.f_out + b0 * p.Ho * p.Wo * p.f_out_ldc;
      if (p.f_res != nullptr) q.f_res = p.f_res + b0 * p.Ho * p.Wo * p.f_res_ldc;
      m->conv_recs.push_back(q);
    }
    if (n > 1) ++m->chunked_convs;
  }
  m->bufs.emplace_back(new DevBuf());
  if (m->bufs.back()->alloc(m->conv_recs.size() * sizeof(ConvParams))) return 1;
  m->convs_dev = (ConvParams*)m->bufs.back()->p;
  ODT_HIP(hipMemcpy(m->convs_dev, m->conv_recs.data(), m->conv_recs.size() * sizeof(ConvParams), hipMemcpyHostToDevice));
  return 0;
}

}  // namespace odt
