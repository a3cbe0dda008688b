// libodt_hip.so -- host-side model of a handle: the static execution plan (ops, conv records, tensors), its memory
// (weights, activation arena, workspaces, ingest slots) and the helpers shared by the plan builders (plan_common.hip,
// plan_fpn.hip, plan_effdet.hip), the runtime (runtime.hip) and the stand-alone op entry points (op_shims.hip).
// Host code only; every device kernel lives in the kernel translation units (conv_*.hip, proposals.hip, ...).
#pragma once
#include "../../include/odt.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <type_traits>
#include <vector>

#include "odt_common.hpp"

namespace odt {

std::string& last_error();          // the calling thread's error message (odt_last_error)
int launch_subsample2(const float* in, int B, int H, int W, int C, float* out, int Ho, int Wo,
                      hipStream_t stream);

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  int alloc(size_t n) {
    bytes = n;
    if (n == 0) n = 256;
    ODT_HIP(hipMalloc(&p, n));
    return 0;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
  ~DevBuf() { release(); }
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
};

struct Tensor {      // NHWC device tensor; (h,w) = logical (possibly sliced) dims
  float* d = nullptr;
  int B = 0, H = 0, W = 0, C = 0;   // allocation dims (C = pixel stride)
  int h = 0, w = 0, c = 0;          // logical dims
  size_t elems() const { return (size_t)B * H * W * C; }
};

struct HostTensor {
  std::vector<float> data;
  std::vector<int64_t> shape;
};

struct ConvOp {
  ConvParams p;
  std::string name;
};

enum OpKind { OP_PRE, OP_CONV, OP_POOL, OP_SUB2, OP_PROPOSALS, OP_ROI_HEAD, OP_DETECT, OP_ROI_FINAL,
              OP_ROI_MASK, OP_MASK_SELECT, OP_PRE_RGB, OP_DW, OP_CMEAN, OP_CSCALE, OP_FUSE, OP_EFF_POST, OP_ROI_EFF, OP_SE_GATE,
              OP_SE_GATE_MEAN, OP_WSCALE, OP_MB_EXPAND_DW };
struct Op {
  OpKind kind;
  int conv = -1;        // index into convs
  Tensor in, out;
  DwConvParams dw{};    // OP_DW
  MbExpandDwParams mb{};   // OP_MB_EXPAND_DW
  FuseParams fuse{};    // OP_FUSE
  SeGateParams se{};    // OP_SE_GATE (aux2 = partial-sum scratch)
  float* aux = nullptr; // OP_CMEAN: means out [B,ldc]; OP_CSCALE / OP_WSCALE: gates in [B,ldc]
  const float* wt0 = nullptr;   // OP_WSCALE: the conv's unscaled weights [Cout][K] (conv = index of the conv whose weights are rebuilt)
  float* aux2 = nullptr;   // OP_CMEAN: partial-sum scratch
  int pad_t = 0, pad_l = 0;   // OP_PRE_RGB
  bool skip = false;          // OP_CONV folded into its producer's epilogue (fuse_rpn_heads): not launched
};

}  // namespace odt

using namespace odt;      // (odt_model is the C ABI's global handle type)

struct odt_model {
  odt_config cfg;
  int device = 0;
  hipStream_t own_stream = nullptr;
  bool finalized = false;
  std::map<std::string, HostTensor> host_w;
  std::vector<std::unique_ptr<DevBuf>> bufs;
  std::map<std::string, Tensor> taps;
  std::vector<ConvOp> convs;
  std::vector<char> conv_fused;      // convs[i] is evaluated inside another conv's kernel (no launch of its own): 1 RPN head, 2 bottleneck conv3
  ConvParams* convs_dev = nullptr;   // device copies of conv_recs
  std::vector<ConvParams> conv_recs; // launch records: convs[i] runs as records [conv_rec0[i], + conv_nrec[i]) (batch ranges,
  std::vector<int> conv_rec0, conv_nrec;   // more than one only where a tensor would reach 2 GiB: upload_conv_records)
  int chunked_convs = 0;
  // |max| slots of the tensors the split conv kernels produce (ConvParams::out_amax / in_amax: the fp16x2 kernels scale
  // their A operand by them).  Two groups, cleared at the start of the ops that fill them: [0, kAmaxSlots) for producers in
  // the trunk, [kAmaxSlots, 2 kAmaxSlots) for producers in the tail ops (which may run under the next forward's trunk).
  static constexpr int kAmaxSlots = 512;
  unsigned* amax_dev = nullptr;
  int amax_used[2] = {0, 0};
  unsigned* range_host = nullptr;        // pinned, device-mapped: the |max| records (kRangeSlots words) of the last completed forward
  std::vector<float> range_baseline;     // per slot: the |max| the host last accepted (odt_range_health)
  unsigned* range_host_dev = nullptr;    // ... its device address (amax_rotate_kernel writes it)
  std::vector<std::string> range_slot_name;
  int convs_h2 = 0;                  // convs on the fp16x2 kernels
  int stem_fused = 0;                // conv0 + pool0 run as one kernel (fuse_stem)
  int convs_h2f = 0;                 // ... of them with the following 1x1 conv folded into the kernel (fuse_bottleneck_tails)
  unsigned* pre_amax = nullptr;      // range slot of the preprocessed frames (OP_PRE records it; conv0 reads it)
  std::vector<Op> ops;
  // geometry
  int Hp = 0, Wp = 0;
  // proposal / head / detection state
  ProposalParams prop{};
  RoiAlignParams roi_head{}, roi_final{}, roi_mask{};
  MaskSelectParams mask_sel{};
  EffPostParams eff_post{};
  RoiAlignParams roi_eff{};
  int eff_filters = 0;
  struct Slot;
  // D2H of the small outputs enqueued behind the forward on the compute stream (odt_submit_ex without the big
  // [M,C,7,7] features): set by odt_submit_ex for the duration of run_plan
  Slot* d2h_slot = nullptr; int d2h_want = 0;
  ConvPolicy policy{};               // conv arithmetic / kernel-family policy of this handle (attach_split_weights)
  int mb_fused = 0;                  // EfficientNet: MBConv blocks whose expand + depthwise run as one kernel (effnet_mbconv.hip)
  // tail overlap: the selection / ROIAlign / box-head / NMS kernels of forward i (a few dozen workgroups each,
  // ~2 ms per 8-frame step) run on a side stream under the backbone of forward i+1.  The next forward's FPN stage
  // (the first op that overwrites what the tail reads: P2..P5, the RPN outputs) waits for the previous tail.
  std::vector<std::string> env_active;   // "ODT_NAME=value" of the overrides set when the handle's plan was built (odt_describe lists them)
  bool knob_tail_overlap_off = false;
  int tail_overlap = -1;             // -1 undecided | 0 off | 1 on (ODT_TAIL_OVERLAP=0 disables; own stream only)
  size_t op_first_fpn = 0, op_tail = 0;
  hipStream_t tail_stream = nullptr, done_stream = nullptr;
  hipEvent_t trunk_done = nullptr, tail_done = nullptr;
  bool tail_pending = false;
  unsigned long long forwards_enqueued = 0;
  int eff_scaled_h = 0, eff_scaled_w = 0;   // EfficientDet: size of the resized frame inside the padded input
  float* final_masks = nullptr;   // [B*per_im, 28, 28] (add_mask)
  DetectParams det{};
  Tensor image_pad, frames_dev;
  int src_h = 0, src_w = 0;          // source frame size (== cfg.height/width unless odt_set_source_size)
  DevBuf frames_src;                 // device staging for source frames larger than the plan's input
  float* anchors_dev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  float* final_feat = nullptr;    // [B*per_im, C, 7, 7] packed
  float* final_pooled = nullptr;  // [B*per_im, C]
  size_t frames_bytes = 0;
  // pipelined ingest: two slots
  struct Slot {
    void* pin_in = nullptr; size_t pin_in_bytes = 0;
    void* dev_in = nullptr; size_t dev_in_bytes = 0;
    float *pin_boxes = nullptr, *pin_probs = nullptr, *pin_feats = nullptr, *pin_pooled = nullptr;
    float* pin_masks = nullptr;
    int *pin_labels = nullptr, *pin_valid = nullptr;
    hipEvent_t h2d_done = nullptr, fwd_done = nullptr, d2h_done = nullptr;
    int ticket = -1;            // outstanding ticket or -1
    int ingest_armed = -1;      // ticket number odt_ingest_buffer() handed pin_in out for (-1: none), and its dtype
    int ingest_dtype = 0;
    int want = 0;               // ODT_WANT_* bits of the outstanding ticket
  } slot[2];
  hipStream_t copy_in = nullptr, copy_out = nullptr;
  int next_ticket = 0;
  hipEvent_t wait_before_detect = nullptr;   // D2H of the previous batch must finish before the tail rewrites outputs
  // profiling
  bool profile = false;
  std::vector<hipEvent_t> ev;
  hipEvent_t ev_total[2] = {nullptr, nullptr};
  double prof_conv_ms = 0, prof_conv_flops = 0, prof_total_ms = 0;
  std::vector<double> prof_layer_ms;
  int prof_launches = 0;

  // ---- activation arena (odt_config.keep_taps == 0): stage tensors get VIRTUAL addresses while the plan is built
  // (kVirtBase + running offset: never dereferenced), plan_arena() assigns each the lowest arena offset that no tensor
  // with an overlapping live range [first op, last op] occupies and rewrites every pointer of the plan.  Tensors that
  // must keep their contents between forwards (zero borders, zero pad channels, outputs) stay dedicated allocations.
  struct VTensor { size_t bytes = 0, voff = 0, off = 0; int first = 1 << 30, last = -1, region = 0; };
  static constexpr uintptr_t kVirtBase = 0x400000000000ull;
  bool arena_on = false;
  std::set<std::string> transient_taps;     // stage names whose memory is reused within a forward (arena mode)
  std::vector<VTensor> vt;
  size_t vnext = 0;
  float* arena[2] = {nullptr, nullptr};     // 0: trunk (live ranges end before the tail) | 1: read / written by the tail ops
  size_t arena_bytes[2] = {0, 0};
  size_t dedicated_tensor_bytes = 0, virtual_tensor_bytes = 0;
  bool is_virtual(const void* p) const {
    const uintptr_t a = (uintptr_t)p;
    return a >= kVirtBase && a < kVirtBase + vnext;
  }
  int vt_index(const void* p) const {       // the virtual tensor an address falls into
    const size_t o = (size_t)((uintptr_t)p - kVirtBase);
    size_t lo = 0, hi = vt.size();
    while (hi - lo > 1) { const size_t mid = (lo + hi) / 2; if (vt[mid].voff <= o) lo = mid; else hi = mid; }
    return (int)lo;
  }

  float* alloc_f(size_t elems, bool zero) {
    bufs.emplace_back(new DevBuf());
    if (bufs.back()->alloc(elems * sizeof(float))) return nullptr;
    if (zero && hipMemset(bufs.back()->p, 0, elems * sizeof(float)) != hipSuccess) return nullptr;
    return (float*)bufs.back()->p;
  }
};

namespace odt {

// ---- plan_common.hip
int ceil_div(int a, int b);
int make_tensor(odt_model* m, const std::string& name, int B, int H, int W, int C, Tensor* t, bool zero = false);
const HostTensor* find_w(odt_model* m, const std::string& name);
int upload_conv(odt_model* m, const std::string& scope, int kh, int kw, int cin, int cout, bool has_bn, const float** wt_out,
                const float** bias_out);
int upload_conv_cat(odt_model* m, const std::string& sa, int cin_a, const std::string& sb, int cin_b, int cout,
                    const float** wt_out, const float** bias_out);
int upload_raw(odt_model* m, const std::vector<float>& v, const float** out);
int add_conv(odt_model* m, const std::string& name, const Tensor& in, int cin, const float* wt, const float* bias, int kh,
             int kw, int cout, int stride, int dil, int pad_t, int pad_l, int Ho, int Wo, int oy, int ox, const Tensor* res,
             int res_mode, bool relu, int out_ldc, Tensor* out, const std::string& tap);
int create_side_stream(hipStream_t* s);
ConvPolicy resolve_conv_policy(const odt_model* m);
int attach_split_weights(odt_model* m);
int fuse_rpn_heads(odt_model* m);
int fuse_bottleneck_tails(odt_model* m);
int fuse_stem(odt_model* m);
void find_overlap_points(odt_model* m);
int plan_arena(odt_model* m);
int upload_conv_records(odt_model* m);
// ---- plan_fpn.hip / plan_effdet.hip
int build_plan(odt_model* m);
int build_plan_effnet(odt_model* m);
// ---- runtime.hip
int run_plan(odt_model* m, const void* frames, int dtype, int on_device, hipStream_t st);

// every device pointer op `oi` reads or writes, as a mutable reference (plan_arena: liveness, then the rewrite)
template <typename F>
void visit_op_ptrs(odt_model* m, size_t oi, F&& f) {
  Op& op = m->ops[oi];
  if (op.skip) return;
  auto roi = [&](RoiAlignParams& r) {
    for (auto& p : r.feat) f(p);
    f(r.boxes); f(r.out_nhwc); f(r.out_nchw); f(r.pooled);
  };
  f(op.in.d); f(op.out.d);
  switch (op.kind) {
    case OP_PRE: case OP_PRE_RGB: f(m->image_pad.d); break;
    case OP_CONV: { ConvParams& c = m->convs[op.conv].p; f(c.in); f(c.res); f(c.out); f(c.in2); f(c.head_out); f(c.f_res); f(c.f_out); break; }
    case OP_PROPOSALS: for (auto& l : m->prop.lvl) f(l.rpn); f(m->prop.props); break;
    case OP_ROI_HEAD: roi(m->roi_head); break;
    case OP_ROI_FINAL: roi(m->roi_final); break;
    case OP_ROI_MASK: roi(m->roi_mask); break;
    case OP_ROI_EFF: roi(m->roi_eff); break;
    case OP_DETECT: f(m->det.head_out); f(m->det.props); break;
    case OP_MASK_SELECT: f(m->mask_sel.logits); break;
    case OP_DW: f(op.dw.in); f(op.dw.out); for (auto& p : op.dw.lin) f(p); for (auto& p : op.dw.lout) f(p); break;
    case OP_MB_EXPAND_DW: f(op.mb.x); f(op.mb.out); break;
    case OP_FUSE: for (auto& p : op.fuse.in) f(p); f(op.fuse.out); break;
    case OP_EFF_POST: for (auto& p : m->eff_post.cls) f(p); for (auto& p : m->eff_post.box) f(p); break;
    case OP_CMEAN: case OP_CSCALE: case OP_SE_GATE: case OP_SE_GATE_MEAN: case OP_WSCALE: case OP_POOL: case OP_SUB2: break;
  }
}

}  // namespace odt
