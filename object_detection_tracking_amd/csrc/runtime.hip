// Runtime of a handle behind the C ABI (include/odt.h): handle life cycle, weight ingest, the forward (op list on the
// handle's streams, tail overlap), pipelined ingest (odt_submit_ex / odt_collect), outputs, taps, profiling, describe.
#include <cmath>
#include <thread>

#include "odt_model.hpp"

namespace odt {

std::string& last_error() {
  static thread_local std::string err;
  return err;
}
void set_error(const std::string& msg) { last_error() = msg; }

}  // namespace odt

using namespace odt;
#define g_err (::odt::last_error())

namespace odt {

// the op list of the static plan (ops [begin, end)), launched on `st`
static int run_ops(odt_model* m, const void* src, int dtype, hipStream_t st, size_t* ev_io, size_t begin = 0,
                   size_t end = (size_t)-1) {
  const odt_config& cfg = m->cfg;
  size_t ev_i = *ev_io;
  for (size_t oi = begin; oi < end && oi < m->ops.size(); ++oi) {
    const Op& op = m->ops[oi];
    switch (op.kind) {
      case OP_PRE:
        if (m->src_h == cfg.height && m->src_w == cfg.width) {
          if (launch_preprocess(src, dtype, cfg.batch, cfg.height, cfg.width, 3, 3, m->Hp, m->Wp, m->image_pad.d, st, m->pre_amax)) return 1;
        } else if (launch_preprocess_resize(src, dtype, cfg.batch, m->src_h, m->src_w, cfg.height, cfg.width, 3, 3,
                                            m->Hp, m->Wp, m->image_pad.d, st, m->pre_amax)) {
          return 1;
        }
        break;
      case OP_CONV: {
        if (op.skip) break;
        const ConvOp& c = m->convs[op.conv];
        if (m->profile) ODT_HIP(hipEventRecord(m->ev[2 * op.conv], st));
        for (int k = 0; k < m->conv_nrec[op.conv]; ++k) {
          const int r = m->conv_rec0[op.conv] + k;
          if (launch_conv(m->conv_recs[r], st, m->convs_dev + r)) { g_err = c.name + ": " + g_err; return 1; }
        }
        if (m->profile) ODT_HIP(hipEventRecord(m->ev[2 * op.conv + 1], st));
        break;
      }
      case OP_POOL:
        if (op.skip) break;
        if (launch_maxpool3x3s2(op.in.d, op.in.B, op.in.h, op.in.w, op.in.C, op.out.d, op.out.H, op.out.W, st)) return 1;
        break;
      case OP_SUB2:
        if (launch_subsample2(op.in.d, op.in.B, op.in.H, op.in.W, op.in.C, op.out.d, op.out.H, op.out.W, st)) return 1;
        break;
      case OP_PROPOSALS:
        if (launch_proposals(m->prop, st)) return 1;
        break;
      case OP_ROI_HEAD:
        if (launch_roi_align(m->roi_head, st)) return 1;
        break;
      case OP_DETECT:
        if (m->wait_before_detect) {
          ODT_HIP(hipStreamWaitEvent(st, m->wait_before_detect, 0));
          m->wait_before_detect = nullptr;
        }
        if (launch_detections(m->det, st)) return 1;
        break;
      case OP_ROI_FINAL:
        if (launch_roi_align(m->roi_final, st)) return 1;
        break;
      case OP_ROI_MASK:
        if (launch_roi_align(m->roi_mask, st)) return 1;
        break;
      case OP_PRE_RGB:
        if (m->src_h == cfg.height && m->src_w == cfg.width) {
          if (launch_preprocess_rgb(src, dtype, cfg.batch, cfg.height, cfg.width, op.pad_t, op.pad_l, m->Hp, m->Wp,
                                    m->image_pad.d, st)) return 1;
        } else if (launch_preprocess_rgb_resize(src, dtype, cfg.batch, m->src_h, m->src_w, m->eff_scaled_h,
                                                m->eff_scaled_w, op.pad_t, op.pad_l, m->Hp, m->Wp, m->image_pad.d, st)) {
          return 1;
        }
        break;
      case OP_DW:
        if (launch_dwconv(op.dw, st)) return 1;
        break;
      case OP_MB_EXPAND_DW:
        if (launch_mbconv_expand_dw(op.mb, st)) return 1;
        break;
      case OP_CMEAN:
        if (launch_channel_mean(op.in.d, op.in.B, op.in.h * op.in.w, op.in.C, op.aux2, op.aux, st)) return 1;
        break;
      case OP_CSCALE:
        if (launch_channel_scale(op.in.d, op.aux, op.in.B, op.in.h * op.in.w, op.in.C, st)) return 1;
        break;
      case OP_FUSE:
        if (launch_bifpn_fuse(op.fuse, st)) return 1;
        break;
      case OP_EFF_POST:
        if (launch_effdet_post(m->eff_post, st)) return 1;
        break;
      case OP_SE_GATE:
        if (launch_se_gate(op.in.d, op.se, op.in.B, op.aux2, st)) return 1;
        break;
      case OP_SE_GATE_MEAN:       // the partial sums came out of the depthwise kernel
        if (launch_se_gate_from_parts(op.se, cfg.batch, st)) return 1;
        break;
      case OP_WSCALE: {           // batch 1: the gate goes into the projection's weights instead of a pass over the activations
        const ConvParams& cp = m->convs[op.conv].p;
        const int K = cp.Cin;
        if (cp.wt_split != nullptr) {
          if (conv_make_split_weights(cp, const_cast<void*>(cp.wt_split), st, op.wt0, op.aux)) return 1;
        } else if (conv_scale_weights(op.wt0, op.aux, cp.Cout, K, const_cast<float*>(cp.wt), st)) {
          return 1;
        }
        break;
      }
      case OP_ROI_EFF:
        if (launch_roi_align(m->roi_eff, st)) return 1;
        break;
      case OP_MASK_SELECT:
        if (launch_mask_select(m->mask_sel, st)) return 1;
        break;
    }
  }
  *ev_io = ev_i;
  return 0;
}

static size_t input_bytes(const odt_model* m, int dtype) {
  return (size_t)m->cfg.batch * m->src_h * m->src_w * 3 * (dtype == ODT_DTYPE_U8 ? 1 : 4);
}

// pageable -> pinned staging copy of a batch of frames (odt_submit_ex): 50 MB at 8 x 1080p, on up to four threads -- the call's
// host time 1.93 -> 1.09 ms on the evidence box's EPYC (profiles/r06_submit_host_time.txt): host time the tracking loop has
// other uses for (detect + track through Python objects runs within 10 % of being host-bound).  Small inputs stay on the
// calling thread (a thread costs ~30 us to start).
static void staging_copy(void* dst, const void* src, size_t n) {
  constexpr size_t kPerThread = (size_t)8 << 20;
  const unsigned hw = std::thread::hardware_concurrency();
  size_t nt = n / kPerThread;
  if (nt > 4) nt = 4;
  if (hw != 0 && nt > hw) nt = hw;
  if (nt <= 1) { std::memcpy(dst, src, n); return; }
  const size_t chunk = ((n / nt) + 4095) & ~(size_t)4095;
  std::vector<std::thread> th;
  for (size_t i = 1; i < nt; ++i) {
    const size_t off = i * chunk, len = off >= n ? 0 : (i + 1 == nt ? n - off : (off + chunk > n ? n - off : chunk));
    if (len) th.emplace_back([=] { std::memcpy(static_cast<char*>(dst) + off, static_cast<const char*>(src) + off, len); });
  }
  std::memcpy(dst, src, chunk < n ? chunk : n);
  for (auto& t : th) t.join();
}

// the small outputs of a slot's forward, copied behind it on the same stream (odt_submit_ex without ODT_WANT_FEATS)
static int enqueue_small_d2h(odt_model* m, hipStream_t st) {
  if (m->d2h_slot == nullptr) return 0;
  odt_model::Slot& sl = *m->d2h_slot;
  const size_t B = m->cfg.batch, per = m->cfg.result_per_im, FC = m->cfg.fpn_channels;
  ODT_HIP(hipMemcpyAsync(sl.pin_valid, m->det.out_valid, B * sizeof(int), hipMemcpyDeviceToHost, st));
  ODT_HIP(hipMemcpyAsync(sl.pin_boxes, m->det.out_boxes, B * per * 4 * sizeof(float), hipMemcpyDeviceToHost, st));
  ODT_HIP(hipMemcpyAsync(sl.pin_probs, m->det.out_probs, B * per * sizeof(float), hipMemcpyDeviceToHost, st));
  ODT_HIP(hipMemcpyAsync(sl.pin_labels, m->det.out_labels, B * per * sizeof(int), hipMemcpyDeviceToHost, st));
  if (m->d2h_want & ODT_WANT_POOLED)
    ODT_HIP(hipMemcpyAsync(sl.pin_pooled, m->final_pooled, B * per * FC * sizeof(float), hipMemcpyDeviceToHost, st));
  if ((m->d2h_want & ODT_WANT_MASKS) && m->final_masks)
    ODT_HIP(hipMemcpyAsync(sl.pin_masks, m->final_masks, B * per * 784 * sizeof(float), hipMemcpyDeviceToHost, st));
  return 0;
}

// profiling: close the step's total event on the stream the forward ends on, wait, accumulate the per-conv times
static int finish_profile(odt_model* m, hipStream_t st) {
  ODT_HIP(hipEventRecord(m->ev_total[1], st));
  ODT_HIP(hipStreamSynchronize(st));
  double ms = 0, fl = 0;
  int launched = 0;
  for (size_t i = 0; i < m->convs.size(); ++i) {
    if (m->conv_fused[i]) continue;          // evaluated inside its producer's epilogue (its FLOPs are counted there)
    float t = 0;
    ODT_HIP(hipEventElapsedTime(&t, m->ev[2 * i], m->ev[2 * i + 1]));
    ms += t; fl += conv_flops(m->convs[i].p);
    if (m->prof_layer_ms.size() < m->convs.size()) m->prof_layer_ms.resize(m->convs.size(), 0.0);
    m->prof_layer_ms[i] += t;
    ++launched;
  }
  float tt = 0;
  ODT_HIP(hipEventElapsedTime(&tt, m->ev_total[0], m->ev_total[1]));
  m->prof_conv_ms += ms; m->prof_conv_flops += fl; m->prof_launches += launched;
  m->prof_total_ms += tt;
  return 0;
}

// the |max| slots the split conv kernels of a group of ops fill (0: trunk, 1: tail) start a forward at zero
// start of a forward (of its tail): a group's |max| records move to the "previous" row and to host-visible memory, the live words
// back to zero (this was a memset)
__global__ void __launch_bounds__(256) amax_rotate_kernel(unsigned* __restrict__ slots, int first, int n, unsigned* __restrict__ host) {
  const int i = first + (int)(blockIdx.x * 256 + threadIdx.x);
  if (i >= first + n) return;
  const unsigned a = slots[i];
  slots[kRangeSlots + i] = a;
  slots[i] = 0u;
  if (host != nullptr) host[i] = a;
}

static int clear_amax(odt_model* m, int group, hipStream_t st) {
  if (m->amax_dev == nullptr || m->amax_used[group] == 0) return 0;
  const int n = m->amax_used[group];
  hipLaunchKernelGGL(amax_rotate_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, m->amax_dev, group * odt_model::kAmaxSlots, n,
                     m->range_host_dev);
  ODT_HIP(hipGetLastError());
  return 0;
}

int run_plan(odt_model* m, const void* frames, int dtype, int on_device, hipStream_t st) {
  const odt_config& cfg = m->cfg;
  ODT_CHECK(m->finalized, "odt_forward: call odt_finalize_weights first");
  ODT_CHECK(frames != nullptr, "odt_forward: null frames");
  ODT_CHECK(dtype == ODT_DTYPE_U8 || dtype == ODT_DTYPE_F32, "odt_forward: bad dtype");
  ODT_HIP(hipSetDevice(m->device));
  const void* src = frames;
  if (!on_device) {
    const size_t n = input_bytes(m, dtype);
    void* stage = n <= m->frames_bytes ? (void*)m->frames_dev.d : m->frames_src.p;
    ODT_HIP(hipMemcpyAsync(stage, frames, n, hipMemcpyHostToDevice, st));
    src = stage;
  }
  size_t ev_i = 0;
  if (m->profile) {
    while (m->ev.size() < 2 * m->convs.size()) { hipEvent_t e; ODT_HIP(hipEventCreate(&e)); m->ev.push_back(e); }
    for (int i = 0; i < 2; ++i) if (!m->ev_total[i]) ODT_HIP(hipEventCreate(&m->ev_total[i]));
    ODT_HIP(hipEventRecord(m->ev_total[0], st));
  }
  // ---- tail overlap (own stream only: a caller's stream must see the whole forward in stream order)
  if (m->tail_overlap < 0) {
    m->tail_overlap = (cfg.graph != ODT_GRAPH_EFFNET && !m->knob_tail_overlap_off && cfg.tail_overlap >= 0) ? 1 : 0;
    find_overlap_points(m);
    if (m->op_tail == 0) m->tail_overlap = 0;
  }
  m->done_stream = st;
  ++m->forwards_enqueued;
  if (m->tail_overlap == 1 && st == m->own_stream) {
    if (!m->tail_stream) {
      if (create_side_stream(&m->tail_stream)) return 1;
      ODT_HIP(hipEventCreateWithFlags(&m->trunk_done, hipEventDisableTiming));
      ODT_HIP(hipEventCreateWithFlags(&m->tail_done, hipEventDisableTiming));
    }
    if (clear_amax(m, 0, st)) return 1;
    if (run_ops(m, src, dtype, st, &ev_i, 0, m->op_first_fpn)) return 1;
    if (m->tail_pending) ODT_HIP(hipStreamWaitEvent(st, m->tail_done, 0));
    if (run_ops(m, src, dtype, st, &ev_i, m->op_first_fpn, m->op_tail)) return 1;
    ODT_HIP(hipEventRecord(m->trunk_done, st));
    ODT_HIP(hipStreamWaitEvent(m->tail_stream, m->trunk_done, 0));
    if (clear_amax(m, 1, m->tail_stream)) return 1;
    if (run_ops(m, src, dtype, m->tail_stream, &ev_i, m->op_tail)) return 1;
    if (enqueue_small_d2h(m, m->tail_stream)) return 1;
    ODT_HIP(hipEventRecord(m->tail_done, m->tail_stream));
    m->tail_pending = true;
    m->done_stream = m->tail_stream;
    if (m->profile) return finish_profile(m, m->tail_stream);
    return 0;
  }
  if (m->tail_pending) {          // a forward on another stream after overlapped ones: order it behind the last tail
    ODT_HIP(hipStreamWaitEvent(st, m->tail_done, 0));
    m->tail_pending = false;
  }
  if (clear_amax(m, 0, st) || clear_amax(m, 1, st)) return 1;
  if (run_ops(m, src, dtype, st, &ev_i)) return 1;
  if (enqueue_small_d2h(m, st)) return 1;
  if (m->profile) return finish_profile(m, st);
  return 0;
}

}  // namespace odt

extern "C" {

const char* odt_last_error(void) { return g_err.c_str(); }

int odt_device_count(int* count) {
  ODT_CHECK(count != nullptr, "null argument");
  ODT_HIP(hipGetDeviceCount(count));
  return 0;
}

int odt_create(const odt_config* cfg, int device, odt_handle* out) {
  ODT_CHECK(cfg && out, "odt_create: null argument");
  ODT_CHECK(cfg->batch >= 1 && cfg->height >= 64 && cfg->width >= 64, "odt_create: bad geometry");
  ODT_CHECK(cfg->graph == ODT_GRAPH_SINGLE || cfg->graph == ODT_GRAPH_MULTI || cfg->graph == ODT_GRAPH_EFFNET,
            "odt_create: bad graph");
  if (cfg->graph != ODT_GRAPH_EFFNET) {
    ODT_CHECK(cfg->rpn_topk >= 1 && cfg->rpn_topk <= kMaxTopKBig, "odt_create: rpn_topk must be in [1,4096]");
    ODT_CHECK(cfg->fpn_channels % 32 == 0 && cfg->head_dim % 32 == 0, "odt_create: channel counts must be multiples of 32");
    ODT_CHECK(cfg->graph == ODT_GRAPH_MULTI || cfg->batch == 1,
              "odt_create: the Mask_RCNN_FPN graph is single-image (obj_detect_tracking.py:241-242)");
  }
  int n = 0;
  ODT_HIP(hipGetDeviceCount(&n));
  ODT_CHECK(device >= 0 && device < n, "odt_create: no such device");
  ODT_HIP(hipSetDevice(device));
  knobs_reload();
  std::unique_ptr<odt_model> m(new odt_model());
  m->env_active = knobs_active();
  m->knob_tail_overlap_off = env_knob_off(K_TAIL_OVERLAP);
  m->cfg = *cfg;
  m->device = device;
  // (non-blocking, round 6: a blocking stream orders itself against every null-stream operation of the process -- another
  // handle's synchronous copy, the host framework's default stream -- which serialises handles that should run side by side)
  ODT_HIP(hipStreamCreateWithFlags(&m->own_stream, hipStreamNonBlocking));
  *out = m.release();
  return 0;
}

int odt_destroy(odt_handle h) {
  if (!h) return 0;
  (void)hipSetDevice(h->device);
  (void)hipDeviceSynchronize();
  for (auto e : h->ev) (void)hipEventDestroy(e);
  for (auto e : h->ev_total) if (e) (void)hipEventDestroy(e);
  for (auto& sl : h->slot) {
    if (sl.pin_in) (void)hipHostFree(sl.pin_in);
    if (sl.dev_in) (void)hipFree(sl.dev_in);
    for (void* q : {(void*)sl.pin_masks, (void*)sl.pin_boxes, (void*)sl.pin_probs, (void*)sl.pin_feats, (void*)sl.pin_pooled,
                    (void*)sl.pin_labels, (void*)sl.pin_valid})
      if (q) (void)hipHostFree(q);
    for (hipEvent_t e : {sl.h2d_done, sl.fwd_done, sl.d2h_done}) if (e) (void)hipEventDestroy(e);
  }
  if (h->range_host) (void)hipHostFree(h->range_host);
  if (h->copy_in) (void)hipStreamDestroy(h->copy_in);
  if (h->copy_out) (void)hipStreamDestroy(h->copy_out);
  if (h->tail_stream) (void)hipStreamDestroy(h->tail_stream);
  if (h->trunk_done) (void)hipEventDestroy(h->trunk_done);
  if (h->tail_done) (void)hipEventDestroy(h->tail_done);
  if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
  delete h;
  return 0;
}

int odt_load_tensor(odt_handle h, const char* name, const float* data, const int64_t* shape, int rank) {
  ODT_CHECK(h && name && data && shape && rank >= 1 && rank <= 4, "odt_load_tensor: bad argument");
  ODT_CHECK(!h->finalized, "odt_load_tensor: weights already finalized");
  HostTensor t;
  size_t n = 1;
  for (int i = 0; i < rank; ++i) { t.shape.push_back(shape[i]); n *= (size_t)shape[i]; }
  t.data.assign(data, data + n);
  h->host_w[name] = std::move(t);
  return 0;
}

int odt_finalize_weights(odt_handle h) {
  ODT_CHECK(h != nullptr, "null handle");
  ODT_CHECK(!h->finalized, "weights already finalized");
  ODT_HIP(hipSetDevice(h->device));
  knobs_reload();          // the plan is built under the environment of THIS call (tests flip A/B knobs between handles)
  h->env_active = knobs_active();
  h->knob_tail_overlap_off = env_knob_off(K_TAIL_OVERLAP);
  if (build_plan(h)) return 1;
  h->conv_fused.resize(h->convs.size(), 0);
  ODT_HIP(hipDeviceSynchronize());
  h->finalized = true;
  h->host_w.clear();
  return 0;
}

int odt_forward_async(odt_handle h, const void* frames, int dtype, int on_device, void* stream) {
  ODT_CHECK(h != nullptr, "null handle");
  hipStream_t st = stream ? (hipStream_t)stream : h->own_stream;
  return run_plan(h, frames, dtype, on_device, st);
}

int odt_synchronize(odt_handle h) {
  ODT_CHECK(h != nullptr, "null handle");
  ODT_HIP(hipSetDevice(h->device));
  ODT_HIP(hipDeviceSynchronize());
  return 0;
}

int odt_forward(odt_handle h, const void* frames, int dtype, int on_device, void* stream, odt_outputs* out) {
  ODT_CHECK(h != nullptr && out != nullptr, "null argument");
  ODT_CHECK(h->cfg.graph != ODT_GRAPH_EFFNET || h->cfg.eff_det >= 0,
            "odt_forward: the backbone-only graph has no detection outputs (odt_forward_async + odt_tap)");
  hipStream_t st = stream ? (hipStream_t)stream : h->own_stream;
  if (run_plan(h, frames, dtype, on_device, st)) return 1;
  return odt_read_outputs(h, out);
}

static hipError_t d2h_on(hipStream_t st, void* dst, const void* src, size_t n, hipMemcpyKind kind) {
  const hipError_t e = hipMemcpyAsync(dst, src, n, kind, st);
  return e != hipSuccess ? e : hipStreamSynchronize(st);
}

int odt_read_outputs(odt_handle h, odt_outputs* out) {
  ODT_CHECK(h != nullptr && out != nullptr, "null argument");
  ODT_CHECK(h->finalized && h->forwards_enqueued > 0, "odt_read_outputs: no forward has been enqueued on this handle");
  ODT_CHECK(h->cfg.graph != ODT_GRAPH_EFFNET || h->cfg.eff_det >= 0,
            "odt_read_outputs: the backbone-only graph has no detection outputs (odt_tap)");
  ODT_HIP(hipSetDevice(h->device));
  hipStream_t st = h->done_stream;     // (the tail may have run on the handle's side stream)
  // (round 6) every copy below is ordered on THIS handle's stream and waited for there: a plain hipMemcpy goes through the
  // null stream, which waits for every other blocking stream of the device -- i.e. for the forwards other handles have in
  // flight (frames in flight on replica handles, several streams per GPU: a collect serialised them all)
  if (h->cfg.graph == ODT_GRAPH_EFFNET) {
    // EfficientDet outputs (efficientdet_wrapper.py:28-35): boxes [R,4] x1y1x2y2 (scaled), probs,
    // labels 1..90, pooled = fpn_box_feat [R, fpn_num_filters]
    ODT_HIP(hipStreamSynchronize(st));
    const EffPostParams& ep = h->eff_post;
    const int B = ep.B, per = ep.max_out, F = h->eff_filters;
    std::vector<int> valid(B);
    ODT_HIP(d2h_on(st, valid.data(), ep.out_valid, B * sizeof(int), hipMemcpyDeviceToHost));
    int total = 0;
    for (int b = 0; b < B; ++b) total += valid[b];
    if (out->valid) std::memcpy(out->valid, valid.data(), B * sizeof(int));
    if (out->boxes) ODT_HIP(d2h_on(st, out->boxes, ep.out_boxes, (size_t)B * per * 4 * sizeof(float), hipMemcpyDeviceToHost));
    if (out->probs) ODT_HIP(d2h_on(st, out->probs, ep.out_scores, (size_t)B * per * sizeof(float), hipMemcpyDeviceToHost));
    if (out->labels) ODT_HIP(d2h_on(st, out->labels, ep.out_labels, (size_t)B * per * sizeof(int), hipMemcpyDeviceToHost));
    ODT_CHECK(out->feats == nullptr && out->masks == nullptr, "odt_forward: EfficientDet returns pooled [R, filters] features only");
    if (out->pooled && total > 0)
      ODT_HIP(d2h_on(st, out->pooled, h->final_pooled, (size_t)total * F * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
  }
  ODT_HIP(hipStreamSynchronize(st));
  const odt_config& cfg = h->cfg;
  const int B = cfg.batch, per = cfg.result_per_im, FC = cfg.fpn_channels;
  std::vector<int> valid(B);
  ODT_HIP(d2h_on(st, valid.data(), h->det.out_valid, B * sizeof(int), hipMemcpyDeviceToHost));
  int total = 0;
  for (int b = 0; b < B; ++b) total += valid[b];
  if (out->valid) std::memcpy(out->valid, valid.data(), B * sizeof(int));
  if (out->boxes) ODT_HIP(d2h_on(st, out->boxes, h->det.out_boxes, (size_t)B * per * 4 * sizeof(float), hipMemcpyDeviceToHost));
  if (out->probs) ODT_HIP(d2h_on(st, out->probs, h->det.out_probs, (size_t)B * per * sizeof(float), hipMemcpyDeviceToHost));
  if (out->labels) ODT_HIP(d2h_on(st, out->labels, h->det.out_labels, (size_t)B * per * sizeof(int), hipMemcpyDeviceToHost));
  if (out->feats && total > 0)
    ODT_HIP(d2h_on(st, out->feats, h->final_feat, (size_t)total * FC * 49 * sizeof(float), hipMemcpyDeviceToHost));
  if (out->pooled && total > 0)
    ODT_HIP(d2h_on(st, out->pooled, h->final_pooled, (size_t)total * FC * sizeof(float), hipMemcpyDeviceToHost));
  if (out->masks) {
    ODT_CHECK(h->final_masks != nullptr, "odt_forward: masks requested but the model was built without add_mask");
    ODT_HIP(d2h_on(st, out->masks, h->final_masks, (size_t)B * per * 784 * sizeof(float), hipMemcpyDeviceToHost));
  }
  return 0;
}

static int slot_prepare(odt_handle h, odt_model::Slot& sl, size_t in_bytes) {
  const odt_config& cfg = h->cfg;
  const size_t B = cfg.batch, per = cfg.result_per_im, FC = cfg.fpn_channels;
  if (!h->copy_in && create_side_stream(&h->copy_in)) return 1;
  if (!h->copy_out && create_side_stream(&h->copy_out)) return 1;
  if (sl.pin_in_bytes < in_bytes) {
    if (sl.pin_in) ODT_HIP(hipHostFree(sl.pin_in));
    if (sl.dev_in) ODT_HIP(hipFree(sl.dev_in));
    ODT_HIP(hipHostMalloc(&sl.pin_in, in_bytes, 0));
    ODT_HIP(hipMalloc(&sl.dev_in, in_bytes));
    sl.pin_in_bytes = sl.dev_in_bytes = in_bytes;
  }
  if (!sl.pin_boxes) {
    ODT_HIP(hipHostMalloc((void**)&sl.pin_boxes, B * per * 4 * sizeof(float), 0));
    ODT_HIP(hipHostMalloc((void**)&sl.pin_probs, B * per * sizeof(float), 0));
    ODT_HIP(hipHostMalloc((void**)&sl.pin_labels, B * per * sizeof(int), 0));
    ODT_HIP(hipHostMalloc((void**)&sl.pin_valid, B * sizeof(int), 0));
    ODT_HIP(hipHostMalloc((void**)&sl.pin_feats, B * per * FC * 49 * sizeof(float), 0));
    ODT_HIP(hipHostMalloc((void**)&sl.pin_pooled, B * per * FC * sizeof(float), 0));
    if (h->final_masks) ODT_HIP(hipHostMalloc((void**)&sl.pin_masks, B * per * 784 * sizeof(float), 0));
    ODT_HIP(hipEventCreate(&sl.h2d_done));
    ODT_HIP(hipEventCreate(&sl.fwd_done));
    ODT_HIP(hipEventCreate(&sl.d2h_done));
  }
  return 0;
}

int odt_set_source_size(odt_handle h, int src_height, int src_width) {
  ODT_CHECK(h, "odt_set_source_size: null handle");
  ODT_CHECK(src_height > 0 && src_width > 0 && src_height < 32768 && src_width < 32768,
            "odt_set_source_size: bad size");
  ODT_CHECK(h->slot[0].ticket < 0 && h->slot[1].ticket < 0, "odt_set_source_size: tickets in flight");
  ODT_HIP(hipSetDevice(h->device));
  const size_t need = (size_t)h->cfg.batch * src_height * src_width * 3 * sizeof(float);
  if (need > h->frames_bytes && need > h->frames_src.bytes) {
    ODT_HIP(hipStreamSynchronize(h->own_stream));
    h->frames_src.release();
    if (h->frames_src.alloc(need)) return 1;
  }
  h->src_h = src_height; h->src_w = src_width;
  if (h->cfg.graph == ODT_GRAPH_EFFNET) {
    // dataloader.py:100-112: image_scale = min(out_w / w, out_h / h) in float32, scaled size by truncation;
    // image_scale_to_original = 1 / image_scale multiplies the output boxes (efficientdet_wrapper.py:57)
    const float sy = (float)h->cfg.height / (float)src_height, sx = (float)h->cfg.width / (float)src_width;
    const float sc = sx < sy ? sx : sy;
    h->eff_scaled_h = (int)((float)src_height * sc); h->eff_scaled_w = (int)((float)src_width * sc);
    ODT_CHECK(h->eff_scaled_h >= 1 && h->eff_scaled_w >= 1 && h->eff_scaled_h <= h->cfg.height &&
              h->eff_scaled_w <= h->cfg.width, "odt_set_source_size: scaled frame does not fit the network input");
    h->eff_post.image_scale = 1.0f / sc;
  }
  return 0;
}

int odt_ingest_buffer(odt_handle h, int dtype, void** buffer, size_t* bytes) {
  ODT_CHECK(h && buffer && bytes, "odt_ingest_buffer: null argument");
  ODT_CHECK(dtype == ODT_DTYPE_U8 || dtype == ODT_DTYPE_F32, "odt_ingest_buffer: bad dtype");
  ODT_HIP(hipSetDevice(h->device));
  const size_t n = input_bytes(h, dtype);
  odt_model::Slot& sl = h->slot[h->next_ticket & 1];
  ODT_CHECK(sl.ticket < 0, "odt_ingest_buffer: slot still in flight (collect its ticket first)");
  if (slot_prepare(h, sl, n)) return 1;
  *buffer = sl.pin_in; *bytes = n;
  sl.ingest_armed = h->next_ticket; sl.ingest_dtype = dtype;     // odt_submit_ex(frames = NULL) takes exactly this buffer
  return 0;
}

int odt_submit(odt_handle h, const void* frames, int dtype, int* ticket) {
  return odt_submit_ex(h, frames, dtype, ODT_WANT_ALL, ticket);
}

int odt_submit_ex(odt_handle h, const void* frames, int dtype, int want, int* ticket) {
  ODT_CHECK(h && ticket, "odt_submit: null argument");
  ODT_CHECK(h->finalized, "odt_submit: call odt_finalize_weights first");
  ODT_CHECK(h->cfg.graph != ODT_GRAPH_EFFNET, "odt_submit: not available for the EfficientNet backbone graph");
  ODT_CHECK(dtype == ODT_DTYPE_U8 || dtype == ODT_DTYPE_F32, "odt_submit: bad dtype");
  ODT_CHECK((want & ~ODT_WANT_ALL) == 0, "odt_submit_ex: unknown ODT_WANT_* bits");
  ODT_HIP(hipSetDevice(h->device));
  const odt_config& cfg = h->cfg;
  const size_t B = cfg.batch, per = cfg.result_per_im, FC = cfg.fpn_channels;
  const size_t n = input_bytes(h, dtype);
  const int t = h->next_ticket;
  odt_model::Slot& sl = h->slot[t & 1];
  odt_model::Slot& prev = h->slot[(t & 1) ^ 1];
  ODT_CHECK(sl.ticket < 0, "odt_submit: two tickets already outstanding (collect one first)");
  // frames == NULL: the caller filled odt_ingest_buffer()'s memory -- which must have been handed out for THIS ticket and
  // dtype (otherwise the slot's pinned buffer holds stale or no frames, or slot_prepare below would even reallocate it)
  ODT_CHECK(frames != nullptr || (sl.ingest_armed == t && sl.ingest_dtype == dtype && sl.pin_in != nullptr && sl.pin_in_bytes >= n),
            "odt_submit: frames == NULL needs odt_ingest_buffer() for this ticket and dtype first");
  sl.ingest_armed = -1;
  if (slot_prepare(h, sl, n)) return 1;
  // (round 6 measured the staging frame by frame -- eight memcpy + H2D pairs instead of one: the first frame is on PCIe
  // earlier, and the eight copies get in the way of the tracker's small H2D / D2H copies on the copy engines: cosine calls
  // 150 -> 500 us next to them, detect + track 284 -> 245 FPS, pipelined 311 -> 302: profiles/r06_frame_staging_ab.txt.  One copy.)
  if (frames != nullptr) staging_copy(sl.pin_in, frames, n);
  ODT_HIP(hipMemcpyAsync(sl.dev_in, sl.pin_in, n, hipMemcpyHostToDevice, h->copy_in));
  ODT_HIP(hipEventRecord(sl.h2d_done, h->copy_in));
  hipStream_t st = h->own_stream;
  ODT_HIP(hipStreamWaitEvent(st, sl.h2d_done, 0));
  sl.want = want;
  if (!(want & ODT_WANT_FEATS)) {
    // nothing large goes back: the outputs are copied right behind the forward on the compute stream (stream order
    // keeps the next forward's tail off the single device output buffers), no event wait inside the plan.  A previous
    // ticket that used the copy stream for its [M,C,7,7] features still has to be waited for.
    if (prev.ticket >= 0 && (prev.want & ODT_WANT_FEATS)) ODT_HIP(hipStreamWaitEvent(st, prev.d2h_done, 0));
    h->wait_before_detect = nullptr;
    h->d2h_slot = &sl; h->d2h_want = want;
    const int rc = run_plan(h, sl.dev_in, dtype, 1, st);
    h->d2h_slot = nullptr; h->d2h_want = 0;
    if (rc) return 1;
    ODT_HIP(hipEventRecord(sl.d2h_done, h->done_stream));
    sl.ticket = t;
    *ticket = t;
    h->next_ticket = t + 1;
    return 0;
  }
  // the previous ticket's D2H reads the (single) device output buffers: the tail of this forward
  // must not overwrite them before that copy is done
  h->wait_before_detect = (prev.ticket >= 0 && (prev.want & ODT_WANT_FEATS)) ? prev.d2h_done : nullptr;
  if (run_plan(h, sl.dev_in, dtype, 1, st)) return 1;
  ODT_HIP(hipEventRecord(sl.fwd_done, h->done_stream));
  hipStream_t co = h->copy_out;
  ODT_HIP(hipStreamWaitEvent(co, sl.fwd_done, 0));
  ODT_HIP(hipMemcpyAsync(sl.pin_valid, h->det.out_valid, B * sizeof(int), hipMemcpyDeviceToHost, co));
  ODT_HIP(hipMemcpyAsync(sl.pin_boxes, h->det.out_boxes, B * per * 4 * sizeof(float), hipMemcpyDeviceToHost, co));
  ODT_HIP(hipMemcpyAsync(sl.pin_probs, h->det.out_probs, B * per * sizeof(float), hipMemcpyDeviceToHost, co));
  ODT_HIP(hipMemcpyAsync(sl.pin_labels, h->det.out_labels, B * per * sizeof(int), hipMemcpyDeviceToHost, co));
  ODT_HIP(hipMemcpyAsync(sl.pin_feats, h->final_feat, B * per * FC * 49 * sizeof(float), hipMemcpyDeviceToHost, co));
  if (want & ODT_WANT_POOLED)
    ODT_HIP(hipMemcpyAsync(sl.pin_pooled, h->final_pooled, B * per * FC * sizeof(float), hipMemcpyDeviceToHost, co));
  if ((want & ODT_WANT_MASKS) && h->final_masks)
    ODT_HIP(hipMemcpyAsync(sl.pin_masks, h->final_masks, B * per * 784 * sizeof(float), hipMemcpyDeviceToHost, co));
  ODT_HIP(hipEventRecord(sl.d2h_done, co));
  sl.ticket = t;
  *ticket = t;
  h->next_ticket = t + 1;
  return 0;
}

int odt_collect(odt_handle h, int ticket, odt_outputs* out) {
  ODT_CHECK(h && out, "odt_collect: null argument");
  odt_model::Slot& sl = h->slot[ticket & 1];
  ODT_CHECK(ticket >= 0 && sl.ticket == ticket, "odt_collect: unknown or already collected ticket");
  ODT_HIP(hipSetDevice(h->device));
  ODT_HIP(hipEventSynchronize(sl.d2h_done));
  const odt_config& cfg = h->cfg;
  const size_t B = cfg.batch, per = cfg.result_per_im, FC = cfg.fpn_channels;
  size_t total = 0;
  for (size_t b = 0; b < B; ++b) total += (size_t)sl.pin_valid[b];
  if (out->valid) std::memcpy(out->valid, sl.pin_valid, B * sizeof(int));
  if (out->boxes) std::memcpy(out->boxes, sl.pin_boxes, B * per * 4 * sizeof(float));
  if (out->probs) std::memcpy(out->probs, sl.pin_probs, B * per * sizeof(float));
  if (out->labels) std::memcpy(out->labels, sl.pin_labels, B * per * sizeof(int));
  ODT_CHECK(!out->feats || (sl.want & ODT_WANT_FEATS), "odt_collect: feats requested but the ticket was submitted without ODT_WANT_FEATS");
  ODT_CHECK(!out->pooled || (sl.want & ODT_WANT_POOLED), "odt_collect: pooled requested but the ticket was submitted without ODT_WANT_POOLED");
  ODT_CHECK(!out->masks || (sl.want & ODT_WANT_MASKS), "odt_collect: masks requested but the ticket was submitted without ODT_WANT_MASKS");
  if (out->feats) std::memcpy(out->feats, sl.pin_feats, total * FC * 49 * sizeof(float));
  if (out->pooled) std::memcpy(out->pooled, sl.pin_pooled, total * FC * sizeof(float));
  if (out->masks) {
    ODT_CHECK(sl.pin_masks != nullptr, "odt_collect: masks requested but the model was built without add_mask");
    std::memcpy(out->masks, sl.pin_masks, (size_t)h->cfg.batch * h->cfg.result_per_im * 784 * sizeof(float));
  }
  sl.ticket = -1;
  return 0;
}

int odt_tap(odt_handle h, const char* name, float* dst, size_t cap_elems, int64_t* shape_out, int* rank_out) {
  ODT_CHECK(h && name && shape_out && rank_out, "odt_tap: null argument");
  ODT_HIP(hipSetDevice(h->device));
  if (std::string(name) == "nproposals") {
    shape_out[0] = h->cfg.batch; *rank_out = 1;
    if (dst) {
      ODT_CHECK(cap_elems >= (size_t)h->cfg.batch, "odt_tap: buffer too small");
      std::vector<int> v(h->cfg.batch);
      ODT_HIP(hipDeviceSynchronize());
      ODT_HIP(hipMemcpy(v.data(), h->prop.nprops, v.size() * sizeof(int), hipMemcpyDeviceToHost));
      for (size_t i = 0; i < v.size(); ++i) dst[i] = (float)v[i];
    }
    return 0;
  }
  auto it = h->taps.find(name);
  ODT_CHECK(it != h->taps.end(), std::string("odt_tap: unknown stage ") + name);
  ODT_CHECK(h->transient_taps.count(name) == 0, std::string("odt_tap: stage tensor ") + name + " is not kept after a forward "
            "(its memory is reused inside the activation arena): create the handle with odt_config.keep_taps = 1");
  const Tensor& t = it->second;
  shape_out[0] = t.B; shape_out[1] = t.H; shape_out[2] = t.W; shape_out[3] = t.C; *rank_out = 4;
  if (dst) {
    ODT_CHECK(cap_elems >= t.elems(), "odt_tap: buffer too small");
    ODT_HIP(hipDeviceSynchronize());
    ODT_HIP(hipMemcpy(dst, t.d, t.elems() * sizeof(float), hipMemcpyDeviceToHost));
  }
  return 0;
}

int odt_profile_enable(odt_handle h, int enable) {
  ODT_CHECK(h != nullptr, "null handle");
  h->profile = enable != 0;
  h->prof_conv_ms = h->prof_conv_flops = h->prof_total_ms = 0; h->prof_launches = 0;
  h->prof_layer_ms.assign(h->convs.size(), 0.0);
  return 0;
}

int odt_profile_layer(odt_handle h, int index, char* name, int name_cap, double* flops, double* ms,
                      int64_t* mnk, int* count) {
  ODT_CHECK(h != nullptr, "null handle");
  if (count) *count = (int)h->convs.size();
  if (index < 0 || index >= (int)h->convs.size()) return 0;
  const ConvOp& c = h->convs[index];
  if (name && name_cap > 0) {    // layers on the bf16x3 split kernel are tagged (bench.py / profile_layers.py group by it)
    const bool fused = index < (int)h->conv_fused.size() && h->conv_fused[index];
    const std::string nm = c.name + (fused ? "[fused into the producer's epilogue]" : (c.p.head_wt != nullptr ? "+head" : (c.p.f_wt != nullptr ? "+conv3" : (c.p.stem_pool ? "+pool0" : "")))) +
                           (!fused && c.p.wt_split != nullptr ? (c.p.wt_split_kind == 2 ? "[fp16x2]" : "[bf16x3]") : "");
    std::strncpy(name, nm.c_str(), name_cap - 1); name[name_cap - 1] = 0;
  }
  if (flops) *flops = (index < (int)h->conv_fused.size() && h->conv_fused[index]) ? 0.0 : conv_flops(c.p);
  if (ms) *ms = index < (int)h->prof_layer_ms.size() ? h->prof_layer_ms[index] : 0.0;
  if (mnk) { mnk[0] = (int64_t)c.p.B * c.p.Ho * c.p.Wo; mnk[1] = c.p.Cout; mnk[2] = (int64_t)c.p.kh * c.p.kw * c.p.Cin; }
  return 0;
}

int odt_describe(odt_handle h, char* buf, int cap) {
  ODT_CHECK(h != nullptr && buf != nullptr && cap > 0, "odt_describe: null argument");
  int fam[4] = {0, 0, 0, 0}, nsk = 0, nfused = 0;
  for (size_t i = 0; i < h->convs.size(); ++i) {
    const ConvOp& c = h->convs[i];
    if (i < h->conv_fused.size() && h->conv_fused[i]) { ++nfused; continue; }
    fam[c.p.wt_split != nullptr ? c.p.wt_split_kind : 0] += 1;
    if (c.p.wt_split != nullptr && c.p.splitk > 1) ++nsk;
  }
  size_t dev_bytes = 0;
  for (const auto& b : h->bufs) dev_bytes += b->bytes;
  dev_bytes += h->frames_src.bytes;
  for (const auto& sl : h->slot) dev_bytes += sl.dev_in_bytes;
  // every ODT_* override that was set when the plan was built, by name (knobs.hpp: nothing else reads the environment)
  std::string envs;
  for (size_t i = 0; i < h->env_active.size(); ++i) {
    std::string e = h->env_active[i];
    for (char& ch : e) if (ch == '"' || ch == '\\' || (unsigned char)ch < 0x20) ch = '?';
    envs += (i ? ", \"" : "\"") + e + "\"";
  }
  char tmp[8192];
  std::snprintf(tmp, sizeof(tmp),
                "{\"conv_arith\": \"%s\", \"conv_launches\": %d, \"convs_fused_into_epilogues\": %d, \"exact_f32_mfma_launches\": %d, "
                "\"bf16x3_split_launches\": %d, \"fp16x2_split_launches\": %d, \"bottleneck_tails_fused\": %d, \"stem_fused\": %d, \"mbconv_expand_dw_fused\": %d, \"split_launches_by_family\": {\"split3_8wave_lds_dma\": %d, "
                "\"one_stage_bk32\": %d, \"h2_8wave_lds_dma\": %d, \"of_split3_with_split_k\": %d}, \"policy\": {\"family\": %d, \"min_tiles\": %ld, "
                "\"min_tiles3\": %ld, \"min_k\": %d}, \"env_overrides_applied\": %d, \"env_overrides\": [%s], "
                "\"memory\": {\"device_bytes\": %zu, \"activation_arena_bytes\": [%zu, %zu], \"arena_tensors\": %zu, "
                "\"arena_tensor_bytes_unshared\": %zu, \"dedicated_tensor_bytes\": %zu, \"keep_taps\": %d}, \"convs_cut_into_batch_ranges\": %d}",
                h->policy.arith != 0 && fam[2] > 0 ? "f32 through fp16x2 / bf16x3 split products"
                    : (h->policy.arith != 0 && fam[1] + fam[3] > 0 ? "f32 through bf16x3 split products" : "exact f32 MFMA"),
                (int)h->convs.size() - nfused, nfused, fam[0], fam[1] + fam[3], fam[2], h->convs_h2f, h->stem_fused, h->mb_fused, fam[3], fam[1], fam[2], nsk, h->policy.family,
                h->policy.min_tiles, h->policy.min_tiles3, h->policy.min_k, (int)h->env_active.size(), envs.c_str(),
                dev_bytes, h->arena_bytes[0], h->arena_bytes[1], h->vt.size(), h->virtual_tensor_bytes,
                h->dedicated_tensor_bytes, h->cfg.keep_taps, h->chunked_convs);
  ODT_CHECK((int)std::strlen(tmp) < cap, "odt_describe: buffer too small");
  std::strncpy(buf, tmp, cap - 1); buf[cap - 1] = 0;
  return 0;
}

int odt_range_health(odt_handle h, int rebase, double* worst_growth, char* tensor, int tensor_cap, double* tensor_amax, long long* tensors_seen) {
  ODT_CHECK(h != nullptr && worst_growth != nullptr, "odt_range_health: null argument");
  double worst = 0.0, wamax = 0.0;
  int wslot = -1;
  long long seen = 0;
  if (h->range_host != nullptr && !h->range_baseline.empty()) {
    const volatile unsigned* r = h->range_host;
    for (int g = 0; g < 2; ++g)
      for (int i = g * odt_model::kAmaxSlots; i < g * odt_model::kAmaxSlots + h->amax_used[g]; ++i) {
        const unsigned bits = r[i];
        float a; std::memcpy(&a, &bits, 4);
        if (!(a > 0.f) || !std::isfinite(a)) continue;
        ++seen;
        float& base = h->range_baseline[i];
        if (!(base > 0.f)) { base = a; continue; }          // first sighting: the level the next calls compare with
        const double f = (double)a / (double)base;
        if (f > worst) { worst = f; wslot = i; wamax = a; }
        if (rebase) base = a;
      }
  }
  *worst_growth = worst;
  if (tensor_amax) *tensor_amax = wamax;
  if (tensors_seen) *tensors_seen = seen;
  if (tensor != nullptr && tensor_cap > 0) {
    const std::string nm = wslot >= 0 && wslot < (int)h->range_slot_name.size() ? h->range_slot_name[wslot] : std::string();
    std::strncpy(tensor, nm.c_str(), tensor_cap - 1); tensor[tensor_cap - 1] = 0;
  }
  return 0;
}

int odt_profile_read(odt_handle h, double* conv_ms, double* conv_flops, int* conv_launches, double* total_ms) {
  ODT_CHECK(h != nullptr, "null handle");
  if (conv_ms) *conv_ms = h->prof_conv_ms;
  if (conv_flops) *conv_flops = h->prof_conv_flops;
  if (conv_launches) *conv_launches = h->prof_launches;
  if (total_ms) *total_ms = h->prof_total_ms;
  return 0;
}

}  // extern "C"
