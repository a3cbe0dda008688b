/*
 * odt.h -- C ABI of libodt_hip.so, the MI355X (gfx950) implementation of the
 * per-frame detection + appearance-feature + cosine-matching hot path of
 * JunweiLiang/Object_Detection_Tracking.
 *
 * The reference has no native code and no FFI: the boundary it crosses once
 * per frame is `sess.run([final_boxes, final_labels, final_probs,
 * fpn_box_feat], feed_dict)` (reference obj_detect_tracking.py:610-635,
 * obj_detect_tracking_multi.py:460-466) plus the numpy call
 * `metric.distance(features, targets)` inside the tracker (reference
 * deep_sort/tracker.py:94-99 -> deep_sort/nn_matching.py:156-177).  Each entry
 * point below names the reference interface it replaces.  Plain pointers and
 * sizes only; every function returns 0 on success, non-zero on error, and
 * never aborts the process (odt_last_error() gives the message).
 *
 * Layouts: frames are HWC BGR (uint8 or float32 0..255) exactly as the
 * reference feeds them; outputs use the reference's layouts (boxes x1,y1,x2,y2
 * in resized-image coordinates, features NCHW [R,256,7,7]).
 *
 * Arithmetic: all tensors and accumulations are float32.  By default
 * (odt_config.conv_arith = ODT_ARITH_DEFAULT, conv_split_family = 0 / 2) the large
 * convolutions evaluate every f32 product as THREE exact f16 x f16 matrix-core
 * products (v_mfma_f32_32x32x16_f16) of a 2-way f16 split of both operands
 * ("fp16x2": hi*hi + hi*lo + lo*hi, 22 significand bits per operand, lo*lo <=
 * 2^-22 |a||b| dropped), each operand carrying an exact power-of-two scale -- one
 * per weight row, one per activation tensor from the |max| its producer recorded
 * (per pixel row inside the fused bottleneck tails).  Error at the level of an f32
 * dot product in another summation order PROVIDED a tensor's useful content lies
 * within ~2^17 of its |max| (below that the low piece is an f16 subnormal: absolute
 * instead of relative precision).  The library does not check that assumption by
 * itself: the Python host guards it by default (models._Engine,
 * conv_split_family = "auto": the first forward also runs on a conv_split_family = 3
 * twin handle and the engine keeps whichever handle the comparison allows); a C
 * caller does the same with two handles, or selects a mode without a range
 * assumption: conv_split_family = 3 (six exact bf16 x bf16 products of a 3-way
 * bf16 split: all 24 bits) or conv_arith = ODT_ARITH_F32 (the exact-f32 matrix
 * instruction in every layer).  Layers the fp16x2 kernels do not take (no recorded
 * input range, tiles that do not fit) run on bf16x3 / exact f32; odt_describe says
 * which launches run on what.
 *
 * Errors: odt_last_error() is per calling thread (errno-style storage): a handle
 * is driven by one thread at a time, so the message is that of the calling
 * thread's last failing call -- never another handle's on another thread.
 */
#ifndef ODT_H_
#define ODT_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct odt_model* odt_handle;

#define ODT_DTYPE_U8 0
#define ODT_DTYPE_F32 1

/* graph semantics: the reference ships two different inference graphs */
#define ODT_GRAPH_SINGLE 0 /* Mask_RCNN_FPN        (models.py:488-973)   b = 1 */
#define ODT_GRAPH_MULTI 1  /* Mask_RCNN_FPN_multi  (models.py:2058-2408) b = B */
#define ODT_GRAPH_EFFNET 2 /* EfficientDet (efficientdet_wrapper.py:12-106): EfficientNet backbone + BiFPN + class / box nets + top-k / NMS */

/* Mirrors the fields of the reference's `args`/config namespace that the
 * inference graph reads (obj_detect_tracking.py:303-387). */
typedef struct odt_config {
  int32_t graph;            /* ODT_GRAPH_SINGLE / ODT_GRAPH_MULTI             */
  int32_t batch;            /* im_batch_size                                  */
  int32_t height, width;    /* resized frame size fed to forward              */
  int32_t num_class;        /* incl. background (15)                          */
  int32_t num_blocks[4];    /* resnet_num_block (3,4,23,3)                    */
  int32_t use_dilations;    /* model version >= 3                             */
  int32_t fpn_channels;     /* fpn_num_channel (256)                          */
  int32_t head_dim;         /* fpn_frcnn_fc_head_dim (1024)                   */
  int32_t rpn_topk;         /* rpn_test_post_nms_topk (<= 4096; script default 1000) */
  int32_t result_per_im;    /* 100                                            */
  int32_t anchor_field;     /* ceil(max_size/stride0): side of level-0 anchor grid */
  float rpn_nms_thresh;     /* rpn_proposal_nms_thres (0.7)                   */
  float rpn_decode_clip;    /* bbox_decode_clip = log(max_size/16)            */
  float head_decode_clip;   /* log(1333/16) (nn.py:1518 default)              */
  float bbox_reg_weights[4];/* fastrcnn_bbox_reg_weights (10,10,5,5)          */
  float result_score_thresh;/* result_score_thres (1e-4)                      */
  float head_nms_thresh;    /* fastrcnn_nms_iou_thres (0.5)                   */
  int32_t add_mask;         /* --add_mask: Mask R-CNN head on the final boxes (models.py:932-962); single-image graph */
  int32_t mask_dim;         /* mrcnn_head_dim (256)                           */
  int32_t eff_backbone;     /* ODT_GRAPH_EFFNET: 0..7 = efficientnet-b0..b7 (efficientdet_wrapper.py:511-587) */
  int32_t eff_det;          /* -1: backbone only; 0..7: efficientdet-d0..d7 feature network + class / box nets */
  int32_t eff_topk;         /* efficientdet_max_detection_topk (5000)         */
  float eff_image_scale;    /* image_scale_to_original applied to the output boxes (wrapper :57) */
  int32_t conv_arith;       /* ODT_ARITH_*: how the conv / FC products are evaluated (fixed per handle, see odt_describe) */
  int32_t conv_split_family;/* 0 library default (2); 1: one-stage bf16x3 kernel only; 3: + the 8-wave bf16x3 kernels where they fit;
                               2: + the fp16x2 kernels (three f16 products per MAC) for layers with 256-row tiles whose input range
                               the producing kernel recorded (A/B runs) */
  int32_t keep_taps;        /* 0 (production): activations live in a liveness-planned arena -- a stage tensor's memory is
                             * reused as soon as its last consumer has run, and odt_tap can only read the tensors that
                             * outlive the forward (outputs, proposals, zero-bordered buffers); 1 (debug / parity runs):
                             * every stage tensor keeps a dedicated buffer and odt_tap can read all of them after a
                             * forward (b=8 @1080p: ~25 GB instead of ~5 GB of activations) */
  int32_t tail_overlap;     /* 0 (default): the selection / box-head / NMS tail of forward i runs on a side stream under the backbone
                             * of forward i + 1 (one handle, forwards back to back: +1 %); -1: the whole forward stays on the compute
                             * stream -- for handles that run consecutive frames side by side (models.predict_stream: with three b = 1
                             * frames in flight 189 FPS with the side streams, 235 without: they compete for hardware queues) */
} odt_config;

/* conv_arith: all modes keep f32 tensors and f32 accumulation.  ODT_ARITH_F32: every product on the exact-f32 MFMA
 * (v_mfma_f32_32x32x2_f32); ODT_ARITH_BF16X3 ("split"): layers large enough to fill the chip evaluate each f32 product
 * through exact low-precision MFMA products of split operands -- six bf16 x bf16 products of a 3-way bf16 split (csrc/
 * conv_split3.hip), or, conv_split_family = 2, three f16 x f16 products of a 2-way f16 split of operands scaled by the
 * tensor's recorded maximum (csrc/conv_h2.hip) -- with the error of an f32 dot product either way (bounds in csrc/
 * conv_split_common.hpp, asserted in tests/test_ops.py); ODT_ARITH_DEFAULT = the split mode.  The ODT_CONV_* environment
 * variables are debug overrides, read once when the handle is created and reported by odt_describe. */
#define ODT_ARITH_DEFAULT 0
#define ODT_ARITH_F32 1
#define ODT_ARITH_BF16X3 2

/* Caller-owned host output buffers (capacities in elements of the row type).
 * Replaces the numpy arrays sess.run returns (models.py:965-973 / :2311-2320).
 *   boxes  [batch, result_per_im, 4] float32
 *   probs  [batch, result_per_im]    float32
 *   labels [batch, result_per_im]    int32 (1-based class id)
 *   valid  [batch]                   int32 number of detections per frame
 *   feats  [sum(valid), C, 7, 7]     float32 (may be NULL)   fpn_box_feat
 *   pooled [sum(valid), C]           float32 (may be NULL)   7x7 mean of feats
 *                                    (deep_sort/utils.py:27-28 done on device)
 *   masks  [result_per_im, 28, 28]   float32 (may be NULL; add_mask only) final_masks: sigmoid of the
 *                                    mask logits of each detection's own class; rows >= valid[0] are 0
 */
typedef struct odt_outputs {
  float* boxes;
  float* probs;
  int32_t* labels;
  int32_t* valid;
  float* feats;
  float* pooled;
  float* masks;
} odt_outputs;

const char* odt_last_error(void);
int odt_device_count(int* count);

/* get_model(config, gpuid) (models.py:97-119): build the static execution
 * plan, allocate weights + workspace on `device`. */
int odt_create(const odt_config* cfg, int device, odt_handle* out);
int odt_destroy(odt_handle h);

/* initialize(config, sess) .npz branch (obj_detect_tracking.py:417-435): load
 * one variable by its reference name ("conv0/W", "group2/block5/conv2/bn/gamma",
 * "fastrcnn/fc6/W", ...) in the reference layout (conv HWIO, dense [in,out]).
 * "anchors/lvl<i>" [S,S,A,4] carries the host-precomputed anchor field. */
int odt_load_tensor(odt_handle h, const char* name, const float* data,
                    const int64_t* shape, int rank);
/* fold BN into conv weights, re-layout to [Cout][kh][kw][Cin], upload. */
int odt_finalize_weights(odt_handle h);

/* sess.run([final_boxes, final_labels, final_probs, fpn_box_feat], feed)
 * (obj_detect_tracking.py:632-635).  frames: [batch,H,W,3] BGR; on_device != 0
 * means `frames` is a device pointer (HBM-resident input).  stream: hipStream_t
 * to run on (NULL = the handle's own stream).  Blocks until outputs are on the
 * host. */
int odt_forward(odt_handle h, const void* frames, int dtype, int on_device,
                void* stream, odt_outputs* out);
/* Same, but only enqueues the device work (no D2H, no sync): used to time the
 * device path with inputs resident in HBM. */
int odt_forward_async(odt_handle h, const void* frames, int dtype,
                      int on_device, void* stream);
int odt_synchronize(odt_handle h);
/* The outputs of the most recently enqueued forward (odt_forward_async, or an odt_forward whose host copies are no
 * longer at hand): waits for it and copies the device result buffers to `out` exactly as odt_forward does.  bench.py
 * uses it to check the replayed, device-resident forwards of its timed region against a blocking odt_forward. */
int odt_read_outputs(odt_handle h, odt_outputs* out);

/* Source frames of a different size than the plan's input (reference obj_detect_tracking.py:597-608:
 * frame.astype("float32") -> resizeImage(frame, short_edge_size, max_size) = cv2.resize(...,
 * INTER_LINEAR), nn.py:1540-1560, on the host for every frame).  After this call odt_forward /
 * odt_forward_async / odt_submit take frames of [batch, src_height, src_width, 3] and the bilinear
 * resize to [height, width] runs on the device, fused with the normalisation (so a 720p or 4K
 * stream crosses PCIe at its native uint8 size).  Boxes come back in resized-image coordinates as
 * in the reference.  Pass the plan's own size to switch back.  No tickets may be in flight. */
int odt_set_source_size(odt_handle h, int src_height, int src_width);

/* Pipelined ingest (replaces the frame.astype(float32) + feed_dict copy of
 * obj_detect_tracking.py:597-635 and the prefetch queue of enqueuer_thread.py:236-303 on the
 * device side): two slots of pinned host staging + device input + pinned output staging.
 * odt_submit copies `frames` (host, [batch,H,W,3], u8 or f32) into the slot's pinned buffer,
 * enqueues H2D on a copy stream, the forward on the compute stream and the D2H of all outputs
 * on a second copy stream, and returns immediately with a ticket; odt_collect waits for that
 * ticket and fills `out`.  At most two tickets may be outstanding; with two in flight the H2D
 * of batch i+1 and the D2H of batch i-1 overlap the forward of batch i.
 * odt_ingest_buffer exposes the next slot's pinned input buffer so a decoder can write into
 * it directly (pass frames == NULL to odt_submit to use what was written there).
 * odt_submit_ex: the same with a choice of what crosses PCIe on the way back (ODT_WANT_* bits).  The
 * only consumer of fpn_box_feat on this path averages it to [M,C] (deep_sort/utils.py:27-28), so the
 * tracking loop asks for ODT_WANT_POOLED only: 0.8 MB per 8-frame batch instead of the 40 MB of
 * [M,256,7,7]; with nothing large to copy the D2H is enqueued right behind the forward on the stream its tail runs on
 * (the handle's side stream when the tail of forward i overlaps the trunk of forward i+1 -- the default -- or the compute
 * stream with ODT_TAIL_OVERLAP=0); tickets that do ask for the features use the copy stream.
 * odt_submit == odt_submit_ex(..., ODT_WANT_ALL, ...). */
#define ODT_WANT_FEATS  1   /* fpn_box_feat [M,C,7,7] */
#define ODT_WANT_POOLED 2   /* its 7x7 mean [M,C] */
#define ODT_WANT_MASKS  4   /* final_masks (add_mask models) */
#define ODT_WANT_ALL    7
int odt_submit(odt_handle h, const void* frames, int dtype, int* ticket);
int odt_submit_ex(odt_handle h, const void* frames, int dtype, int want, int* ticket);
int odt_collect(odt_handle h, int ticket, odt_outputs* out);
int odt_ingest_buffer(odt_handle h, int dtype, void** buffer, size_t* bytes);

/* What this handle runs, as a JSON object: conv arithmetic mode, launches per kernel family, policy thresholds,
 * number of ODT_CONV_* environment overrides that were applied at creation. */
int odt_describe(odt_handle h, char* buf, int cap);

/* Debug / parity taps: copy a named stage tensor (device layout: NHWC) to the
 * host (handles created with odt_config.keep_taps = 1; without it only the tensors that outlive a forward are
 * readable and the others return an error).  shape_out receives up to 4 dims.  Names: "image_pad", "conv0",
 * "pool0", "c2".."c5", "p2".."p6", "rpn2".."rpn6" (16 ch: 3 logits + 12
 * deltas + 1 pad), "proposals", "nproposals", "roi_feat", "fc7", "head_out",
 * "decoded_boxes", "label_probs". */
int odt_tap(odt_handle h, const char* name, float* dst, size_t cap_elems,
            int64_t* shape_out, int* rank_out);

/* Continuous range watch of the fp16x2 kernels (round 6).  Those kernels scale a conv's activations by ONE power of two per
 * source tensor, taken from the |max| its producer recorded; a tensor whose maximum moves far above its useful content (an
 * outlier channel that a scene cut or an exposure change switches on) leaves f32 level.  Every forward records those maxima
 * anyway; at the start of the next forward they go to host-visible memory.  This call compares them (no copy, no
 * synchronisation; the numbers are one or two forwards old) with the level the caller last accepted: *worst_growth = the
 * largest ratio |max| now / |max| accepted over the plan's tensors, with the producing layer's name and its |max|.  A tensor
 * seen for the first time sets its own level; rebase != 0 accepts the current maxima as the new level (after reporting).
 * The reference has no counterpart (TensorFlow computes in f32 throughout); the host side (models._Engine, the "auto"
 * conv_split_family) re-arms its fp16x2-vs-bf16x3 comparison when a maximum has grown past its watch ratio
 * (obj_detect_tracking.py:577-635 runs one model over a whole video).  A handle without fp16x2 launches reports 0. */
int odt_range_health(odt_handle h, int rebase, double* worst_growth, char* tensor, int tensor_cap, double* tensor_amax,
                     long long* tensors_seen);

/* Per-launch timing of the implicit-GEMM conv kernel family, measured with
 * HIP events on the launch stream (for bench.py's roofline object). */
int odt_profile_enable(odt_handle h, int enable);
int odt_profile_read(odt_handle h, double* conv_ms, double* conv_flops,
                     int* conv_launches, double* total_ms);
/* Per-launch view of the same measurement: conv launch `index` of the plan (0 <=
 * index < *count): layer name, algorithmic FLOPs, HIP-event milliseconds summed
 * since odt_profile_enable, GEMM view M, N, K. */
int odt_profile_layer(odt_handle h, int index, char* name, int name_cap,
                      double* flops, double* ms, int64_t* mnk, int* count);

/* Measurement support (bench.py's `roofline.sustained_peak`; no model code calls it): what the bf16 matrix pipe of this
 * device sustains on the instruction mix of the bf16x3 split conv kernels (csrc/conv_split.hip: six
 * v_mfma_f32_32x32x16_bf16 products per f32 MAC, a 2 x 4 accumulator-tile wave, one 8-wave workgroup per CU) with
 * random bf16 operands held in registers -- the upper bound of that kernel family under this box's power budget.
 * Runs back-to-back launches: `warm_ms` uncounted, then at least `min_ms` timed as one region.  lds_reads = 1: the
 * 18 operand fragments of every k16 step are re-read from LDS as the conv kernels do; lds_reads = 2: the fp16x2 kernels'
 * mix instead (csrc/conv_h2.hip: three v_mfma_f32_32x32x16_f16 products per tile and k16 step, 24 fragment reads per
 * BK = 32 stage).  tflops_bf16 = executed 16-bit MFMA TFLOP/s (divide by the products per f32 MAC -- 6 or 3 -- for the
 * f32-work ceiling); clock_ghz = shader clock read in the kernel (s_memtime per s_memrealtime). */
int odt_probe_mfma_bf16(int device, double warm_ms, double min_ms, int lds_reads, double* tflops_bf16,
                        double* clock_ghz, double* measured_ms, int* launches);

/* NearestNeighborDistanceMetric.distance (deep_sort/nn_matching.py:156-177,
 * _nn_cosine_distance :78-96): gallery [G,D] float32 rows of all tracks
 * concatenated, seg_offsets [T+1] row ranges per track, dets [N,D];
 * cost [T,N] float64 = min over a track's rows of 1 - cos.  Host pointers. */
int odt_nn_cosine(int device, const float* gallery, const int32_t* seg_offsets,
                  int T, const float* dets, int N, int D, double* cost);

/* ---- DeepSORT tracker core, float64 on the host + the HIP cosine kernel.
 * Tracker(metric, max_iou_distance, max_age, n_init) / predict / update / tracks
 * (deep_sort/tracker.py:40-138, track.py, kalman_filter.py, linear_assignment.py,
 * iou_matching.py); scipy.optimize.linear_sum_assignment restated as odt_lsap. */
typedef struct odt_tracker* odt_tracker_handle;
int odt_tracker_create(double max_cosine_distance, int nn_budget, double max_iou_distance,
                       int max_age, int n_init, int device, odt_tracker_handle* out);
int odt_tracker_destroy(odt_tracker_handle t);
int odt_tracker_predict(odt_tracker_handle t);
/* detections of one frame: tlwh [N,4] float64, confidence [N] float64, features [N,D] float32 */
int odt_tracker_update(odt_tracker_handle t, const double* tlwh, const double* conf,
                       const float* feats, int N, int D);
/* snapshot of the track table in list order; any output pointer may be NULL; *n = #tracks.
 * state: 1 tentative, 2 confirmed.  mean [n,8], covariance [n,8,8]. */
int odt_tracker_tracks(odt_tracker_handle t, int cap, int32_t* ids, int32_t* state,
                       int32_t* time_since_update, int32_t* hits, int32_t* age, double* mean,
                       double* covariance, int* n);
/* scipy.optimize.linear_sum_assignment(cost[nr,nc]) -> n = min(nr,nc) (row, col) pairs by row */
int odt_lsap(const double* cost, int nr, int nc, int32_t* rows, int32_t* cols, int* n);
/* application_util/preprocessing.py:6-73 non_max_suppression (the tracker-side duplicate filter of obj_detect_tracking.py:
 * 662-668): boxes (x, y, w, h) float64 [n,4]; candidates are visited from the back of `order` [n] (the caller's
 * np.argsort of the scores -- or of the bottom edges when the reference is called without scores -- so that ties keep
 * numpy's order; NULL: a stable ascending sort of `scores`, or of the bottom edges when scores is NULL too); overlap =
 * intersection / area of the lower-ranked box with the +1 pixel convention, suppressed when overlap > max_overlap.
 * pick receives the kept indices in pick order. */
int odt_tracker_nms(const double* boxes_xywh, const double* scores, const int32_t* order, int n, double max_overlap, int32_t* pick,
                    int* npick);

/* ---- TMOT / JDE tracker core (reference tmot/multitracker.py JDETracker; driver
 * obj_detect_tracking_multi_queuer_tmot.py:543-583, :707-714).  Host C++ like the DeepSORT core.
 * id_counter is BaseTrack._count (basetrack.py:13,33-36): the reference shares it between all
 * tracker instances of the process, so the caller owns it. ----------------------------------- */
typedef struct odt_tmot* odt_tmot_handle;
int odt_tmot_create(double conf_thres, double track_max_second_lost, double emb_max_dist, double iou_max_dist1,
                    double iou_max_dist2, double emb_smooth_alpha, double frame_gap, double frame_rate,
                    odt_tmot_handle* out);
int odt_tmot_destroy(odt_tmot_handle t);
int odt_tmot_reset(odt_tmot_handle t);
/* JDETracker.update(detections): tlwh [n,4] f64, conf [n] f64, feats [n,dim] f32 -> n_out output tracks */
int odt_tmot_update(odt_tmot_handle t, const double* tlwh, const double* conf, const float* feats, int n, int dim,
                    int* id_counter, int* n_out);
/* which: 0 output_stracks of the last update | 1 tracked | 2 lost | 3 removed; NULL arrays are skipped */
int odt_tmot_tracks(odt_tmot_handle t, int which, int cap, int32_t* ids, int32_t* state, int32_t* activated,
                    double* tlwh, double* det_tlwh, double* det_conf, double* score, int32_t* tracklet_len,
                    int32_t* start_frame, int32_t* frame_id, int* n);

/* ---- stand-alone op entry points (host pointers), used by the staged parity
 * tests; each runs exactly the kernels odt_forward uses. ------------------- */

/* conv2d (nn.py:337-381) + folded BN + optional residual + ReLU.
 * in [B,H,W,ldc] NHWC (first Cin channels used), wt HWIO [kh,kw,Cin,Cout],
 * out [B,Ho+oy,Wo+ox,Cout] written at offset (oy,ox) (rest zero).
 * res_mode 0 none, 1 same-shape [B,Ho,Wo,Cout], 2 nearest-2x of
 * [B,ceil(Ho/2),ceil(Wo/2),Cout]. */
int odt_op_conv2d(int device, const float* in, int B, int H, int W, int Cin,
                  const float* wt, const float* bias, int kh, int kw, int Cout,
                  int stride, int dil, int pad_t, int pad_l, int Ho, int Wo,
                  int oy, int ox, const float* res, int res_mode, int relu,
                  float* out);

/* 1x1 conv over the K-concatenation of two inputs (the fused conv3 + convshortcut of a stage-entry
 * bottleneck): out[b,y,x,:] = relu?( a[b,y,x,:] @ wa + b2[b, y*stride_b, x*stride_b, :] @ wb + bias ).
 * a [B,Ho,Wo,Ca], b2 [B,Hb,Wb,Cb], wa [Ca,Cout], wb [Cb,Cout], out [B,Ho,Wo,Cout]. */
int odt_op_conv2d_cat(int device, const float* a, int B, int Ho, int Wo, int Ca, const float* b2, int Hb,
                      int Wb, int Cb, int stride_b, const float* wa, const float* wb, const float* bias,
                      int Cout, int relu, float* out);
/* The tail of a bottleneck block (nn.py:503-521) on the fp16x2 kernels: conv2 (3x3 stride 1, 'SAME' for dilation dil = 1 or 2,
 * C -> C = 64, 128 or 256 (res2 / res3 / res4 tails), C3 % 64 == 0, + bias, ReLU; w2 [3,3,C,C] HWIO) -> conv3 (1x1, C -> C3, + bias (+ res [B,H,W,C3]), ReLU if relu3;
 * w3 [C,C3]).  fuse = 1: one launch, conv3 evaluated from conv2's accumulators inside the 3x3 kernel (what the plan runs
 * for res4 at b = 8 @1080p); fuse = 0: the two launches it replaces.  in [B,H,W,C], out [B,H,W,C3]. */
int odt_op_bottleneck_tail(int device, const float* in, int B, int H, int W, int C, const float* w2,
                           const float* b2, int dil, const float* w3, const float* b3, int C3,
                           const float* res, int relu3, int fuse, float* out);
/* The ResNet stem on an already padded frame tensor (nn.py:860-896 + 784-792): conv0 7x7 stride 2 VALID over frame_pad
 * [B,Hp,Wp,3] (+ bias, ReLU) -> 3x3 stride 2 max-pool over the top/left zero-padded map; fp16x2 arithmetic.  fuse = 1: the
 * one launch of conv_stem_kernel, 0: the two launches it replaces.  out: [B, Hq, Wq, 64], Hq = ((Hp - 7) / 2 + 1 + 1 - 3) / 2 + 1
 * (Wq alike).  grid > 0: that many workgroups (tests: several tiles per persistent workgroup). */
int odt_op_stem(int device, const float* frame_pad, int B, int Hp, int Wp, const float* w_hwio, const float* bias,
                int fuse, int grid, float* out);
/* image preprocess (models.py:340-355) + zero pad -> [B,Hp,Wp,4] */
int odt_op_preprocess(int device, const void* frames, int dtype, int B, int H,
                      int W, int pad_t, int pad_l, int Hp, int Wp, float* out);
/* 3x3 stride-2 max pool after zero pad top/left 1 (nn.py:890-896), NHWC */
int odt_op_maxpool(int device, const float* in, int B, int H, int W, int C,
                   float* out);
/* tf.nn.top_k canonical (score desc, index asc) */
int odt_op_topk(int device, const float* scores, int n, int k, int32_t* idx_out);
/* tf.image.non_max_suppression (ties: lower index first) */
int odt_op_nms(int device, const float* boxes, const float* scores, int n,
               int max_out, float iou_thresh, int32_t* idx_out, int* n_out);
/* generate_fpn_proposals (models.py:402-436 / :2458-2522): rpn [L][B,h,w,16]
 * (ch 0..2 logits, 3+4a+c deltas, ch 15 unused),
 * anchors [L][S_l,S_l,3,4]; out props [B,K,4], nprops [B]. */
int odt_op_proposals(int device, int graph, int B, int L, const int* hs,
                     const int* ws, const int* fields, const float* const* rpn,
                     const float* const* anchors, int img_h, int img_w, int K,
                     float nms_thresh, float decode_clip, float* props,
                     int32_t* nprops);
/* multilevel_roi_align (models.py:465-485): feats[4] NHWC [B,h_l,w_l,C];
 * boxes [R,4] image coords, box_ind [R]; out NCHW [R,C,7,7], pooled [R,C]. */
int odt_op_roi_align(int device, int B, int C, const int* hs, const int* ws,
                     const float* const* feats, const float* strides,
                     const float* boxes, const int32_t* box_ind, int R,
                     float* out_nchw, float* pooled);
/* inference tail (models.py:828-843 + fastrcnn_predictions :1258-1304 or
 * fastrcnn_predictions_multibatch :2924-2976).  cls_logits [B*K,C],
 * box_logits [B*K,C,4] (class 0 present, ignored), props [B,K,4], nprops [B]. */
int odt_op_detections(int device, int graph, int B, int K, int C,
                      const float* cls_logits, const float* box_logits,
                      const float* props, const int32_t* nprops, int img_h,
                      int img_w, const float* reg_weights, float decode_clip,
                      float score_thresh, float nms_thresh, int per_im,
                      float* boxes, float* probs, int32_t* labels,
                      int32_t* valid);

/* the selection half of the tail alone, on caller-supplied boxes / scores: per-class NMS + merged top per_im
 * (graph ODT_GRAPH_SINGLE: candidates are score > score_thresh, nms_return_masks + fastrcnn_predictions,
 * models.py:1202-1223,1258-1304; ODT_GRAPH_MULTI: every one of the first ncand[b] rows is a candidate,
 * tf.image.combined_non_max_suppression(score_threshold=-inf, clip_boxes=False), models.py:2959-2965, the other
 * images' rows counting as zero-score padding slots).  boxes_in [B,N,C,4], scores_in [B,N,C] (foreground classes
 * only), ncand [B]; labels come back 1-based.  For feeding the kernels TensorFlow's published known-answer vectors. */
int odt_op_class_nms(int device, int graph, int B, int N, int C, const float* boxes_in,
                     const float* scores_in, const int32_t* ncand, float score_thresh,
                     float nms_thresh, int per_im, float* boxes, float* scores,
                     int32_t* labels, int32_t* valid);

#ifdef __cplusplus
}
#endif
#endif /* ODT_H_ */
