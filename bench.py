#!/usr/bin/env python
"""Headline benchmark: detector FPS @1920x1080, batch 8, per MI355X (BASELINE.json).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path (preprocess -> ResNet-101-dilated+FPN -> RPN -> proposals
-> ROIAlign -> box head -> per-class NMS -> appearance-feature ROIAlign + 7x7 mean) over one
batch of 8 synthetic 1920x1080 frames that are already resident in HBM (uint8).  One process
per GPU, one video stream per GPU, weights replicated, no data-path collective (streams are
independent: "weak" scaling; RCCL only carries the barrier and the max-over-ranks timing).
Rank 0 prints ONE JSON line with the contract fields plus `roofline` (the conv implicit-GEMM
launches, HIP-event timed on their launch stream, split by kernel family: the split kernels -- f32 convolution
through exact 16-bit MFMA products, three f16 products per MAC on the `conv_h2` kernels, six bf16 ones on the
`conv_split` kernels -- and the exact-f32 `conv_igemm_kernel`) and, at N=1, `cpu_baseline` (the oracle -- a CPU
restatement of the reference's TF graph, kind "port" -- timed on the host cores over a bounded
sample of the same workload).
"""
import argparse
import json
import os
import sys
import time

# Hardware queues per stream priority (ROCclr reads it when libamdhip64 loads, i.e. before `import torch`): a handle has one
# compute stream and three side streams, the trackers one each; with the default of 4 a second handle's streams share
# hardware queues with the first one's and wait behind its event barriers (two streams per GPU: 309 -> 318 FPS,
# profiles/r06_hw_queues_and_cosine_priority_ab.txt).  An explicit setting in the environment wins.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
BF16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_bf16, dense
H2_PRODUCTS = 3                 # f16 MFMA products per f32 MAC in the fp16x2 kernels (csrc/conv_h2.hip)
SPLIT_PRODUCTS = 6              # bf16 MFMA products per f32 MAC in the conv_split kernels (csrc/conv_split3.hip, conv_split1.hip)


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--gpus", type=int, default=1)
  ap.add_argument("--steps", type=int, default=10)
  ap.add_argument("--warmup", type=int, default=2)
  ap.add_argument("--batch", type=int, default=8)
  ap.add_argument("--height", type=int, default=1080)
  ap.add_argument("--width", type=int, default=1920)
  ap.add_argument("--topk", type=int, default=300, help="rpn_test_post_nms_topk (BASELINE: 300)")
  ap.add_argument("--graph", default="multi", choices=["multi", "single"],
                  help="multi = Mask_RCNN_FPN_multi (batched graph, the headline); single = Mask_RCNN_FPN (BASELINE config #2: "
                       "the b=1 graph of obj_detect_tracking.py -- min-size filter, prob > 1e-4, per-class NMS; needs --batch 1)")
  ap.add_argument("--no-cpu-baseline", action="store_true")
  ap.add_argument("--keep-taps", action="store_true", help="A/B: a dedicated buffer per stage tensor instead of the activation arena")
  ap.add_argument("--no-d7", action="store_true", help="skip the EfficientDet-D7 leg of `extra` (BASELINE config #5)")
  ap.add_argument("--no-affinity", action="store_true", help="N > 1: do not pin the ranks to the CPUs next to their GPU")
  ap.add_argument("--no-extras", action="store_true", help="skip the `extra` measurements (A/B runs)")
  ap.add_argument("--no-live-traffic", action="store_true",
                  help="roofline.traffic from the committed counter summary instead of this run's own rocprofv3 --pmc child passes")
  ap.add_argument("--no-nn-matching", action="store_true",
                  help="A/B: leave the DeepSORT appearance matching (BASELINE config #3: T = 64 tracks x budget 5 against N = 100 "
                       "detections, once per frame) out of the timed step")
  ap.add_argument("--cpu-frames", type=int, default=1, help="frames in the bounded CPU sample (7 passes: 2 warm-up + 5 timed)")
  ap.add_argument("--profile-steps", type=int, default=2)
  ap.add_argument("--rotate", type=int, default=4,
                  help="resident batches (different seeds) rotated through the timed region, so that range slots, proposal "
                       "counts and cache contents change from step to step (1: replay one batch)")
  ap.add_argument("--dist-backend", default="nccl",
                  help="nccl (= RCCL, the default) | gloo (debug: lets N ranks share one GPU)")
  ap.add_argument("--streams", type=int, default=1,
                  help="video streams (independent handles, one HIP stream each) per GPU; each step runs "
                       "one batch on every stream")
  ap.add_argument("--device", type=int, default=None,
                  help="debug: force every rank onto this GPU (with --dist-backend gloo)")
  ap.add_argument("--launcher-selftest", action="store_true",
                  help="no GPU work: start the N ranks exactly as a real run does (self-launch, rendezvous, the "
                       "barrier / all-gather / max-over-ranks reduction of the timed region) over gloo and print the "
                       "contract line's launch fields -- what the CPU suite runs")
  args = ap.parse_args()

  if "WORLD_SIZE" not in os.environ and args.gpus > 1:
    # `python bench.py --gpus N`: start the N ranks ourselves (one process per GPU, the same launcher
    # line the driver uses) and relay rank 0's JSON line
    sys.exit(self_launch(args.gpus))

  import faulthandler
  faulthandler.dump_traceback_later(1500, exit=True)     # a hung run leaves a stack trace instead of a silent timeout

  import torch
  import torch.distributed as dist

  rank = int(os.environ.get("RANK", "0"))
  local_rank = int(os.environ.get("LOCAL_RANK", "0"))
  world = int(os.environ.get("WORLD_SIZE", "1"))
  if world != args.gpus:
    raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s)" % (args.gpus, world))
  if args.launcher_selftest:
    return launcher_selftest(args, rank, world)
  if not torch.cuda.is_available():
    raise SystemExit("bench.py needs a GPU: the product path is the HIP library only")
  if args.device is not None:
    local_rank = args.device
  elif torch.cuda.device_count() < world:
    raise SystemExit("bench.py: --gpus %d needs %d visible GPUs, found %d (one process per GPU; "
                     "--dist-backend gloo --device 0 shares one GPU for debugging)"
                     % (world, world, torch.cuda.device_count()))
  torch.cuda.set_device(local_rank)
  # host placement: with several ranks on one host each is pinned to its share of the CPUs next to its GPU (N = 1
  # keeps the whole host: the CPU baseline below uses every core); reported per rank either way
  from object_detection_tracking_amd.parallel import bind_rank_to_gpu_numa
  try:
    affinity = bind_rank_to_gpu_numa(local_rank, max(world, 1), apply=(world > 1 and not args.no_affinity and args.device is None))
  except Exception as ex:          # never fatal
    affinity = {"bound": False, "error": repr(ex)}
  if world > 1:
    if args.dist_backend == "nccl":
      dist.init_process_group("nccl", rank=rank, world_size=world,
                              device_id=torch.device("cuda", local_rank))
    else:
      dist.init_process_group(args.dist_backend, rank=rank, world_size=world)

  from object_detection_tracking_amd import models
  from object_detection_tracking_amd._lib import ODT_DTYPE_U8
  from object_detection_tracking_amd.config import make_config
  from object_detection_tracking_amd.weights import synthetic_frames, synthetic_weights

  B, H, W = args.batch, args.height, args.width
  multi = args.graph == "multi"
  if not multi and B != 1:
    raise SystemExit("bench.py: --graph single is the b=1 graph (obj_detect_tracking.py:241-242): pass --batch 1")
  cfg = make_config(rpn_test_post_nms_topk=args.topk, im_batch_size=B, max_size=max(H, W),
                    short_edge_size=min(H, W), keep_taps=bool(args.keep_taps))
  weights = synthetic_weights(cfg, seed=0)
  S = max(1, args.streams)
  all_models = [models.get_model(cfg, local_rank, weights=weights, is_multi=multi) for _ in range(S)]
  engs = [m.engine(B, H, W) for m in all_models]
  model, eng = all_models[0], engs[0]
  # one video stream per handle: each gets its own seeded frames -- NROT different batches per stream, all resident in
  # HBM, taken in turn by the steps of the timed region (frame content, recorded tensor ranges, proposal / detection
  # counts and what the caches hold differ from one step to the next, as they do on a video)
  NROT = max(1, args.rotate)
  rot_frames = [[synthetic_frames(B, H, W, seed=1234 + rank + 1000 * i + 77 * r) for r in range(NROT)] for i in range(S)]
  frames = rot_frames[0][0]
  dev_rot = [[torch.from_numpy(f).cuda(local_rank) for f in fs] for fs in rot_frames]      # HBM-resident uint8 input
  dev_frames = dev_rot[0][0]
  torch.cuda.synchronize()
  step_no = [0]
  # engine bring-up, before the warm-up steps: the default handle is guarded (conv_split_family = "auto": the first forward
  # also runs on a bf16x3-only twin handle and the engine keeps whichever the comparison allows); that one-time calibration
  # must never fall into the timed region, whatever --warmup says
  for e, ds in zip(engs, dev_rot):
    e.forward_device_async(ds[0].data_ptr(), ODT_DTYPE_U8)
    e.synchronize()
  # BASELINE config #3 = the detector step + nn_matching: per frame of the batch one cosine nearest-neighbour cost matrix of
  # T = 64 confirmed tracks x budget 5 gallery rows against N = 100 detections, 256-d (deep_sort/nn_matching.py:156-177) --
  # host-to-host through odt_nn_cosine on the tracker's own stream, next to the forward in flight.  Seeded unit-norm
  # features around 64 cluster centres; the forward's own features would need a D2H inside the step.
  nn_match = None
  if multi and not args.no_nn_matching:
    from object_detection_tracking_amd import ops as _ops
    rngm = np.random.default_rng(7)
    centres = rngm.standard_normal((64, 256)).astype(np.float32)
    gal = centres.repeat(5, 0) + 0.1 * rngm.standard_normal((320, 256)).astype(np.float32)
    gal /= np.linalg.norm(gal, axis=1, keepdims=True)
    det = centres[rngm.integers(0, 64, 100)] + 0.1 * rngm.standard_normal((100, 256)).astype(np.float32)
    det /= np.linalg.norm(det, axis=1, keepdims=True)
    seg = (np.arange(65) * 5).astype(np.int32)
    nn_match = (gal.astype(np.float32), seg, det.astype(np.float32))
    _ops.nn_cosine(*nn_match, device=local_rank)
  nn_calls = [0]

  def step():
    r = step_no[0] % NROT
    step_no[0] += 1
    for e, ds in zip(engs, dev_rot):
      e.forward_device_async(ds[r].data_ptr(), ODT_DTYPE_U8)
      if nn_match is not None:
        for _ in range(B):
          _ops.nn_cosine(*nn_match, device=local_rank)
          nn_calls[0] += 1

  def sync_all():
    for e in engs:
      e.synchronize()

  def barrier():
    if world > 1:
      dist.barrier()

  for _ in range(args.warmup):
    step()
  sync_all()
  torch.cuda.synchronize()
  barrier()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(args.steps):
    step()
  sync_all()
  torch.cuda.synchronize()
  barrier()
  dt = time.perf_counter() - t0
  dt, ranks_seen, rank_dts = reduce_timing(dt, rank, world, "cuda" if args.dist_backend == "nccl" else "cpu")
  rank_fps = [S * B * args.steps / t for t in rank_dts]

  # the timed region, checked: the outputs the LAST replayed forward of every stream left in HBM must equal, bit for
  # bit, a blocking odt_forward of the same frames through the host boundary -- and must be a non-trivial detection set
  last = (step_no[0] - 1) % NROT
  verified = verify_timed_region(engs, [fs[last] for fs in rot_frames])
  verified["resident_batches_rotated"] = NROT
  verified["batch_verified"] = last

  # roofline of the dominant kernel family (implicit-GEMM conv, ~99% of the FLOPs): HIP events
  # around every launch on the launch stream, outside the timed region.
  nprof = max(1, args.profile_steps)
  prof = profile_convs(eng, dev_frames.data_ptr(), nprof)

  sustained = None
  if rank == 0:
    sustained = sustained_peak(eng.lib, local_rank)

  # every rank: the PCIe-inclusive pipelined rate of its own stream (the figure that degrades with host contention when
  # N ranks share one host; `value` above is HBM-resident) -- measured by all ranks at the same time
  per_rank = None
  if world > 1:
    barrier()
    mine = {"rank": rank, "local_rank": local_rank, "affinity": affinity}
    try:
      mine.update(pipelined_leg(eng, frames, B, nbatches=6))
    except Exception as ex:
      mine["pipelined_failed"] = repr(ex)
    mine["device_resident_fps"] = rank_fps[rank] if rank < len(rank_fps) else None
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    per_rank = sorted(gathered, key=lambda d: d["rank"])

  # Not part of `value`: (a) the same step through the host boundary (pageable host frames ->
  # odt_forward -> host outputs incl. [M,256,7,7] features): PCIe-inclusive rate; (b) the DeepSORT
  # appearance matching kernel at BASELINE config #3 size (T=64 tracks x budget 5, N=100 dets).
  extra = {}
  if rank == 0 and not args.no_extras and world == 1:        # (N > 1: the scaling runs report the timed region only)
    try:
      from object_detection_tracking_amd import ops
      t1 = time.perf_counter()
      for _ in range(3):
        eng.forward(frames)
      extra["pcie_inclusive_fps"] = 3 * B / (time.perf_counter() - t1)
      # same, through the pipelined ingest (pinned double-buffered staging, H2D / D2H on copy
      # streams overlapping the forward; odt_submit / odt_collect)
      extra.update(pipelined_leg(eng, frames, B, nbatches=16))
      if nn_match is not None:
        # the timed region's step WITHOUT the per-frame nn_matching calls (what rounds 1-4 timed as `value`): the detector
        # alone, same handle, same rotating batches -- the cosine kernels run on a highest-priority stream and take CUs from
        # the one-round conv launches in flight, which is what the difference to `value` measures
        for k in range(2 + 10):
          if k == 2:
            eng.synchronize(); t1 = time.perf_counter()
          eng.forward_device_async(dev_rot[0][k % NROT].data_ptr(), ODT_DTYPE_U8)
        eng.synchronize()
        extra["detector_only_fps_without_nn_matching_in_the_step"] = 10 * B / (time.perf_counter() - t1)
      if NROT > 1:
        # the timed region's step with ONE resident batch replayed (rounds 1-3 timed this): what the rotation changes
        for k in range(2 + 10):
          if k == 2:
            eng.synchronize(); t1 = time.perf_counter()
          eng.forward_device_async(dev_frames.data_ptr(), ODT_DTYPE_U8)
        eng.synchronize()
        extra["single_resident_batch_replayed_fps"] = 10 * B / (time.perf_counter() - t1)
      if world == 1 and S == 1:
        # (c) two independent video streams (two handles, one HIP stream each) sharing the GPU: the
        # tails / low-occupancy layers of one forward overlap with the other stream's kernels
        m2 = models.get_model(cfg, local_rank, weights=weights, is_multi=multi)
        e2 = m2.engine(B, H, W)
        d2 = torch.from_numpy(synthetic_frames(B, H, W, seed=99)).cuda(local_rank)
        for k in range(2 + 6):
          if k == 2:
            eng.synchronize(); e2.synchronize(); t1 = time.perf_counter()
          eng.forward_device_async(dev_frames.data_ptr(), ODT_DTYPE_U8)
          e2.forward_device_async(d2.data_ptr(), ODT_DTYPE_U8)
        eng.synchronize(); e2.synchronize()
        extra["two_streams_per_gpu_fps"] = 2 * 6 * B / (time.perf_counter() - t1)
        m2.close()
        # (d0) the same step with conv_split_family = 3: every split layer on the bf16x3 kernels (six products per MAC),
        # the round-2 arithmetic -- the fp16x2 kernels' gain, same process, same box
        try:
          cfg4 = make_config(rpn_test_post_nms_topk=args.topk, im_batch_size=B, max_size=max(H, W),
                             short_edge_size=min(H, W), conv_split_family=3)
          m4 = models.get_model(cfg4, local_rank, weights=weights, is_multi=multi)
          e4 = m4.engine(B, H, W)
          for k in range(2 + 8):
            if k == 2:
              e4.synchronize(); t1 = time.perf_counter()
            e4.forward_device_async(dev_frames.data_ptr(), ODT_DTYPE_U8)
          e4.synchronize()
          extra["bf16x3_only_fps"] = 8 * B / (time.perf_counter() - t1)
          extra["bf16x3_only_handle"] = {k: e4.describe()[k] for k in ("conv_arith", "bf16x3_split_launches", "fp16x2_split_launches")}
          m4.close()
        except Exception as ex:     # never fatal: `value` above is already measured
          extra["bf16x3_only_fps"] = "failed: %r" % (ex,)
        # (d) the same step with every conv on the exact-f32 MFMA kernel (config conv_arith = "f32"): the
        # other arithmetic mode of the library, measured in the same process on the same box
        try:
          cfg3 = make_config(rpn_test_post_nms_topk=args.topk, im_batch_size=B, max_size=max(H, W),
                             short_edge_size=min(H, W), conv_arith="f32")
          m3 = models.get_model(cfg3, local_rank, weights=weights, is_multi=multi)
          e3 = m3.engine(B, H, W)
          for k in range(1 + 5):
            if k == 1:
              e3.synchronize(); t1 = time.perf_counter()
            e3.forward_device_async(dev_frames.data_ptr(), ODT_DTYPE_U8)
          e3.synchronize()
          extra["exact_f32_mfma_only_fps"] = 5 * B / (time.perf_counter() - t1)
          extra["exact_f32_mfma_only_handle"] = e3.describe()["conv_arith"]
          m3.close()
        except Exception as ex:     # never fatal: `value` above is already measured
          extra["exact_f32_mfma_only_fps"] = "failed: %r" % (ex,)
      try:
        extra.update(detect_track_leg(eng, frames, B, local_rank))
        fast = detect_track_leg(eng, frames, B, local_rank, arrays=True)
        extra["detect_track_arrays_fps"] = fast["detect_track_fps"]        # create_obj_arrays + native tracker NMS + Tracker.update_arrays
        extra["detect_track_arrays_host_ms_per_frame"] = fast["detect_track"]["host_tracking_ms_per_frame"]
      except Exception as ex:       # never fatal: `value` above is already measured
        extra["detect_track_fps"] = "failed: %r" % (ex,)
      if world == 1 and S == 1:
        # (e) trained RPNs score most anchors negative, so images keep fewer than K proposals (zero-padded NMS slots,
        # models.py:2487-2520); the synthetic weights' +1 RPN class bias keeps all K alive.  Same step with the bias
        # shifted by -3 (less box-head work, every selection kernel on its short-list path)
        try:
          w2 = dict(weights)
          w2["rpn/class/b"] = (w2["rpn/class/b"] - 3.0).astype(np.float32)
          m4 = models.get_model(cfg, local_rank, weights=w2, is_multi=multi)
          e4 = m4.engine(B, H, W)
          for k in range(1 + 5):
            if k == 1:
              e4.synchronize(); t1 = time.perf_counter()
            e4.forward_device_async(dev_frames.data_ptr(), ODT_DTYPE_U8)
          e4.synchronize()
          extra["negative_rpn_bias_fps"] = 5 * B / (time.perf_counter() - t1)
          extra["negative_rpn_bias_nproposals"] = [int(v) for v in e4.tap("nproposals").reshape(-1)]
          m4.close()
        except Exception as ex:
          extra["negative_rpn_bias_fps"] = "failed: %r" % (ex,)
      if world == 1 and S == 1 and multi and (B, H, W) == (8, 1080, 1920):
        # (f) BASELINE config #2 on the graph the config names: Mask_RCNN_FPN (models.py:488-973), batch 1
        try:
          extra.update(single_graph_leg(models, make_config, weights, args.topk, H, W, local_rank, sustained=sustained, rotate=NROT))
        except Exception as ex:
          extra["b1_single_graph_fps"] = "failed: %r" % (ex,)
      rng = np.random.default_rng(0)
      gal = rng.standard_normal((320, 256)).astype(np.float32)
      seg = (np.arange(65) * 5).astype(np.int32)
      det = rng.standard_normal((100, 256)).astype(np.float32)
      ops.nn_cosine(gal, seg, det, device=local_rank)
      t1 = time.perf_counter()
      for _ in range(50):
        ops.nn_cosine(gal, seg, det, device=local_rank)
      extra["nn_matching_ms_per_call_host_to_host"] = 1e3 * (time.perf_counter() - t1) / 50

    except Exception as ex:       # the extras are never fatal: `value` is already measured
      extra["extras_failed"] = repr(ex)
  if rank == 0:
    fps = world * S * B * args.steps / dt
    if (B, H, W) == (8, 1080, 1920) and world == 1 and not args.no_extras and not args.no_live_traffic:
      live_pmc_traffic()
    roofline = conv_roofline(prof, sustained, (B, H, W) == (8, 1080, 1920))
    fam = prof["fam"]
    out = {
        "metric": "detector FPS @%dx%d b=%d per MI355X" % (W, H, B),
        "value": fps,
        "unit": "frames/s",
        "n_gpus": world,
        "ranks_seen": ranks_seen,
        "per_rank_fps": rank_fps,
        "per_rank": per_rank if per_rank is not None else [{"rank": 0, "local_rank": local_rank, "affinity": affinity}],
        "verified": bool(verified.get("ok")),
        "verification": verified,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "value_scope": "device-resident: uint8 frames already in HBM when the timed region starts (%d different batches per "
                       "stream, taken in turn), outputs left in HBM (the PCIe-inclusive pipelined rate and the detect+track "
                       "rate are under `extra`)" % NROT,
        "dtype": "f32",
        "arithmetic": arithmetic_of(eng.describe()),
        "handle": eng.describe(),      # the arithmetic mode / kernel families as the handle itself reports them
        "data": "synthetic",
        "config": {"workload": "ResNet-101-dilated+FPN detector + RoI appearance features%s, "
                               "%dx%d, batch %d per GPU, rpn_post_nms_topk %d, 15 classes, "
                               "random-init weights, frames resident in HBM (uint8)" %
                               (" + nn_matching (cosine NN cost of 64 tracks x budget 5 vs 100 detections, once per frame, inside the "
                                "timed step: odt_nn_cosine host-to-host on the tracker stream)" if nn_match is not None else "",
                                W, H, B, args.topk),
                   "graph": "Mask_RCNN_FPN_multi" if multi else "Mask_RCNN_FPN", "streams_per_gpu": S,
                   "nn_matching_calls_in_timed_and_warmup_steps": nn_calls[0]},
        "roofline": roofline,
    }
    out["extra"] = extra
    if world == 1 and not args.no_cpu_baseline:
      out["cpu_baseline"] = cpu_baseline(cfg, weights, frames, args.cpu_frames, multi)
    if world == 1 and S == 1 and not args.no_extras and not args.no_d7 and (B, H, W) == (8, 1080, 1920):
      # (g) BASELINE config #5: EfficientDet-D7 at 1536 x 1536 + the TMOT tracker, its own roofline (HBM) and CPU baseline
      # (in a child process with a time limit: whatever happens there, the headline line above is printed)
      for m in all_models:
        m.close()
      all_models = []
      extra["efficientdet_d7"] = efficientdet_leg(local_rank, not args.no_cpu_baseline)
    print(json.dumps(out), flush=True)
  for m in all_models:
    m.close()
  if world > 1:
    dist.barrier()
    dist.destroy_process_group()



def profile_convs(eng, dev_ptr, nprof):
  """HIP-event times of every conv launch over `nprof` profiled forwards of the device-resident batch at dev_ptr, grouped
  by kernel family (odt_profile_layer tags the layers [fp16x2] / [bf16x3])."""
  from object_detection_tracking_amd._lib import ODT_DTYPE_U8
  eng.profile(True)
  for _ in range(nprof):
    eng.forward_device_async(dev_ptr, ODT_DTYPE_U8)
  eng.synchronize()
  prof = eng.profile_read()
  layers = eng.profile_layers()
  eng.profile(False)
  fam = {"h2": [0, 0.0, 0.0], "b3": [0, 0.0, 0.0], "f32": [0, 0.0, 0.0]}           # launches, flops (per step), ms (sum)
  for name, fl, ms, _ in layers:
    if "[fused into" in name:
      continue                                   # no launch of its own: its FLOPs are counted with the kernel that evaluates it
    f = fam["h2" if name.endswith("[fp16x2]") else ("b3" if name.endswith("[bf16x3]") else "f32")]
    f[0] += 1; f[1] += fl; f[2] += ms
  # the split kernels as one family: fp16x2 (three f16 MFMA products per f32 MAC) + bf16x3 (six bf16 products)
  fam["split"] = [fam["h2"][0] + fam["b3"][0], fam["h2"][1] + fam["b3"][1], fam["h2"][2] + fam["b3"][2]]
  prof["fam"] = fam
  prof["nprof"] = nprof
  return prof


def conv_roofline(prof, sustained, with_pmc_traffic):
  """The `roofline` object of a bench leg from profile_convs()'s event times (see the module docstring / DESIGN.md section 5)."""
  fam, nprof = prof["fam"], prof["nprof"]
  def tf(f):
    return f[1] * nprof / (f[2] * 1e-3) / 1e12 if f[2] > 0 else 0.0
  achieved = prof["conv_flops"] / (prof["conv_ms"] * 1e-3) / 1e12 if prof["conv_ms"] > 0 else 0.0
  split_tf, f32_tf = tf(fam["split"]), tf(fam["f32"])
  # matrix-pipe products the split launches execute per algorithmic f32 MAC (FLOP-weighted over the two kinds)
  split_products = ((H2_PRODUCTS * fam["h2"][1] + SPLIT_PRODUCTS * fam["b3"][1]) / fam["split"][1]) if fam["split"][1] > 0 else SPLIT_PRODUCTS
  common = {
      "all_conv_launches": {"achieved": achieved, "unit": "TFLOP/s (algorithmic f32)",
                            "launches_per_step": prof["conv_launches"] // nprof,
                            "vs_f32_mfma_peak": achieved / F32_MFMA_PEAK_TFLOPS},
      "conv_ms_per_step": prof["conv_ms"] / nprof,
      "step_ms_profiled": prof["total_ms"] / nprof,
      "algorithmic_gflop_per_step": prof["conv_flops"] / nprof / 1e9,
  }
  f32_family = {"kernel": "conv_igemm_kernel (v_mfma_f32_32x32x2_f32 implicit GEMM, %d launches/step)" % fam["f32"][0],
                "achieved": f32_tf, "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": f32_tf / F32_MFMA_PEAK_TFLOPS, "ms_per_step": fam["f32"][2] / nprof}
  if fam["split"][2] > fam["f32"][2] * 0.5:
    # dominant kernel (most of the FLOPs): the split kernels.  `achieved` is ALGORITHMIC f32 FLOP/s; every f32 MAC costs
    # three f16 (fp16x2 launches) or six bf16 (bf16x3 launches) MFMA MACs, so the matrix-pipe ceiling for it is the dense
    # 16-bit peak / products per MAC (executed rate and its fraction of the 16-bit peak given beside it).
    peak = BF16_MFMA_PEAK_TFLOPS / split_products
    roofline = {
        "bound": "mfma",
        "kernel": "split conv kernels: f32 through exact 16-bit MFMA products -- conv_h2 kernels (3 x v_mfma_f32_32x32x16_f16 "
                  "per MAC, %d launches/step, %.0f%% of the conv FLOPs) + conv_split kernels (6 x v_mfma_f32_32x32x16_bf16 "
                  "per MAC, %d launches/step, %.0f%%)" %
                  (fam["h2"][0], 100.0 * fam["h2"][1] / max(1.0, fam["split"][1] + fam["f32"][1]),
                   fam["b3"][0], 100.0 * fam["b3"][1] / max(1.0, fam["split"][1] + fam["f32"][1])),
        "achieved": split_tf, "peak": peak, "unit": "TFLOP/s", "frac": split_tf / peak,
        "peak_note": "dense 16-bit MFMA peak 2500 TFLOP/s / %.3f products per f32 MAC (FLOP-weighted: 3 on the fp16x2 "
                     "launches, 6 on the bf16x3 ones); frac = executed 16-bit MFMA rate / 2500" % split_products,
        "products_per_mac": split_products,
        "executed_bf16_tflops": split_tf * split_products,
        "vs_f32_mfma_peak": split_tf / F32_MFMA_PEAK_TFLOPS,
        "ms_per_step": fam["split"][2] / nprof,
        "by_kind": {
            "fp16x2": {"launches": fam["h2"][0], "achieved": tf(fam["h2"]), "ms_per_step": fam["h2"][2] / nprof,
                       "frac": tf(fam["h2"]) * H2_PRODUCTS / BF16_MFMA_PEAK_TFLOPS},
            "bf16x3": {"launches": fam["b3"][0], "achieved": tf(fam["b3"]), "ms_per_step": fam["b3"][2] / nprof,
                       "frac": tf(fam["b3"]) * SPLIT_PRODUCTS / BF16_MFMA_PEAK_TFLOPS}},
        "traffic": pmc_traffic("split")[0] if with_pmc_traffic else None,
        "traffic_source": pmc_traffic("split")[1] if with_pmc_traffic else None,
        "f32_mfma_family": f32_family,
    }
  else:
    roofline = dict(f32_family)
    roofline.update({"bound": "mfma", "traffic": pmc_traffic("f32")[0] if with_pmc_traffic else None,
                     "traffic_source": pmc_traffic("f32")[1] if with_pmc_traffic else None})
  roofline.update(common)
  if sustained is not None and "bf16_tflops" in sustained and fam["split"][0] > 0:
    # what the bf16 matrix pipe of THIS box sustains on the split kernels' own MFMA mix (random operands, >= 300 ms
    # of back-to-back launches, measured a moment ago in this process): the ceiling under the box's power budget
    roofline["sustained_peak"] = sustained
    # ceiling of the step's split launches at the sustained rates of their own mixes (fp16x2 launches: the f16 three-product
    # mix; bf16x3 launches: the bf16 six-product mix) over their measured time
    sus_h2 = sustained.get("fp16x2_mix_f16_tflops") or sustained["bf16_tflops"]
    ceil_ms = (H2_PRODUCTS * fam["h2"][1] / sus_h2 + SPLIT_PRODUCTS * fam["b3"][1] / sustained["bf16_tflops"]) * nprof / 1e9
    roofline["frac_of_sustained"] = ceil_ms / fam["split"][2] if fam["split"][2] > 0 else 0.0
    roofline["sustained_f32_work_tflops"] = {"fp16x2": sus_h2 / H2_PRODUCTS, "bf16x3": sustained["bf16_tflops"] / SPLIT_PRODUCTS}
  return roofline


def arithmetic_of(desc):
  """The line's `arithmetic` field, written from what the timed handle reports about itself (odt_describe)."""
  nh2, nb3, nf32 = desc.get("fp16x2_split_launches", 0), desc.get("bf16x3_split_launches", 0), desc.get("exact_f32_mfma_launches", 0)
  nfold, ntail = desc.get("convs_fused_into_epilogues", 0), desc.get("bottleneck_tails_fused", 0)
  if nh2 + nb3 == 0:
    return "f32 tensors, f32 accumulation; exact-f32 MFMA products (v_mfma_f32_32x32x2_f32) in all %d conv launches" % nf32
  return ("f32 tensors, f32 accumulation; of the %d conv launches %d evaluate every f32 product as THREE exact f16 x f16 MFMA "
          "products of a 2-way f16 split of both operands (fp16x2: 22 significand bits per operand, one power of two per weight "
          "row and per activation tensor from its recorded |max| -- per pixel row inside the %d fused bottleneck tails; lo x lo "
          "dropped), %d as SIX exact bf16 x bf16 products of a 3-way bf16 split (bf16x3: all 24 bits), %d on the exact-f32 MFMA "
          "(v_mfma_f32_32x32x2_f32); %d further convs run inside another launch's epilogue.  Error at the exact-f32 kernel's "
          "level on this workload (DESIGN.md section 3 and 4; conv_split_family = 3 / conv_arith = \"f32\" select the stricter modes)"
          % (nh2 + nb3 + nf32, nh2, ntail, nb3, nf32, nfold))


_LIVE_TRAFFIC = {}


def sum_counter_csv(path, counter, pred):
  """(sum of `counter` over the dispatches whose kernel name satisfies pred, number of such dispatches) of a rocprofv3
  `--pmc ... --output-format csv` counter_collection file."""
  import csv
  v, disp = 0.0, set()
  with open(path) as fh:
    for r in csv.DictReader(fh):
      if r["Counter_Name"] == counter and pred(r["Kernel_Name"]):
        v += float(r["Counter_Value"]); disp.add(r["Dispatch_Id"])
  return v, len(disp)


def live_pmc_traffic(timeout_s=100):
  """HBM bytes per split-conv launch MEASURED BY THIS RUN (round 6): counters cannot be collected inside the timed region, so the
  bench starts itself twice as a child under `rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE` (separate passes, as
  MI355X_MICROARCH.md prescribes; two forwards each, no extras) and reads the counter CSVs: (2 x FETCH_SIZE + WRITE_SIZE) x 1024
  bytes over the split-conv launches (FETCH_SIZE doubled: the guide's gfx950 correction).  Fills _LIVE_TRAFFIC; any failure --
  no rocprofv3, a timeout -- leaves it empty and pmc_traffic() falls back to the committed summary of the builder's run."""
  import glob, shutil, subprocess, tempfile
  exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
  if not os.path.exists(exe):
    return
  # (the fp16x2 kernels only -- 87 of the step's 88 split launches, 99 % of its FLOPs: the child's bring-up forward on the guard's
  # bf16x3 twin handle runs conv_split3 kernels, which must not count)
  is_split = lambda k: ("conv_h2" in k or "conv_stem" in k) and "kernel" in k and "split_weights" not in k
  tot, launches = {}, None
  t0 = time.perf_counter()
  for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    d = tempfile.mkdtemp(prefix="odt_pmc_")
    try:
      cmd = [exe, "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", d, "-o", "pmc", "--", sys.executable,
             os.path.abspath(__file__), "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-extras", "--no-d7",
             "--profile-steps", "1", "--no-live-traffic"]
      env = dict(os.environ); env.setdefault("TMPDIR", "/tmp")
      subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s, check=True)
      files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
      v, n = sum_counter_csv(files[0], ctr, is_split)
      tot[ctr] = v
      launches = n if launches is None else min(launches, n)
    except Exception as ex:
      _LIVE_TRAFFIC["error"] = repr(ex)[:200]
      return
    finally:
      shutil.rmtree(d, ignore_errors=True)
  if launches:
    _LIVE_TRAFFIC["split"] = ((2.0 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024.0 / launches,
                              "measured by this run: two child processes of this command under rocprofv3 --kernel-trace --pmc "
                              "(FETCH_SIZE | WRITE_SIZE, separate passes, %d fp16x2 conv launches each; (2 x FETCH_SIZE + WRITE_SIZE) x "
                              "1024 B per launch, FETCH_SIZE doubled per the gfx950 correction); %.0f s" % (launches, time.perf_counter() - t0))


def pmc_traffic(mode):
  """HBM bytes per conv launch: this run's own counter passes (live_pmc_traffic) where they ran, else the newest committed summary
  of the separate rocprofv3 --pmc passes of this same command (FETCH_SIZE doubled per the gfx950 correction + WRITE_SIZE;
  tools/pmc_summary.py), or None."""
  if mode in _LIVE_TRAFFIC:
    return _LIVE_TRAFFIC[mode]
  names = ["r06_pmc_summary.json", "r05_pmc_summary.json", "r04_pmc_summary.json", "r03_pmc_summary.json", "r02_pmc_summary.json"] if mode == "f32" else \
          ["r06_pmc_summary_split.json", "r05_pmc_summary_split.json", "r04_pmc_summary_split.json", "r03_pmc_summary_split.json", "r02_pmc_summary_split.json"]
  for name in names:         # the newest committed summary
    try:
      with open(os.path.join(ROOT, "profiles", name)) as fh:
        v = float(json.load(fh)["conv_hbm_bytes_per_launch_fetch_x2" if mode == "f32" else
                                "split_hbm_bytes_per_launch_fetch_x2"])
      return v, ("profiles/%s: separate rocprofv3 --pmc passes of this bench command on the builder's evidence box -- NOT "
                 "measured by the run that prints this line (counters cannot be collected inside a timed run)" % name)
    except Exception:
      continue
  return None, None


def cpu_model_name():
  try:
    with open("/proc/cpuinfo") as fh:
      for line in fh:
        if line.startswith("model name"):
          return line.split(":", 1)[1].strip()
  except OSError:
    pass
  return "unknown"


def verify_timed_region(engs, all_frames):
  """Outputs of the last timed (replayed, device-resident) forward of every stream vs a blocking forward of the same
  frames from host memory: every array bit-equal, detections present.  Returns the `verification` object."""
  import zlib
  res = {"ok": True, "streams": []}
  for e, fr in zip(engs, all_frames):
    got = e.read_outputs(want_feats=True, want_pooled=True)
    nprop = e.tap("nproposals").reshape(-1)
    ref = e.forward(fr, want_feats=True, want_pooled=True)
    names = ("final_boxes", "final_labels", "final_probs", "final_valid_indices", "fpn_box_feat", "pooled")
    equal = {n: bool(np.array_equal(a, b)) for n, a, b in zip(names, got, ref)}
    crc = 0
    for a in got[:4]:
      crc = zlib.crc32(np.ascontiguousarray(a).tobytes(), crc)
    ndet = int(got[3].sum())
    ok = all(equal.values()) and ndet > 0 and int(nprop.min()) > 0 and bool(np.isfinite(got[0]).all())
    res["streams"].append({"bit_equal_to_blocking_forward": equal, "detections": ndet,
                           "proposals_per_frame": [int(v) for v in nprop], "checksum_crc32": "%08x" % (crc & 0xffffffff)})
    res["ok"] = res["ok"] and ok
  res["what"] = ("outputs left in HBM by the last forward of the timed loop (odt_read_outputs) == blocking odt_forward of the "
                 "same frames from host memory, all six output arrays bit for bit; detections > 0; proposals > 0 per frame")
  return res


def sustained_peak(lib, device):
  """roofline.sustained_peak: odt_probe_mfma_bf16 (csrc/probe.hip) -- the split kernels' MFMA mix on random register
  operands, 100 ms warm-up + >= 300 ms timed back-to-back; once more with the kernels' LDS fragment reads."""
  import ctypes as C
  out = {}
  try:
    for key, lds in (("", 0), ("with_lds_fragment_reads_", 1), ("fp16x2_mix_", 2)):
      tf = C.c_double(); ghz = C.c_double(); ms = C.c_double(); n = C.c_int()
      lib.check(lib.dll.odt_probe_mfma_bf16(device, 100.0, 300.0, lds, C.byref(tf), C.byref(ghz), C.byref(ms), C.byref(n)))
      out[key + "bf16_tflops"] = tf.value
      out[key + "clock_ghz"] = ghz.value
      out[key + "measured_ms"] = ms.value
    # (the third loop is the fp16x2 kernels' own mix: v_mfma_f32_32x32x16_f16, three products per tile and k16 step, 24 LDS
    # fragment reads per two k-steps; "bf16_tflops" in its keys means 16-bit MFMA TFLOP/s)
    out["fp16x2_mix_f16_tflops"] = out.pop("fp16x2_mix_bf16_tflops")
    # the ceiling is the better of the two loops: on some boxes the bare register-operand stream is throttled harder
    # (lower matrix-pipe duty at the same clock) than the one that pauses for its LDS fragment reads
    out["registers_only_bf16_tflops"] = out["bf16_tflops"]
    out["bf16_tflops"] = max(out["bf16_tflops"], out["with_lds_fragment_reads_bf16_tflops"])
    out["f32_work_tflops"] = out["bf16_tflops"] / SPLIT_PRODUCTS
    out["frac_of_datasheet_bf16_peak"] = out["bf16_tflops"] / BF16_MFMA_PEAK_TFLOPS
    out["what"] = ("v_mfma_f32_32x32x16_bf16 in the bf16x3 kernels' mix (2 x 4 accumulator tiles per wave, 6 products per k16 "
                   "step, one 8-wave workgroup per CU), random bf16 operands, no global-memory traffic; operands held in "
                   "registers / re-read from LDS per k16 step as the conv kernels do -- bf16_tflops is the better of the two; "
                   "fp16x2_mix_*: v_mfma_f32_32x32x16_f16 in the fp16x2 kernels' mix (3 products per k16 step, 24 LDS fragment "
                   "reads per two k-steps), random f16 operands; back-to-back launches, 100 ms warm-up, >= 300 ms timed each")
  except Exception as ex:
    out["failed"] = repr(ex)
  return out


def pipelined_leg(eng, frames, B, nbatches=10):
  """PCIe-inclusive rates of the ingest pipeline (odt_submit_ex / odt_collect, pooled features back):
  (a) frames handed over in pageable host memory (one staging copy into the pinned slot per batch);
  (b) frames already in the slot's pinned ingest buffer (odt_ingest_buffer: the decoder's output buffer IS the staging
      buffer -- submit(NULL)), i.e. with a pinned frame source."""
  # steady state: the rate between the second result and the last one (two batches are in flight: the first two results carry
  # the pipeline's fill -- two staging copies + H2D in front of the first forward -- which a video pays once, not per batch);
  # the rate over the whole call, fill and drain included, is reported beside it
  nb = max(nbatches, 6)
  t1 = time.perf_counter()
  stamps = []
  for _ in eng.forward_stream([frames] * nb):
    stamps.append(time.perf_counter())
  out = {"pcie_inclusive_pipelined_fps": (len(stamps) - 2) * B / (stamps[-1] - stamps[1]),
         "pcie_inclusive_pipelined_incl_fill_fps": len(stamps) * B / (stamps[-1] - t1),
         "pcie_inclusive_pipelined_batches": nb}
  try:
    pending = []
    for k in range(2):                # fill both slots' pinned buffers once (a decoder would write every batch there)
      np.copyto(eng.ingest_buffer(frames.dtype), frames)
      pending.append(eng.submit(None, want_feats=False, want_pooled=True))
    for t in pending:
      eng.collect(t)
    t1 = time.perf_counter()
    pending, n = [], 0
    for k in range(nbatches):
      eng.ingest_buffer(frames.dtype)           # (selects the slot; its buffer still holds the frames)
      pending.append(eng.submit(None, want_feats=False, want_pooled=True))
      if len(pending) == 2:
        eng.collect(pending.pop(0)); n += 1
    while pending:
      eng.collect(pending.pop(0)); n += 1
    out["pcie_inclusive_pipelined_pinned_source_fps"] = n * B / (time.perf_counter() - t1)
  except Exception as ex:
    out["pcie_inclusive_pipelined_pinned_source_fps"] = "failed: %r" % (ex,)
  return out


def single_graph_leg(models, make_config, weights, topk, H, W, device, steps=20, warmup=3, sustained=None, rotate=4):
  """BASELINE config #2: ResNet-101-dilated+FPN, 1920x1080, batch 1, on `Mask_RCNN_FPN` (reference models.py:488-973:
  min-size filter, prob > 1e-4, per-class tf.image.non_max_suppression) -- device-resident like the headline."""
  import torch
  from object_detection_tracking_amd._lib import ODT_DTYPE_U8
  from object_detection_tracking_amd.weights import synthetic_frames
  cfg1 = make_config(rpn_test_post_nms_topk=topk, im_batch_size=1, max_size=max(H, W), short_edge_size=min(H, W))
  m = models.get_model(cfg1, device, weights=weights, is_multi=False)
  try:
    e = m.engine(1, H, W)
    frs = [synthetic_frames(1, H, W, seed=1234 + 77 * r) for r in range(max(1, rotate))]      # resident frames, taken in turn
    ds = [torch.from_numpy(fr).cuda(device) for fr in frs]
    for k in range(warmup + steps):
      if k == warmup:
        e.synchronize(); t1 = time.perf_counter()
      e.forward_device_async(ds[k % len(ds)].data_ptr(), ODT_DTYPE_U8)
    e.synchronize()
    dt = (time.perf_counter() - t1) / steps
    # the timed region checks itself like the headline's: what the last forward left in HBM == a blocking forward of that frame
    last = (warmup + steps - 1) % len(ds)
    got = e.read_outputs(want_feats=True, want_pooled=True)
    ref = e.forward(frs[last], want_feats=True, want_pooled=True)
    names = ("final_boxes", "final_labels", "final_probs", "final_valid_indices", "fpn_box_feat", "pooled")
    equal = {n: bool(np.array_equal(a, b)) for n, a, b in zip(names, got, ref)}
    ok = all(equal.values()) and int(got[3].sum()) > 0 and bool(np.isfinite(got[0]).all())
    prof = profile_convs(e, ds[0].data_ptr(), 3)
    roof1, arith1, handle1 = conv_roofline(prof, sustained, False), arithmetic_of(e.describe()), e.describe()
    # the same stream with TWO consecutive frames in flight (frame t on handle t mod 2, one replica handle more: what
    # models.predict_stream does by default): a single 1080p frame's launches are latency-bound and leave most of the chip idle.
    # (Three / four frames on handles without side streams reach 223 / 236 FPS in a process of their own -- and 192 / 210 next to
    # other handles' idle streams, as here: profiles/r06_b1_stream_set.txt.)
    two = {}
    try:
      K = 2
      engsK = [e, m.engine(1, H, W, replica=1)]
      engsK[1].forward_device_async(ds[0].data_ptr(), ODT_DTYPE_U8); engsK[1].synchronize()
      nK = 4 * steps
      for k in range(2 * K + nK):
        if k == 2 * K:
          for ek in engsK: ek.synchronize()
          t2 = time.perf_counter()
        engsK[k % K].forward_device_async(ds[k % len(ds)].data_ptr(), ODT_DTYPE_U8)
      for ek in engsK: ek.synchronize()
      dtK = (time.perf_counter() - t2) / nK
      okK = True
      for j, ek in enumerate(engsK):              # the last frame each handle saw: frames k with k % K == j
        kl = max(k for k in range(2 * K + nK) if k % K == j)
        gK = ek.read_outputs(want_feats=True, want_pooled=True)
        rK = ek.forward(frs[kl % len(ds)], want_feats=True, want_pooled=True)
        okK = okK and all(bool(np.array_equal(a, b)) for a, b in zip(gK, rK)) and int(gK[3].sum()) > 0
      two = {"frames_in_flight_fps": 1.0 / dtK, "frames_in_flight": K, "frames_in_flight_verified": bool(okK),
             "frames_in_flight_note": "frame t on handle t mod %d (models.predict_stream): each forward is still batch 1; results in "
                                      "frame order, one frame later" % K}
    except Exception as ex:
      two = {"frames_in_flight_fps": "failed: %r" % (ex,)}
    return {"b1_single_graph_fps": 1.0 / dt, "b1_frames_in_flight_fps": two.get("frames_in_flight_fps"), "b1_single_graph": {"graph": "Mask_RCNN_FPN", "ms_per_frame": 1e3 * dt, "steps": steps, **two,
                                                               "detections": int(got[3].sum()), "verified": bool(ok),
                                                               "verification": {"bit_equal_to_blocking_forward": equal,
                                                                                "resident_frames_rotated": len(ds)},
                                                               "roofline": roof1,
                                                               "arithmetic": arith1,
                                                               "handle": handle1}}
  finally:
    m.close()


def efficientdet_leg(device, with_cpu_baseline, timeout=420):
  """tools/bench_efficientdet.py (EfficientDet-D7 @1536x1536 + TMOT) as a child process; its JSON object, or a
  {"failed": ...} record -- never an exception."""
  import subprocess
  cmd = [sys.executable, os.path.join(ROOT, "tools", "bench_efficientdet.py"), "--model", "efficientdet-d7", "--steps", "20",
         "--warmup", "3", "--device", str(device)] + ([] if with_cpu_baseline else ["--no-cpu-baseline"])
  try:
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if r.returncode != 0 or not lines:
      return {"failed": "rc %d: %s" % (r.returncode, r.stderr[-400:])}
    return json.loads(lines[-1])
  except Exception as ex:
    return {"failed": repr(ex)}


def cpu_baseline(cfg, weights, frames, nframes, multi=True):
  """Oracle (CPU restatement of the TF graph; NOT TensorFlow) on a bounded sample, by the protocol of
  BASELINE.md section 3: all host cores, 2 warm-up passes, median of 5 timed passes."""
  import torch
  from oracle.graph import OracleModel
  n = max(1, min(nframes, frames.shape[0]))
  nproc = os.cpu_count() or 1
  # torch's own default thread count (= physical cores; forcing all 256 logical CPUs of the GPU box's host
  # oversubscribes the oneDNN convs: one pass went from 8.9 s to more than 80 s)
  om = OracleModel(cfg, weights)
  times = []
  for i in range(2 + 5):
    t0 = time.perf_counter()
    if multi:
      om.forward_multi(frames[:n])
    else:
      om.forward(frames[0])
    if i >= 2:
      times.append(time.perf_counter() - t0)
  med = float(np.median(times))
  return {"value": n / med, "unit": "frames/s", "cores": int(torch.get_num_threads()), "nproc": nproc,
          "cpu_model": cpu_model_name(), "torch": torch.__version__,
          "kind": "port",
          "protocol": "2 warm-up + median of 5 timed passes", "pass_seconds": [round(t, 3) for t in times],
          "sample": "%d of the step's %d frames through oracle.graph.OracleModel.forward%s "
                    "(torch-CPU fp32 conv/matmul + numpy selection ops: a CPU restatement of the TF graph, not TensorFlow)"
                    % (n, frames.shape[0], "_multi" if multi else "")}


def detect_track_leg(eng, frames, B, device, nbatches=8, arrays=False):
  """BASELINE config #3 end to end: pipelined ingest (uint8 frames from host memory, pooled appearance features
  back) -> create_obj_infos -> tracker-side NMS -> native DeepSORT Tracker.predict / update, two tracked classes,
  frame by frame (reference obj_detect_tracking.py:597-760).  With random-init weights the detector's labels carry
  no meaning, so the bench maps odd class ids to "Person" and even ones to "Vehicle" and keeps every score
  (min_confidence 0): each tracker sees about half of the ~100 detections of a frame, and because the stream
  repeats its frames the tracks persist (T ~ N): the matching cascade, gating, assignment and the cosine kernel
  all run at the size config #3 names."""
  from object_detection_tracking_amd.application_util import preprocessing
  from object_detection_tracking_amd.deep_sort import (NearestNeighborDistanceMetric, Tracker, create_obj_arrays,
                                                       create_obj_infos)
  id2class = {i: ("Person" if i % 2 else "Vehicle") for i in range(0, 1024)}
  trackers = {c: Tracker(NearestNeighborDistanceMetric("cosine", 0.5, 5), max_iou_distance=0.5, max_age=60, n_init=1,
                         device=device) for c in ("Person", "Vehicle")}
  t_track = 0.0
  nd, nt, nframes = [], [], 0
  t0 = time.perf_counter()
  for boxes, labels, probs, valid, _, pooled in eng.forward_stream([frames] * nbatches):
    t1 = time.perf_counter()
    off = 0
    for b in range(B):
      v = int(valid[b])
      fb, fl, fp, ff = boxes[b, :v], labels[b, :v], probs[b, :v], pooled[off:off + v]
      off += v
      for cname, trk in trackers.items():
        if arrays:          # same selection / arithmetic on arrays, no Detection objects
          tl, cf, ft = create_obj_arrays(fb, fp, fl, ff, id2class, [cname], 0.0, 0, 1.0)
          keep = preprocessing.non_max_suppression_native(tl, 0.85, cf)
          trk.predict()
          trk.update_arrays(tl[keep], cf[keep], ft[keep])
          nd.append(len(keep))
          continue
        dets = create_obj_infos(nframes, fb, fp, fl, ff, id2class, [cname], 0.0, 0, 1.0)
        keep = preprocessing.non_max_suppression(np.array([d.tlwh for d in dets]).reshape(-1, 4), 0.85,
                                                 np.array([d.confidence for d in dets]))
        dets = [dets[i] for i in keep]
        trk.predict()
        trk.update(dets)
        nd.append(len(dets))
      nframes += 1
    nt.append(sum(len(t.tracks) for t in trackers.values()))
    t_track += time.perf_counter() - t1
  dt = time.perf_counter() - t0
  return {"detect_track_fps": nframes / dt,
          "detect_track": {"frames": nframes, "tracked_classes": 2,
                           "host_tracking_ms_per_frame": 1e3 * t_track / nframes,
                           "detections_per_frame_per_class": float(np.mean(nd)) if nd else 0.0,
                           "tracks_total_last": nt[-1] if nt else 0,
                           "note": "uint8 frames from pageable host memory through odt_submit_ex (pooled features only), "
                                   "create_obj_infos + tracker NMS + native Tracker (C++ core, one HIP cosine call per "
                                   "update) per class and frame; wall time over the whole loop incl. fill / drain"}}


def reduce_timing(dt, rank, world, cdev):
  """Max over ranks of the timed region + who took part (all-gather of (rank, own clock))."""
  if world == 1:
    return dt, [0], [dt]
  import torch
  import torch.distributed as dist
  mine = torch.tensor([float(rank), dt], dtype=torch.float64, device=cdev)
  allr = [torch.zeros(2, dtype=torch.float64, device=cdev) for _ in range(world)]
  dist.all_gather(allr, mine)
  allr = sorted(allr, key=lambda t: t[0].item())
  t = torch.tensor([dt], dtype=torch.float64, device=cdev)
  dist.all_reduce(t, op=dist.ReduceOp.MAX)
  return float(t.item()), [int(t[0].item()) for t in allr], [float(t[1].item()) for t in allr]


def launcher_selftest(args, rank, world):
  """The launch / rendezvous / reduction path of a real run without a GPU (gloo)."""
  import torch.distributed as dist
  from object_detection_tracking_amd.parallel import bind_rank_to_gpu_numa
  if world > 1:
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dist.barrier()
  # the per-rank record of a real run: who ran where (no GPU here: the affinity plan comes back unbound)
  mine = {"rank": rank, "local_rank": int(os.environ.get("LOCAL_RANK", "0")),
          "affinity": bind_rank_to_gpu_numa(int(os.environ.get("LOCAL_RANK", "0")), world, bdfs=[None] * world, apply=False)}
  per_rank = [mine]
  if world > 1:
    per_rank = [None] * world
    dist.all_gather_object(per_rank, mine)
    per_rank = sorted(per_rank, key=lambda d: d["rank"])
  t0 = time.perf_counter()
  time.sleep(0.01 * (1 + rank))            # ranks finish at different times: the max must win
  if world > 1:
    dist.barrier()
  dt = time.perf_counter() - t0
  dt, seen, dts = reduce_timing(dt, rank, world, "cpu")
  if rank == 0:
    print(json.dumps({"launcher_selftest": True, "n_gpus": world, "ranks_seen": seen, "per_rank": per_rank,
                      "max_over_ranks_ok": bool(dt >= max(dts) - 1e-12), "steps": args.steps, "warmup": args.warmup}),
          flush=True)
  if world > 1:
    dist.barrier()
    dist.destroy_process_group()
  return 0


def self_launch(n):
  """python bench.py --gpus N (no launcher in front): run torch.distributed.run ourselves."""
  import socket
  import subprocess
  with socket.socket() as sk:
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
  cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
         "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
  env = dict(os.environ)
  env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
  return subprocess.call(cmd, env=env)


if __name__ == "__main__":
  main()
