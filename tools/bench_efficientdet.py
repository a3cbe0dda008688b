#!/usr/bin/env python
"""BASELINE config #5: EfficientDet forward on one MI355X (random-init weights of the real
architecture, synthetic frames).  python tools/bench_efficientdet.py [--model efficientdet-d7]
[--size 1536] [--frame 1080x1920] [--steps 20]  -> one JSON line."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def measure(model="efficientdet-d7", size=0, frame="", steps=20, warmup=3, cpu_baseline=True, device=0, tmot_frames=30,
            pmc_profile=None):
  """One EfficientDet configuration on one GPU -> the dict bench.py prints under `extra.efficientdet_d7` and this tool
  prints as its JSON line (metric / value / roofline / cpu_baseline / extra)."""
  import torch
  from object_detection_tracking_amd import models
  from object_detection_tracking_amd._lib import ODT_DTYPE_U8
  from object_detection_tracking_amd.config import make_config
  from object_detection_tracking_amd.efficientdet import arch
  from object_detection_tracking_amd.weights import synthetic_frames
  S = size or arch.det_config(model)["image_size"]
  fh, fw = (int(v) for v in frame.split("x")) if frame else (S, S)
  cfg = make_config(is_efficientdet=True, efficientdet_modelname=model, efficientdet_max_detection_topk=5000,
                    short_edge_size=S, max_size=S)
  cfg.max_size = S
  m = models.get_model(cfg, device, weights=arch.synthetic_det_weights(model, 0, gain=arch.bench_gain(model)))
  fr = synthetic_frames(1, fh, fw)[0]
  e = m.engine((fh, fw))
  dev = torch.from_numpy(fr[None].copy()).cuda(device)              # HBM-resident uint8 frame
  step = lambda: e.lib.check(e.lib.dll.odt_forward_async(e.h, dev.data_ptr(), ODT_DTYPE_U8, 1, None))
  for _ in range(warmup):
    step()
  e.synchronize(); torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(steps):
    step()
  e.synchronize(); torch.cuda.synchronize()
  dt = (time.perf_counter() - t0) / steps
  # the timed region checks itself (as bench.py's headline does): what its last device-resident forward left in HBM must
  # equal, bit for bit, a blocking host-to-host forward of the same frame -- and be a non-trivial detection set
  m._inflight = (e, None)
  got = m.predict_collect()
  ref = m.predict(fr)
  names = ("final_boxes", "final_labels", "final_probs", "fpn_box_feat")
  equal = {n: bool(np.array_equal(a, b)) for n, a, b in zip(names, got, ref)}
  verification = {"bit_equal_to_blocking_forward": equal, "detections": int(len(got[0])),
                  "what": "outputs left in HBM by the last forward of the timed loop (odt_read_outputs) == blocking odt_forward of "
                          "the same frame from host memory, all four output arrays bit for bit; detections > 0"}
  verified = all(equal.values()) and len(got[0]) > 0 and bool(np.isfinite(got[0]).all())
  # the same video with K consecutive frames in flight (round 6: frame t on handle t mod K, one stream each -- what
  # EfficientDet.predict_stream does; one D7 frame is ~600 dependent launches, most of them far too small for the chip, and
  # frames are independent): device-resident like `value`, every handle's last outputs checked against the blocking forward
  in_flight = {}
  try:
    engs = [e]
    for K in (2, 3, 4):
      while len(engs) < K:
        ek = m.engine((fh, fw), replica=len(engs))
        ek.lib.check(ek.lib.dll.odt_forward_async(ek.h, dev.data_ptr(), ODT_DTYPE_U8, 1, None)); ek.synchronize()
        engs.append(ek)
      n = 3 * steps
      for k in range(2 * K + n):
        if k == 2 * K:
          for ek in engs: ek.synchronize()
          tk = time.perf_counter()
        ek = engs[k % K]
        ek.lib.check(ek.lib.dll.odt_forward_async(ek.h, dev.data_ptr(), ODT_DTYPE_U8, 1, None))
      for ek in engs: ek.synchronize()
      rate = n / (time.perf_counter() - tk)
      same = all(all(bool(np.array_equal(a, b)) for a, b in zip(m._collect(ek), ref)) for ek in engs)
      in_flight[str(K)] = {"fps": rate, "verified": bool(same)}
  except Exception as ex:
    in_flight["failed"] = repr(ex)
  # `value`: the throughput of the product's streaming path (EfficientDet.predict_stream: three frames in flight) where that leg
  # verified; the one-frame-at-a-time rate (what rounds 2-5 reported as `value`) stays beside it
  dt_one = dt
  best_k = 1
  for K in ("2", "3", "4"):
    r = in_flight.get(K)
    if isinstance(r, dict) and r["verified"] and 1.0 / r["fps"] < dt:
      dt = 1.0 / r["fps"]; best_k = int(K)
  t1 = time.perf_counter()
  for _ in range(5):
    out = m.predict(fr)
  host = (time.perf_counter() - t1) / 5
  # BASELINE config #5 end to end: detector (host to host, frame by frame) + tracker-side NMS + the native TMOT / JDE
  # tracker core per tracked class (reference obj_detect_tracking_multi_queuer_tmot.py:536-583); random-init heads carry
  # no class meaning, so odd class ids count as "Person", even ones as "Vehicle", every score is kept
  from object_detection_tracking_amd.application_util import preprocessing
  from object_detection_tracking_amd.deep_sort import create_obj_arrays
  from object_detection_tracking_amd.tmot.multitracker import JDETracker
  id2class = {i: ("Person" if i % 2 else "Vehicle") for i in range(0, 1024)}
  jde = {c: JDETracker(0.0, frame_gap=1.0) for c in ("Person", "Vehicle")}
  nfr, ntr, t_trk = tmot_frames, 0, 0.0

  def track(boxes, labels, probs, feats):
    n = 0
    for cname, trk in jde.items():
      tl, cf, ft = create_obj_arrays(boxes, probs, labels, feats, id2class, [cname], 0.0, 0, 1.0)
      keep = preprocessing.non_max_suppression_native(tl, 0.85, cf)
      # (random-init box heads of the deep models also emit zero-area / overflowed boxes and all-zero features: the aspect
      # ratio / the L2 normalisation of such a detection is NaN in the reference's tracker as well; the bench drops them)
      keep = [k for k in keep if np.isfinite(tl[k]).all() and 1.0 <= tl[k, 2] < 1e6 and 1.0 <= tl[k, 3] < 1e6 and
              np.isfinite(ft[k]).all() and float(np.abs(ft[k]).max()) > 0.0]
      n = len(trk.update([(tl[k], cf[k], ft[k]) for k in keep]))
    return n

  t2 = time.perf_counter()
  for i in range(nfr):
    boxes, labels, probs, feats = m.predict(fr)[:4]
    t3 = time.perf_counter()
    ntr = track(boxes, labels, probs, feats)
    t_trk += time.perf_counter() - t3
  tmot_dt = (time.perf_counter() - t2) / nfr
  # the same loop with the tracker's host work of frame i under the detector's forward of frame i+1 (the forward is
  # enqueued on the handle's stream before the previous frame's detections are tracked; odt_read_outputs collects it)
  for trk in jde.values():
    trk.reset() if hasattr(trk, "reset") else None
  jde = {c: JDETracker(0.0, frame_gap=1.0) for c in ("Person", "Vehicle")}
  piped_dt = None
  if hasattr(m, "predict_async"):
    t2 = time.perf_counter()
    m.predict_async(fr)
    for i in range(nfr):
      res = m.predict_collect()
      if i + 1 < nfr:
        m.predict_async(fr)
      track(*res[:4])
    piped_dt = (time.perf_counter() - t2) / nfr
  algo_bytes, algo_flops = arch.algorithmic_traffic_and_flops(model, S, S)
  fused_bytes, _ = arch.algorithmic_traffic_and_flops(model, S, S, fused=True)
  res_extra = {"host_to_host_ms": host * 1e3, "detections": int(len(out[0])),
               "algorithmic_gflop_per_frame": algo_flops / 1e9, "effective_tflops": algo_flops / dt / 1e12,
               "detect_tmot_fps": 1.0 / tmot_dt, "tmot_host_ms_per_frame": 1e3 * t_trk / nfr, "tmot_tracks_last": int(ntr)}
  if piped_dt is not None:
    res_extra["detect_tmot_pipelined_fps"] = 1.0 / piped_dt
  res_extra["frames_in_flight"] = in_flight
  res_extra["one_frame_at_a_time_fps"] = 1.0 / dt_one
  res_extra["value_frames_in_flight"] = best_k
  if hasattr(m, "predict_stream"):
    # detect + TMOT through predict_stream (three frames in flight): the tracker's host work of frame i under the forwards of i+1 .. i+3
    jde = {c: JDETracker(0.0, frame_gap=1.0) for c in ("Person", "Vehicle")}
    t2 = time.perf_counter()
    for res in m.predict_stream([fr] * nfr):
      track(*res[:4])
    res_extra["detect_tmot_predict_stream_fps"] = nfr / (time.perf_counter() - t2)
  try:
    res_extra["handle"] = e.describe()
  except Exception:
    pass
  cpu = None
  if cpu_baseline:
    # the oracle (torch-CPU fp32 restatement of the TF graph; NOT TensorFlow) on the same frame
    import torch as _t
    from oracle import effnet
    w = m.weights
    t2 = time.perf_counter()
    x, sc = effnet.preprocess_resized(fr, (S, S))
    redf = effnet.backbone_forward(m.cfg["backbone"], w, x)
    fpn = effnet.feature_network(model, w, {l: _t.from_numpy(redf[l]) for l in (3, 4, 5)}, (S, S))
    cb = effnet.class_box_nets(model, w, fpn)
    effnet.detect(model, cb, (S, S), image_scale=sc)
    cdt = time.perf_counter() - t2
    cpu = {"value": 1.0 / cdt, "unit": "frames/s", "cores": int(_t.get_num_threads()), "kind": "port",
           "sample": "one frame through oracle.effnet (torch-CPU fp32 + numpy tail), one pass, %.1f s" % cdt}
  traffic = None
  traffic_source = None
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  for name in ([pmc_profile] if pmc_profile else ["r05_pmc_summary_effdet_d7.json", "r04_pmc_summary_effdet_d7.json", "r03_pmc_summary_effdet_d7.json", "r02_pmc_summary_effdet_d7.json"]):
    pmc = os.path.join(root, "profiles", name)
    if model == "efficientdet-d7" and S == 1536 and os.path.exists(pmc):
      # HBM bytes per forward from the committed rocprofv3 --pmc passes (tools/gpurun/r2_effdet_pmc.sh), FETCH_SIZE doubled
      traffic = json.load(open(pmc))["hbm_GB_per_forward_fetch_x2"] * 1e9
      traffic_source = ("profiles/%s: separate rocprofv3 --pmc passes of tools/bench_efficientdet.py on the builder's evidence box -- "
                        "NOT measured by the run that prints this line" % name)
      break
  res = {"roofline": {"bound": "hbm", "kernel": "whole network (depthwise / SE / fusion kernels are HBM-bound, "
                      "the 1x1 convs small-K MFMA)", "achieved": algo_bytes / dt / 1e9, "peak": 8000.0, "unit": "GB/s",
                      "frac": algo_bytes / dt / 1e9 / 8000.0, "traffic": traffic, "traffic_source": traffic_source,
                      "algorithmic_bytes_per_frame": algo_bytes,
                      "fused_graph_bytes_per_frame": fused_bytes, "frac_vs_fused_graph_bytes": fused_bytes / dt / 1e9 / 8000.0,
                      "frac_measured_traffic": (traffic / dt / 1e9 / 8000.0) if traffic else None,
                      "note": "achieved / frac: bytes of the UNFUSED graph (every operator reads its inputs and writes its "
                              "output once) per second; frac_vs_fused_graph_bytes: against the byte count of the best fusion "
                              "the graph allows (arch.algorithmic_traffic_and_flops(fused=True)); frac_measured_traffic: "
                              "the kernels' own PMC-counted HBM bytes per second"},
         "cpu_baseline": cpu, "metric": "%s FPS @%dx%d input per MI355X" % (model, S, S), "value": 1.0 / dt, "unit": "frames/s",
         "n_gpus": 1, "steps": steps, "warmup": warmup, "ms_per_step": dt * 1e3, "dtype": "f32",
         "verified": bool(verified), "verification": verification,
         "data": "synthetic", "config": {"workload": "%s (EfficientNet backbone + BiFPN + class/box nets + top-5000 / NMS / "
         "per-level ROI features), frame %dx%d scaled on the device, batch 1, 90 classes, random-init weights, frame "
         "resident in HBM (uint8); %d consecutive frame(s) in flight, frame t on handle t mod %d (EfficientDet.predict_stream; "
         "extra.one_frame_at_a_time_fps: one handle, one frame at a time)" % (model, fw, fh, best_k, best_k)},
         "extra": res_extra}
  m.close()
  return res


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--model", default="efficientdet-d7")
  ap.add_argument("--size", type=int, default=0, help="network input (square); 0 = the model's native size")
  ap.add_argument("--frame", default="", help="HxW of the source frames (default: the network input size)")
  ap.add_argument("--steps", type=int, default=20)
  ap.add_argument("--warmup", type=int, default=3)
  ap.add_argument("--no-cpu-baseline", action="store_true")
  ap.add_argument("--device", type=int, default=0)
  a = ap.parse_args()
  print(json.dumps(measure(a.model, a.size, a.frame, a.steps, a.warmup, not a.no_cpu_baseline, device=a.device)), flush=True)


if __name__ == "__main__":
  main()
