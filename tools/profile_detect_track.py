#!/usr/bin/env python
"""bench.py's detect_track leg (arrays path) alone, with the time spent in each piece of the per-frame host loop while
the detector works on the next batch.  GPU only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main():
  import bench
  from object_detection_tracking_amd import models
  from object_detection_tracking_amd.config import make_config
  from object_detection_tracking_amd.weights import synthetic_frames, synthetic_weights
  B, H, W = 8, 1080, 1920
  cfg = make_config(rpn_test_post_nms_topk=300, im_batch_size=B, max_size=max(H, W), short_edge_size=min(H, W))
  m = models.get_model(cfg, 0, weights=synthetic_weights(cfg, seed=0), is_multi=True)
  eng = m.engine(B, H, W)
  frames = synthetic_frames(B, H, W, seed=1234)
  for _ in eng.forward_stream([frames] * 3):
    pass
  for arrays in (False, True):
    r = bench.detect_track_leg(eng, frames, B, 0, nbatches=10, arrays=arrays)
    print("arrays=%s  %.1f FPS  host tracking %.2f ms/frame  dets/class %.1f" % (
      arrays, r["detect_track_fps"], r["detect_track"]["host_tracking_ms_per_frame"], r["detect_track"]["detections_per_frame_per_class"]))
  # the arrays loop again, piece by piece
  from object_detection_tracking_amd.application_util import preprocessing
  from object_detection_tracking_amd.deep_sort import NearestNeighborDistanceMetric, Tracker, create_obj_arrays
  id2class = {i: ("Person" if i % 2 else "Vehicle") for i in range(0, 1024)}
  trackers = {c: Tracker(NearestNeighborDistanceMetric("cosine", 0.5, 5), max_iou_distance=0.5, max_age=60, n_init=1, device=0)
              for c in ("Person", "Vehicle")}
  acc = {"slice": 0.0, "create": 0.0, "nms": 0.0, "predict": 0.0, "update": 0.0}
  nfr = 0
  P = time.perf_counter
  t0 = P()
  for boxes, labels, probs, valid, _, pooled in eng.forward_stream([frames] * 10):
    off = 0
    for b in range(B):
      a = P()
      v = int(valid[b])
      fb, fl, fp, ff = boxes[b, :v], labels[b, :v], probs[b, :v], pooled[off:off + v]
      off += v
      acc["slice"] += P() - a
      for cname, trk in trackers.items():
        a = P(); tl, cf, ft = create_obj_arrays(fb, fp, fl, ff, id2class, [cname], 0.0, 0, 1.0); acc["create"] += P() - a
        a = P(); keep = preprocessing.non_max_suppression_native(tl, 0.85, cf); acc["nms"] += P() - a
        a = P(); trk.predict(); acc["predict"] += P() - a
        a = P(); trk.update_arrays(tl[keep], cf[keep], ft[keep]); acc["update"] += P() - a
      nfr += 1
  dt = P() - t0
  print("pieces (ms/frame): " + "  ".join("%s %.3f" % (k, 1e3 * v / nfr) for k, v in acc.items()) + "   | %.1f FPS" % (nfr / dt))
  # and with the GPU idle (same trackers, same detections replayed)
  eng.synchronize()
  outs = list(eng.forward_stream([frames] * 2))
  acc2 = 0.0; n2 = 0
  for boxes, labels, probs, valid, _, pooled in outs * 3:
    off = 0
    for b in range(B):
      v = int(valid[b]); fb, fl, fp, ff = boxes[b, :v], labels[b, :v], probs[b, :v], pooled[off:off + v]; off += v
      for cname, trk in trackers.items():
        tl, cf, ft = create_obj_arrays(fb, fp, fl, ff, id2class, [cname], 0.0, 0, 1.0)
        keep = preprocessing.non_max_suppression_native(tl, 0.85, cf)
        trk.predict()
        a = P(); trk.update_arrays(tl[keep], cf[keep], ft[keep]); acc2 += P() - a
      n2 += 1
  print("update with the GPU idle: %.3f ms/frame, tracks %s" % (1e3 * acc2 / n2, [len(t.tracks) for t in trackers.values()]))
  # fresh trackers, GPU idle, the same 80 frames
  trackers = {c: Tracker(NearestNeighborDistanceMetric("cosine", 0.5, 5), max_iou_distance=0.5, max_age=60, n_init=1, device=0)
              for c in ("Person", "Vehicle")}
  per_batch = []
  for boxes, labels, probs, valid, _, pooled in [outs[0]] * 10:
    off = 0; a = P()
    for b in range(B):
      v = int(valid[b]); fb, fl, fp, ff = boxes[b, :v], labels[b, :v], probs[b, :v], pooled[off:off + v]; off += v
      for cname, trk in trackers.items():
        tl, cf, ft = create_obj_arrays(fb, fp, fl, ff, id2class, [cname], 0.0, 0, 1.0)
        keep = preprocessing.non_max_suppression_native(tl, 0.85, cf)
        trk.predict()
        trk.update_arrays(tl[keep], cf[keep], ft[keep])
    per_batch.append(1e3 * (P() - a) / B)
  print("fresh trackers, GPU idle, ms/frame per batch: " + " ".join("%.2f" % v for v in per_batch))
  m.close()


if __name__ == "__main__":
  main()
