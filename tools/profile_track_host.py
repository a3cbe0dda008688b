#!/usr/bin/env python
"""Host-side cost of the per-frame tracking glue of bench.py's detect_track leg (arrays path), without the detector:
synthetic detections of the bench's size (about 100 per frame, two tracked classes, 256-d features, boxes that drift
slowly so that tracks persist), cProfile of the loop.  GPU box: the cosine kernel runs on the device as in the bench."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def main():
  from object_detection_tracking_amd.application_util import preprocessing
  from object_detection_tracking_amd.deep_sort import NearestNeighborDistanceMetric, Tracker, create_obj_arrays
  rng = np.random.default_rng(0)
  id2class = {i: ("Person" if i % 2 else "Vehicle") for i in range(0, 64)}
  trackers = {c: Tracker(NearestNeighborDistanceMetric("cosine", 0.5, 5), max_iou_distance=0.5, max_age=60, n_init=1, device=0)
              for c in ("Person", "Vehicle")}
  n = 100
  boxes0 = rng.uniform(0, 1500, (n, 4)).astype(np.float32); boxes0[:, 2:] = boxes0[:, :2] + rng.uniform(30, 200, (n, 2)).astype(np.float32)
  labels = rng.integers(1, 15, n).astype(np.int32); probs = rng.uniform(0.1, 1, n).astype(np.float32)
  feats0 = rng.standard_normal((n, 256)).astype(np.float32)
  def frame(i):
    boxes = boxes0 + np.float32(0.5 * np.sin(i / 7.0))
    feats = feats0 + 0.01 * rng.standard_normal((n, 256)).astype(np.float32)
    for cname, trk in trackers.items():
      tl, cf, ft = create_obj_arrays(boxes, probs, labels, feats, id2class, [cname], 0.0, 0, 1.0)
      keep = preprocessing.non_max_suppression_native(tl, 0.85, cf)
      trk.predict()
      trk.update_arrays(tl[keep], cf[keep], ft[keep])
  for i in range(30):
    frame(i)
  t = time.perf_counter()
  for i in range(30, 130):
    frame(i)
  print("arrays path: %.3f ms/frame, tracks %s" % ((time.perf_counter() - t) / 100 * 1e3, [len(t.tracks) for t in trackers.values()]))
  pr = cProfile.Profile(); pr.enable()
  for i in range(130, 230):
    frame(i)
  pr.disable()
  pstats.Stats(pr).sort_stats("tottime").print_stats(18)


if __name__ == "__main__":
  main()
