"""Detector output -> ``[Detection]`` (host glue between the two halves of the hot path).

Mirrors the reference's deep_sort/utils.py:5-44 ``create_obj_infos``: boxes are divided by the
resize ``scale``; rows are kept when their (optionally COCO->ActEV mapped) class name is one of
``tracking_objs`` and ``round(prob, 7) >= min_confidence``; x1y1x2y2 becomes xywh; a [C,7,7]
feature is averaged to [C] (a [C] feature -- the pooled output of the HIP ROIAlign -- is used
as is); detections shorter than ``min_detection_height`` are dropped.
"""
import numpy as np

from .detection import Detection


def create_obj_infos(cur_frame, final_boxes, final_probs, final_labels, box_feats,
                     targetid2class, tracking_objs, min_confidence, min_detection_height, scale,
                     is_coco_model=False, coco_to_actev_mapping=None):
  # same dtypes and operation order as the reference, so the Detections are bit-identical to its own:
  # the division stays in the boxes' dtype (float32 from the detector), width / height are subtracted in
  # that dtype (utils.py:24-25 does it in place on the row), the score is rounded by numpy's float32
  # __round__ (utils.py:21), and only Detection() widens to float64
  boxes = np.asarray(final_boxes) / scale
  probs = np.asarray(final_probs)
  detections = []
  for j in range(len(boxes)):
    name = targetid2class[int(final_labels[j])]
    if is_coco_model:
      if name not in coco_to_actev_mapping:
        continue
      name = coco_to_actev_mapping[name]
    conf = float(round(probs[j], 7))
    if name not in tracking_objs or conf < min_confidence:
      continue
    box = boxes[j].copy()
    box[2] -= box[0]
    box[3] -= box[1]             # x, y, w, h
    if box[3] < min_detection_height:
      continue
    feat = np.asarray(box_feats[j])
    if feat.ndim > 2:            # [C, 7, 7] -> [C]
      feat = np.mean(feat, axis=(1, 2))
    detections.append(Detection([box[0], box[1], box[2], box[3]], conf, feat))
  return detections


def create_obj_arrays(final_boxes, final_probs, final_labels, box_feats, targetid2class, tracking_objs,
                      min_confidence, min_detection_height, scale, is_coco_model=False,
                      coco_to_actev_mapping=None):
  """The same selection and arithmetic as :func:`create_obj_infos`, vectorised: returns
  (tlwh [n,4] float64, confidence [n] float64, feature [n,D] float32) instead of ``Detection`` objects, for
  ``Tracker.update_arrays`` (a frame's ~100 Python objects cost more than the tracker update itself)."""
  boxes = np.asarray(final_boxes) / scale
  probs = np.asarray(final_probs)
  labels = np.asarray(final_labels)
  if boxes.size == 0 or len(labels) == 0:       # a frame without detections (also as empty lists / 1-D empty arrays)
    feats = np.asarray(box_feats, dtype=np.float32)
    dim = feats.shape[1] if feats.ndim >= 2 else 0
    return np.zeros((0, 4), np.float64), np.zeros((0,), np.float64), np.zeros((0, dim), np.float32)
  boxes = boxes.reshape(-1, 4)
  names = [targetid2class[int(l)] for l in labels]
  if is_coco_model:
    names = [coco_to_actev_mapping.get(n) for n in names]
  conf = np.round(probs, 7).astype(np.float64) if probs.dtype == np.float32 else np.asarray([float(round(p, 7)) for p in probs])
  keep = np.asarray([n in tracking_objs for n in names], bool) & (conf >= min_confidence) if len(names) else np.zeros(0, bool)
  b = boxes[keep].copy()
  b[:, 2] -= b[:, 0]
  b[:, 3] -= b[:, 1]
  tall = ~(b[:, 3] < min_detection_height)
  feats = np.asarray(box_feats)[keep][tall]
  if feats.ndim > 2:
    feats = np.mean(feats, axis=(2, 3))
  return b[tall].astype(np.float64), conf[keep][tall], np.ascontiguousarray(feats, dtype=np.float32)
