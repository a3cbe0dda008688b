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
