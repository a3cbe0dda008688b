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
  boxes = np.asarray(final_boxes) / scale
  detections = []
  for j in range(len(boxes)):
    name = targetid2class[int(final_labels[j])]
    if is_coco_model:
      if name not in coco_to_actev_mapping:
        continue
      name = coco_to_actev_mapping[name]
    conf = float(round(float(final_probs[j]), 7))
    if name not in tracking_objs or conf < min_confidence:
      continue
    x1, y1, x2, y2 = (float(v) for v in boxes[j])
    w, h = x2 - x1, y2 - y1
    if h < min_detection_height:
      continue
    feat = np.asarray(box_feats[j])
    if feat.ndim > 2:            # [C, 7, 7] -> [C]
      feat = feat.mean(axis=(1, 2))
    detections.append(Detection([x1, y1, w, h], conf, feat))
  return detections
