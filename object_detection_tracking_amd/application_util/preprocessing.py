"""Tracker-side suppression of near-duplicate detections (host, <= 100 boxes per frame).

Mirror of the reference's application_util/preprocessing.py:6-73 ``non_max_suppression``:
boxes are (x, y, w, h); candidates are visited by descending confidence (``np.argsort`` of the
scores, highest last); the overlap measure is NOT IoU but intersection / area of the
lower-scored box, with the +1 pixel convention; boxes whose overlap with a picked box exceeds
``max_bbox_overlap`` are dropped.  Returns the picked indices in pick order.
"""
import numpy as np


def non_max_suppression(boxes, max_bbox_overlap, scores=None):
  if len(boxes) == 0:
    return []
  b = np.asarray(boxes, dtype=np.float64)
  x1, y1 = b[:, 0], b[:, 1]
  x2, y2 = b[:, 0] + b[:, 2], b[:, 1] + b[:, 3]
  area = (x2 - x1 + 1) * (y2 - y1 + 1)
  order = np.argsort(scores) if scores is not None else np.argsort(y2)
  pick = []
  while order.size:
    i = order[-1]
    rest = order[:-1]
    pick.append(i)
    iw = np.maximum(0, np.minimum(x2[i], x2[rest]) - np.maximum(x1[i], x1[rest]) + 1)
    ih = np.maximum(0, np.minimum(y2[i], y2[rest]) - np.maximum(y1[i], y1[rest]) + 1)
    overlap = (iw * ih) / area[rest]
    order = rest[~(overlap > max_bbox_overlap)]
  return pick


def non_max_suppression_native(boxes, max_bbox_overlap, scores=None, lib=None):
  """The same filter through the native core (``odt_tracker_nms``): identical picks -- the visiting order is the same
  ``np.argsort`` call, so ties fall as in the loop above -- without the per-pick numpy round trips (1 ms per call
  for a frame's ~50 boxes)."""
  import ctypes as C
  from .. import _lib
  lib = lib if lib is not None else _lib.get_lib()
  b = np.ascontiguousarray(boxes, dtype=np.float64).reshape(-1, 4)
  n = b.shape[0]
  if n == 0:
    return []
  order = np.ascontiguousarray(np.argsort(scores) if scores is not None else np.argsort(b[:, 1] + b[:, 3]), dtype=np.int32)
  pick = np.zeros(n, np.int32); k = C.c_int()
  lib.check(lib.dll.odt_tracker_nms(b.ctypes.data_as(_lib.c_double_p), None, _lib.iptr(order), n, float(max_bbox_overlap),
                                    _lib.iptr(pick), C.byref(k)))
  return [int(i) for i in pick[:k.value]]
