"""``Detection``: the per-object record the tracker consumes.

Same constructor and attributes as the reference's deep_sort/detection.py:5-49
(``tlwh`` float64 [4] = top-left x, y, width, height; ``confidence`` python float;
``feature`` float32 [D]) and the same two conversions, so objects built here can be fed to the
reference's unmodified ``Tracker.update``.
"""
import numpy as np


class Detection(object):

  def __init__(self, tlwh, confidence, feature):
    self.tlwh = np.asarray(tlwh, dtype=np.float64)
    self.confidence = float(confidence)
    self.feature = np.asarray(feature, dtype=np.float32)

  def to_tlbr(self):
    """(x, y, w, h) -> (min x, min y, max x, max y)."""
    x, y, w, h = self.tlwh
    return np.array([x, y, x + w, y + h], dtype=np.float64)

  def to_xyah(self):
    """(x, y, w, h) -> (centre x, centre y, w / h, h)."""
    x, y, w, h = self.tlwh
    return np.array([x + w / 2, y + h / 2, w / h, h], dtype=np.float64)
