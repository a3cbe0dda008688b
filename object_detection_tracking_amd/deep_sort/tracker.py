"""Native DeepSORT tracker behind the reference's ``Tracker`` surface.

``Tracker(metric, max_iou_distance=0.5, max_age=60, n_init=1)`` with ``predict()``,
``update(detections)`` and ``.tracks`` -- the calls and attributes the reference drivers use
(reference deep_sort/tracker.py:40-138; obj_detect_tracking.py:551-558, :666-695).  All state
(Kalman filter, matching cascade, IoU matching, assignment, gallery) lives in the C++ core of
libodt_hip.so (csrc/tracker_core.cpp); the appearance distances are the HIP cosine kernel.
``metric`` only supplies ``matching_threshold`` and ``budget`` (the gallery is kept natively).
"""
import ctypes as C

import numpy as np

from .. import _lib
from .._lib import c_double_p, f32, fptr, iptr


class TrackState(object):            # reference deep_sort/track.py:4-13
  Tentative = 1
  Confirmed = 2
  Deleted = 3


class Track(object):
  """Read-only view of one native track (reference deep_sort/track.py:15-170 accessors)."""

  def __init__(self, track_id, state, time_since_update, hits, age, mean, covariance):
    self.track_id = int(track_id)
    self.state = int(state)
    self.time_since_update = int(time_since_update)
    self.hits = int(hits)
    self.age = int(age)
    self.mean = mean
    self.covariance = covariance

  def to_tlwh(self):
    ret = self.mean[:4].copy()
    ret[2] *= ret[3]
    ret[:2] -= ret[2:] / 2
    return ret

  def to_tlbr(self):
    ret = self.to_tlwh()
    ret[2:] = ret[:2] + ret[2:]
    return ret

  def is_tentative(self):
    return self.state == TrackState.Tentative

  def is_confirmed(self):
    return self.state == TrackState.Confirmed

  def is_deleted(self):
    return self.state == TrackState.Deleted


class Tracker(object):

  def __init__(self, metric, max_iou_distance=0.5, max_age=60, n_init=1, lib=None, device=0):
    self.metric = metric
    self.max_iou_distance = max_iou_distance
    self.max_age = max_age
    self.n_init = n_init
    self._lib = lib if lib is not None else (getattr(metric, "_lib", None) or _lib.get_lib())
    budget = metric.budget if metric.budget is not None else 0
    self._h = C.c_void_p()
    self._lib.check(self._lib.dll.odt_tracker_create(
        float(metric.matching_threshold), int(budget), float(max_iou_distance), int(max_age),
        int(n_init), int(device), C.byref(self._h)))
    self._tracks = None

  def __del__(self):
    try:
      if self._h:
        self._lib.dll.odt_tracker_destroy(self._h)
        self._h = None
    except Exception:
      pass

  def predict(self):
    self._lib.check(self._lib.dll.odt_tracker_predict(self._h))
    self._tracks = None

  def update(self, detections):
    n = len(detections)
    if n:
      tlwh = np.ascontiguousarray([d.tlwh for d in detections], dtype=np.float64)
      conf = np.ascontiguousarray([d.confidence for d in detections], dtype=np.float64)
      feats = f32([d.feature for d in detections])
      self._lib.check(self._lib.dll.odt_tracker_update(
          self._h, tlwh.ctypes.data_as(c_double_p), conf.ctypes.data_as(c_double_p), fptr(feats),
          n, feats.shape[1]))
    else:
      self._lib.check(self._lib.dll.odt_tracker_update(self._h, None, None, None, 0, 0))
    self._tracks = None

  def update_arrays(self, tlwh, confidence, features):
    """``update`` on arrays (tlwh [n,4] float64, confidence [n], features [n,D] float32) -- what
    ``deep_sort.utils.create_obj_arrays`` returns -- without building ``Detection`` objects."""
    tlwh = np.ascontiguousarray(tlwh, dtype=np.float64).reshape(-1, 4)
    n = tlwh.shape[0]
    if n:
      conf = np.ascontiguousarray(confidence, dtype=np.float64)
      feats = f32(features)
      self._lib.check(self._lib.dll.odt_tracker_update(
          self._h, tlwh.ctypes.data_as(c_double_p), conf.ctypes.data_as(c_double_p), fptr(feats), n, feats.shape[1]))
    else:
      self._lib.check(self._lib.dll.odt_tracker_update(self._h, None, None, None, 0, 0))
    self._tracks = None

  @property
  def tracks(self):
    if self._tracks is None:
      n = C.c_int()
      self._lib.check(self._lib.dll.odt_tracker_tracks(self._h, 0, None, None, None, None, None,
                                                       None, None, C.byref(n)))
      k = n.value
      ids = np.zeros(k, np.int32); st = np.zeros(k, np.int32); tsu = np.zeros(k, np.int32)
      hits = np.zeros(k, np.int32); age = np.zeros(k, np.int32)
      mean = np.zeros((k, 8)); cov = np.zeros((k, 8, 8))
      self._lib.check(self._lib.dll.odt_tracker_tracks(
          self._h, k, iptr(ids), iptr(st), iptr(tsu), iptr(hits), iptr(age),
          mean.ctypes.data_as(c_double_p), cov.ctypes.data_as(c_double_p), C.byref(n)))
      self._tracks = [Track(ids[i], st[i], tsu[i], hits[i], age[i], mean[i], cov[i])
                      for i in range(k)]
    return self._tracks


def linear_sum_assignment(cost, lib=None):
  """scipy.optimize.linear_sum_assignment through the native core (odt_lsap)."""
  lib = lib if lib is not None else _lib.get_lib()
  c = np.ascontiguousarray(cost, dtype=np.float64)
  nr, nc = c.shape
  k = min(nr, nc)
  rows = np.zeros(max(1, k), np.int32); cols = np.zeros(max(1, k), np.int32)
  n = C.c_int()
  lib.check(lib.dll.odt_lsap(c.ctypes.data_as(c_double_p), nr, nc, iptr(rows), iptr(cols),
                             C.byref(n)))
  return rows[:n.value].astype(np.int64), cols[:n.value].astype(np.int64)
