"""``JDETracker`` with the reference's constructor and ``update(detections)`` (reference
tmot/multitracker.py:169-343; driver obj_detect_tracking_multi_queuer_tmot.py:543-583, :707-714).

All state lives in the native core (csrc/tracker_core.cpp: STrack life cycle, Kalman filter,
embedding + Mahalanobis association, the two IoU stages, lapjv-style thresholded assignment,
duplicate removal); ``update`` returns read-only :class:`STrack` views with the attributes the
driver reads (``track_id``, ``tlwh``, ``cur_det_tlwh``, ``cur_det_conf``, ``score`` ...).
``BaseTrack._count`` is the id counter the reference shares between all tracker instances.
"""
import ctypes as C

import numpy as np

from .. import _lib
from .._lib import c_double_p, f32, fptr, iptr


class TrackState(object):            # reference tmot/basetrack.py:5-9
  New = 0
  Tracked = 1
  Lost = 2
  Removed = 3


class BaseTrack(object):             # reference tmot/basetrack.py:12-36 (the shared id counter)
  _count = 0

  @staticmethod
  def next_id():
    BaseTrack._count += 1
    return BaseTrack._count


class STrack(object):
  """Read-only view of one native track."""

  def __init__(self, track_id, state, is_activated, tlwh, cur_det_tlwh, cur_det_conf, score,
               tracklet_len, start_frame, frame_id):
    self.track_id = int(track_id)
    self.state = int(state)
    self.is_activated = bool(is_activated)
    self._tlwh = tlwh
    self.cur_det_tlwh = cur_det_tlwh
    self.cur_det_conf = float(cur_det_conf)
    self.score = float(score)
    self.tracklet_len = int(tracklet_len)
    self.start_frame = int(start_frame)
    self.frame_id = int(frame_id)

  @property
  def tlwh(self):
    return self._tlwh.copy()

  @property
  def tlbr(self):
    ret = self._tlwh.copy()
    ret[2:] += ret[:2]
    return ret

  @property
  def end_frame(self):
    return self.frame_id

  def __repr__(self):
    return "OT_{}_({}-{})".format(self.track_id, self.start_frame, self.end_frame)


class JDETracker(object):

  def __init__(self, conf_thres, track_max_second_lost=4.0, emb_max_dist=0.7, iou_max_dist1=0.8,
               iou_max_dist2=0.9, emb_smooth_alpha=0.9, frame_gap=8., frame_rate=30., lib=None):
    self._lib = lib if lib is not None else _lib.get_lib()
    self.det_thresh = conf_thres
    self.max_frame_lost = track_max_second_lost * frame_rate / frame_gap
    self.frame_id = 0
    self._h = C.c_void_p()
    self._lib.check(self._lib.dll.odt_tmot_create(
        float(conf_thres), float(track_max_second_lost), float(emb_max_dist), float(iou_max_dist1),
        float(iou_max_dist2), float(emb_smooth_alpha), float(frame_gap), float(frame_rate),
        C.byref(self._h)))

  def __del__(self):
    try:
      if self._h:
        self._lib.dll.odt_tmot_destroy(self._h)
        self._h = None
    except Exception:
      pass

  def reset(self):
    """reference multitracker.py:198-206 (also resets the shared id counter)."""
    self._lib.check(self._lib.dll.odt_tmot_reset(self._h))
    self.frame_id = 0
    BaseTrack._count = 0

  def update(self, detections):
    """detections: list of (tlwh, conf, feature).  Returns the activated tracked STracks."""
    n = len(detections)
    cnt = C.c_int(BaseTrack._count)
    nout = C.c_int()
    if n:
      tlwh = np.ascontiguousarray([d[0] for d in detections], dtype=np.float64)
      conf = np.ascontiguousarray([d[1] for d in detections], dtype=np.float64)
      feats = f32([d[2] for d in detections])
      self._lib.check(self._lib.dll.odt_tmot_update(
          self._h, tlwh.ctypes.data_as(c_double_p), conf.ctypes.data_as(c_double_p), fptr(feats), n,
          feats.shape[1], C.byref(cnt), C.byref(nout)))
    else:
      self._lib.check(self._lib.dll.odt_tmot_update(self._h, None, None, None, 0, 0, C.byref(cnt),
                                                    C.byref(nout)))
    BaseTrack._count = cnt.value
    self.frame_id += 1
    return self._tracks(0)

  def _tracks(self, which):
    n = C.c_int()
    none = [None] * 10
    self._lib.check(self._lib.dll.odt_tmot_tracks(self._h, which, 0, *none, C.byref(n)))
    k = n.value
    i32 = lambda: np.zeros(k, np.int32)
    ids, st, act, tl, sf, fid = i32(), i32(), i32(), i32(), i32(), i32()
    tlwh = np.zeros((k, 4)); dtlwh = np.zeros((k, 4)); dconf = np.zeros(k); score = np.zeros(k)
    dp = lambda a: a.ctypes.data_as(c_double_p)
    self._lib.check(self._lib.dll.odt_tmot_tracks(
        self._h, which, k, iptr(ids), iptr(st), iptr(act), dp(tlwh), dp(dtlwh), dp(dconf), dp(score),
        iptr(tl), iptr(sf), iptr(fid), C.byref(n)))
    return [STrack(ids[i], st[i], act[i], tlwh[i], dtlwh[i], dconf[i], score[i], tl[i], sf[i], fid[i])
            for i in range(k)]

  @property
  def tracked_stracks(self):
    return self._tracks(1)

  @property
  def lost_stracks(self):
    return self._tracks(2)

  @property
  def removed_stracks(self):
    return self._tracks(3)
