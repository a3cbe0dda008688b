"""``EfficientDet`` behind the reference's model surface (reference efficientdet_wrapper.py:12-106:
``image`` placeholder (uint8 BGR frame), ``final_boxes`` [R,4] x1y1x2y2 scaled to the original
frame, ``final_labels`` [R] 1..90, ``final_probs`` [R], ``fpn_box_feat`` [R, fpn_num_filters];
``get_feed_dict_forward``), fetched through ``models.Session.run`` like the FPN models.

Config fields read (reference obj_detect_tracking*.py argument names): ``efficientdet_modelname``,
``efficientdet_max_detection_topk``, ``short_edge_size`` / ``max_size`` (= the padded input size,
efficientdet_wrapper.py:237), ``result_score_thres``, ``result_per_im``, ``use_partial_classes`` /
``partial_classes``.

Input: frames of any size; the reference's input processor (dataloader.py:100-123: scale by
min(out_w / w, out_h / h), TF-1.x bilinear resize of the normalised image, zero padding to
(short_edge_size, max_size)) runs on the device, and the boxes come back multiplied by
image_scale_to_original like the reference's.
"""
import numpy as np

from .. import _lib
from .._lib import OdtOutputs, fptr, iptr
from .arch import NUM_ANCHORS, NUM_CLASSES, det_config, feat_sizes
from .backbone import EfficientNetBackbone


def generate_anchors(height, width, anchor_scale, num_scales=3,
                     aspect_ratios=((1.0, 1.0), (1.4, 0.7), (0.7, 1.4))):
  """anchors.Anchors._generate_boxes (reference efficientdet/anchors.py:182-258) -> [N,4] float32
  (y1, x1, y2, x2), level-major, then cell row-major, then (scale octave, aspect)."""
  sizes = feat_sizes(height, width)
  out = []
  for lvl in range(3, 8):
    level = []
    stride = (height / float(sizes[lvl][0]), width / float(sizes[lvl][1]))
    for so in range(num_scales):
      for aspect in aspect_ratios:
        octave = so / float(num_scales)
        ax2 = anchor_scale * stride[1] * 2 ** octave * aspect[0] / 2.0
        ay2 = anchor_scale * stride[0] * 2 ** octave * aspect[1] / 2.0
        x = np.arange(stride[1] / 2, width, stride[1])
        y = np.arange(stride[0] / 2, height, stride[0])
        xv, yv = np.meshgrid(x, y)
        xv = xv.reshape(-1); yv = yv.reshape(-1)
        boxes = np.swapaxes(np.vstack((yv - ay2, xv - ax2, yv + ay2, xv + ax2)), 0, 1)
        level.append(np.expand_dims(boxes, axis=1))
    out.append(np.concatenate(level, axis=1).reshape([-1, 4]))
  return np.vstack(out).astype(np.float32)


def select_partial_classes(weights, class_idxs, num_classes=NUM_CLASSES):
  """``partial_class_idxs`` (efficientdet_wrapper.py:243-250, :402-410: gather of the class logits):
  the columns of the class-predict pointwise conv are gathered once at load time."""
  cols = np.asarray([a * num_classes + int(i) for a in range(NUM_ANCHORS) for i in class_idxs], np.int64)
  out = dict(weights)
  W = np.asarray(weights["class_net/class-predict/pointwise_kernel"])
  out["class_net/class-predict/pointwise_kernel"] = np.ascontiguousarray(W[..., cols])
  out["class_net/class-predict/bias"] = np.ascontiguousarray(np.asarray(weights["class_net/class-predict/bias"])[cols])
  return out


class EfficientDet(object):

  def __init__(self, config, gpuid=0, weights=None, lib=None):
    from ..models import TensorHandle
    self.config = config
    self.gpuid = gpuid
    self.lib = lib if lib is not None else _lib.get_lib()
    self.model_name = config.efficientdet_modelname
    self.cfg = det_config(self.model_name)
    if weights is None:
      # --model_path efficientdet-d0/ (reference COMMANDS.md:27-33): a TF checkpoint directory with
      # the automl variable names, read without TensorFlow; .npz with the same names also works
      path = getattr(config, "model_path", None)
      if not path:
        raise ValueError("EfficientDet: pass weights={name: array} or set config.model_path")
      if str(path).endswith(".npz"):
        from ..weights import load_npz
        weights = load_npz(path)
      else:
        from ..tf_checkpoint import load_checkpoint
        weights = load_checkpoint(str(path))
    self.num_classes = NUM_CLASSES
    if getattr(config, "use_partial_classes", False) and getattr(config, "partial_class_idxs", None):
      weights = select_partial_classes(weights, config.partial_class_idxs)
      self.num_classes = len(config.partial_class_idxs)
    self.weights = weights
    self.height, self.width = int(config.short_edge_size), int(config.max_size)
    self._engines = {}
    self.image = TensorHandle(self, "image")
    self.final_boxes = TensorHandle(self, "final_boxes")
    self.final_labels = TensorHandle(self, "final_labels")
    self.final_probs = TensorHandle(self, "final_probs")
    self.fpn_box_feat = TensorHandle(self, "fpn_box_feat")

  def engine(self, src_hw=None, replica=0):
    """The plan for frames of this size; ``replica`` > 0: a further handle of the same plan (own weights copy, arena and stream):
    ``predict_stream`` keeps consecutive frames in flight on them."""
    size = tuple(src_hw) if src_hw is not None else (self.height, self.width)
    key = size if not replica else size + ("replica", int(replica))
    if key not in self._engines:
      w = dict(self.weights)
      w["effdet/anchors"] = generate_anchors(self.height, self.width, self.cfg["anchor_scale"])
      e = EfficientNetBackbone(
          self.cfg["backbone"], w, 1, self.height, self.width, device=self.gpuid, lib=self.lib,
          det=self.model_name, num_classes=self.num_classes,
          topk=int(getattr(self.config, "efficientdet_max_detection_topk", 5000)),
          score_thresh=float(getattr(self.config, "result_score_thres", 0.0)),
          per_im=int(getattr(self.config, "result_per_im", 100)),
          keep_taps=bool(getattr(self.config, "keep_taps", False)))
      if size != (self.height, self.width):
        e.set_source_size(*size)
      self._engines[key] = e
    return self._engines[key]

  def get_feed_dict_forward(self, imgdata):      # efficientdet_wrapper.py:99-105
    return {self.image: imgdata}

  def predict(self, frame):
    """uint8 (or float32) BGR frame [H0,W0,3] -> (boxes [R,4], labels [R] int32, probs [R],
    fpn_box_feat [R, filters])."""
    frame = np.asarray(frame)
    e = self.engine(frame.shape[:2])
    per = int(getattr(self.config, "result_per_im", 100))
    F_ = self.cfg["fpn_num_filters"]
    boxes = np.zeros((1, per, 4), np.float32); probs = np.zeros((1, per), np.float32)
    labels = np.zeros((1, per), np.int32); valid = np.zeros((1,), np.int32)
    pooled = np.zeros((per, F_), np.float32)
    out = OdtOutputs()
    out.boxes = fptr(boxes); out.probs = fptr(probs); out.labels = iptr(labels); out.valid = iptr(valid)
    out.feats = None; out.pooled = fptr(pooled); out.masks = None
    fr = np.ascontiguousarray(frame[None])
    import ctypes as C
    from .._lib import ODT_DTYPE_F32, ODT_DTYPE_U8
    dt = ODT_DTYPE_U8 if fr.dtype == np.uint8 else ODT_DTYPE_F32
    if dt == ODT_DTYPE_F32:
      fr = np.ascontiguousarray(fr, np.float32)
    self.lib.check(self.lib.dll.odt_forward(e.h, fr.ctypes.data_as(C.c_void_p), dt, 0, None, C.byref(out)))
    r = int(valid[0])
    return boxes[0, :r].copy(), labels[0, :r].copy(), probs[0, :r].copy(), pooled[:r].copy()

  def _enqueue(self, frame, replica=0):
    import ctypes as C
    from .._lib import ODT_DTYPE_F32, ODT_DTYPE_U8
    frame = np.asarray(frame)
    e = self.engine(frame.shape[:2], replica=replica)
    fr = np.ascontiguousarray(frame[None])
    dt = ODT_DTYPE_U8 if fr.dtype == np.uint8 else ODT_DTYPE_F32
    if dt == ODT_DTYPE_F32:
      fr = np.ascontiguousarray(fr, np.float32)
    self.lib.check(self.lib.dll.odt_forward_async(e.h, fr.ctypes.data_as(C.c_void_p), dt, 0, None))
    return e, fr                      # (the frame array must outlive the asynchronous H2D copy)

  def predict_async(self, frame):
    """Enqueue the forward of one frame and return at once (odt_forward_async on the handle's stream): the host is free
    -- e.g. to run the tracker on the PREVIOUS frame's detections (the reference's queuer loop does the detector call
    and the tracker update back to back, obj_detect_tracking_multi_queuer_tmot.py:536-583) -- until predict_collect()."""
    self._inflight = self._enqueue(frame)

  def predict_stream(self, frames, in_flight=3):
    """A video through the detector with ``in_flight`` consecutive frames on the GPU at once (round 6): frame t runs on handle
    t mod in_flight, each on its own stream; results come back in frame order, as predict() would return them.  One D7 frame is
    ~600 dependent launches of ~20 us each, most of them far too small for the chip; frames are independent: 74 -> 98 -> 106
    -> 111 frames/s with one / two / three / four in flight, fewer again beyond -- and four already loses (92) when another
    process holds hardware queues on the same GPU, hence three (profiles/r06_d7_frames_in_flight*.txt, r06_bench_n1.json)."""
    import collections
    n = max(1, int(in_flight))
    pending = collections.deque()
    for k, frame in enumerate(frames):
      if len(pending) == n:
        yield self._collect(pending.popleft()[0])
      pending.append(self._enqueue(frame, replica=k % n))
    while pending:
      yield self._collect(pending.popleft()[0])

  def predict_collect(self):
    """Wait for the forward of predict_async() and return predict()'s tuple (odt_read_outputs)."""
    e, _ = self._inflight
    self._inflight = None
    return self._collect(e)

  def _collect(self, e):
    import ctypes as C
    per = int(getattr(self.config, "result_per_im", 100))
    F_ = self.cfg["fpn_num_filters"]
    boxes = np.zeros((1, per, 4), np.float32); probs = np.zeros((1, per), np.float32)
    labels = np.zeros((1, per), np.int32); valid = np.zeros((1,), np.int32)
    pooled = np.zeros((per, F_), np.float32)
    out = OdtOutputs()
    out.boxes = fptr(boxes); out.probs = fptr(probs); out.labels = iptr(labels); out.valid = iptr(valid)
    out.feats = None; out.pooled = fptr(pooled); out.masks = None
    self.lib.check(self.lib.dll.odt_read_outputs(e.h, C.byref(out)))
    r = int(valid[0])
    return boxes[0, :r].copy(), labels[0, :r].copy(), probs[0, :r].copy(), pooled[:r].copy()

  def _fetch(self, fetches, feed_dict):
    boxes, labels, probs, feats = self.predict(feed_dict[self.image])
    table = {"final_boxes": boxes, "final_labels": labels, "final_probs": probs, "fpn_box_feat": feats}
    return [table[f.name] for f in fetches]

  def close(self):
    for e in self._engines.values():
      e.close()
    self._engines = {}
