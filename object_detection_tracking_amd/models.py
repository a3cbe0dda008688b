"""Host-side mirror of the reference's model surface for the hot path.

The reference builds a TF1 graph once and exposes tensor *handles*
(``model.final_boxes`` ...) that callers fetch with ``sess.run(handles,
feed_dict)`` (reference models.py:97-119 ``get_model``, :965-973 output
identities, :1629-1636 ``get_feed_dict_forward``, :3301
``get_feed_dict_forward_multi``; call sites obj_detect_tracking.py:505-517,
:610-635 and obj_detect_tracking_multi.py:460-467).  This module keeps exactly
that surface -- same factory, same attribute names, same feed-dict methods,
same output dtypes/shapes -- over the C ABI of libodt_hip.so, and adds the
plain ``predict()`` / ``predict_batch()`` calls BASELINE.json asks for.

There is no CPU path: constructing a model without the HIP library or without a
GPU raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import (ODT_DTYPE_F32, ODT_DTYPE_U8, ODT_GRAPH_MULTI, ODT_GRAPH_SINGLE, OdtConfig,
                   OdtOutputs, c_i64_p, f32, fptr, iptr)
from .anchors import fpn_anchor_fields
from .config import HEAD_DECODE_CLIP, finalize_config
from .nn import get_new_hw
from .frozen_pb import load_frozen_pb
from .tf_checkpoint import load_checkpoint
from .weights import expand_class_agnostic_box, load_npz, select_partial_classes


class TensorHandle(object):
  """Opaque stand-in for a TF tensor handle (fetch key for Session.run)."""

  def __init__(self, model, name):
    self.model = model
    self.name = name

  def __repr__(self):
    return "<odt tensor %s:0>" % self.name


class _Graph(object):
  """The part of ``tf.Graph`` the frozen-model route uses: tensors by name (reference
  models.py:219-238 ``self.graph.get_tensor_by_name("%s/final_boxes:0" % var_prefix)``)."""

  def __init__(self):
    self._tensors = {}

  def _register(self, prefix, handle):
    self._tensors["%s/%s:0" % (prefix, handle.name)] = handle

  def _unregister(self, prefix):
    for k in [k for k in self._tensors if k.startswith(prefix + "/")]:
      del self._tensors[k]

  def get_tensor_by_name(self, name):
    if name in self._tensors:
      return self._tensors[name]
    # un-prefixed "final_boxes:0" resolves when exactly one imported model carries it
    hits = [h for k, h in self._tensors.items() if k.split("/", 1)[-1] == name]
    if len(hits) == 1:
      return hits[0]
    raise KeyError("The name %r refers to a Tensor which does not exist%s" %
                   (name, " (ambiguous: %d imported models)" % len(hits) if hits else ""))


_DEFAULT_GRAPH = _Graph()


def get_default_graph():
  return _DEFAULT_GRAPH


class _Engine(object):
  """One static plan (fixed batch, H, W) on one GPU."""

  def __init__(self, lib, config, graph, batch, height, width, weights, device,
               num_class=None):
    self.lib = lib
    self.batch, self.height, self.width = batch, height, width
    self.src_height, self.src_width = height, width
    self.per_im = int(config.result_per_im)
    self.channels = int(config.fpn_num_channel)
    c = OdtConfig()
    c.graph = graph; c.batch = batch; c.height = height; c.width = width
    c.num_class = int(num_class if num_class is not None else config.num_class)
    for i, n in enumerate(config.resnet_num_block):
      c.num_blocks[i] = int(n)
    c.use_dilations = int(bool(config.use_dilations))
    c.fpn_channels = self.channels
    c.head_dim = int(config.fpn_frcnn_fc_head_dim)
    c.rpn_topk = int(config.rpn_test_post_nms_topk)
    c.result_per_im = self.per_im
    c.anchor_field = int(np.ceil(config.max_size / config.anchor_strides[0]))
    c.rpn_nms_thresh = float(config.rpn_proposal_nms_thres)
    c.rpn_decode_clip = float(config.bbox_decode_clip)
    c.head_decode_clip = HEAD_DECODE_CLIP
    for i in range(4):
      c.bbox_reg_weights[i] = float(config.fastrcnn_bbox_reg_weights[i])
    c.result_score_thresh = float(config.result_score_thres)
    c.head_nms_thresh = float(config.fastrcnn_nms_iou_thres)
    self.add_mask = bool(getattr(config, "add_mask", False))
    c.add_mask = int(self.add_mask)
    c.mask_dim = int(getattr(config, "mrcnn_head_dim", 256))
    # arithmetic of the conv / FC products (include/odt.h ODT_ARITH_*): None / "default" | "f32" | "bf16x3"
    arith = getattr(config, "conv_arith", None)
    c.conv_arith = {None: 0, "default": 0, "f32": _lib.ODT_ARITH_F32, "bf16x3": _lib.ODT_ARITH_BF16X3}[arith]
    # conv_split_family: "auto" (the DEFAULT, also for an args object that does not carry the field): start on the fp16x2
    # kernels, run the first forward(s) on a bf16x3-only twin as well and stay on bf16x3 when the pyramid / RPN tensors of
    # the two differ by more than f32 rounding level (_auto_calibrate) -- the guard for checkpoints whose activations do
    # not fit the fp16x2 kernels' per-tensor range (useful content more than 2^17 below a tensor's maximum, DESIGN.md
    # section 3.1).  Explicit: 0 / 2 the fp16x2 kernels where eligible, unguarded | 3 bf16x3 only | 1 one-stage bf16x3.
    fam = getattr(config, "conv_split_family", "auto")
    fam = "auto" if fam is None else fam
    self._auto = None
    self._family_requested = fam
    if fam == "auto" and c.conv_arith == 0:
      # (round 6) the guard stays on watch for the whole stream: every forward records its tensors' |max| anyway (the fp16x2
      # kernels scale by them); when one has grown past `watch_ratio` times the level the last comparison accepted
      # (odt_range_health) -- a scene cut, an exposure change: an outlier the first frames never showed -- the comparison
      # against the bf16x3 twin is re-armed for the next forward.  `args` therefore lives as long as the engine (the weights
      # dict is the model's own).
      self._auto = {"pending": int(getattr(config, "conv_split_auto_frames", 1) or 1), "chosen": 2, "checks": [],
                    "tolerance": float(getattr(config, "conv_split_auto_tol", 2e-5)), "deferred": 0, "incomplete": False,
                    "frames": int(getattr(config, "conv_split_auto_frames", 1) or 1),
                    "watch_ratio": float(getattr(config, "conv_split_auto_watch_ratio", 8.0)), "rearmed": 0, "watch": None,
                    "args": (lib, config, graph, batch, height, width, weights, device, num_class)}
    if fam == "auto":
      fam = 2
    c.conv_split_family = int(fam)
    # debug / parity runs: every stage tensor keeps its own buffer and tap() can read it after a forward; the production
    # default plans the activations into an arena (a stage's memory is reused once its consumers have run)
    c.keep_taps = int(bool(getattr(config, "keep_taps", False)))
    c.tail_overlap = -1 if getattr(config, "tail_overlap", True) in (False, -1) else 0      # (False: everything on the compute stream)
    self.h = C.c_void_p()
    lib.check(lib.dll.odt_create(C.byref(c), device, C.byref(self.h)))
    try:
      for name, arr in weights.items():
        self._load(name, arr)
      for i, a in enumerate(fpn_anchor_fields(config)):
        self._load("anchors/lvl%d" % i, a)
      lib.check(lib.dll.odt_finalize_weights(self.h))
    except Exception:
      lib.dll.odt_destroy(self.h)
      self.h = None
      raise
    B, P, Cn = batch, self.per_im, self.channels
    self._boxes = np.zeros((B, P, 4), np.float32)
    self._probs = np.zeros((B, P), np.float32)
    self._labels = np.zeros((B, P), np.int32)
    self._valid = np.zeros((B,), np.int32)
    self._feats = np.zeros((B * P, Cn, 7, 7), np.float32)
    self._pooled = np.zeros((B * P, Cn), np.float32)
    self._masks = np.zeros((B * P, 28, 28), np.float32) if self.add_mask else None
    self._ticket_want = {}           # ticket -> (want_feats, want_pooled) as submitted
    self._ingest_dtype = ODT_DTYPE_U8
    self._ingest_view = None         # numpy view of the pinned buffer armed by ingest_buffer() (until the next submit)
    self._ingest_views_out = False   # a view into this handle's pinned memory was ever handed out
    self._retired = []               # handles replaced by the "auto" guard whose pinned memory a caller may still hold
    self._profile_on = False

  def set_source_size(self, src_height, src_width):
    """Frames of [B, src_height, src_width, 3] from now on; the bilinear resize to the plan's
    [height, width] (reference resizeImage, nn.py:1540-1560) runs on the device."""
    if (src_height, src_width) != (self.src_height, self.src_width):
      self.lib.check(self.lib.dll.odt_set_source_size(self.h, int(src_height), int(src_width)))
      self.src_height, self.src_width = int(src_height), int(src_width)
      twin = (self._auto or {}).get("twin")
      if twin is not None:
        twin.set_source_size(src_height, src_width)

  def _load(self, name, arr):
    a = f32(arr)
    shape = (C.c_int64 * a.ndim)(*a.shape)
    self.lib.check(self.lib.dll.odt_load_tensor(self.h, name.encode(), fptr(a),
                                                C.cast(shape, c_i64_p), a.ndim))

  def close(self):
    d = object.__getattribute__(self, "__dict__")
    a = d.get("_auto")
    if a is not None and "twin" in a:
      a.pop("twin").close()
    for h in d.get("_retired", []):
      self.lib.dll.odt_destroy(h)
    d["_retired"] = []
    d["_ingest_view"] = None
    if self.h is not None:
      self.lib.dll.odt_destroy(self.h)
      self.h = None

  def __getattribute__(self, name):
    # a closed engine (model.close(), or evicted from the model's plan cache) fails loudly instead of passing a null handle
    if name in ("forward", "submit", "collect", "read_outputs", "forward_device_async", "tap", "set_source_size",
                "profile", "describe") and object.__getattribute__(self, "h") is None:
      raise _lib.OdtError("engine closed (model.close(), or evicted from the model's plan cache: _DetectorBase.max_engines)")
    return object.__getattribute__(self, name)

  def __del__(self):
    try:
      self.close()
    except Exception:
      pass

  def _outputs(self, want_feats, want_pooled):
    out = OdtOutputs()
    out.boxes = fptr(self._boxes); out.probs = fptr(self._probs)
    out.labels = iptr(self._labels); out.valid = iptr(self._valid)
    out.feats = fptr(self._feats) if want_feats else None
    out.pooled = fptr(self._pooled) if want_pooled else None
    out.masks = fptr(self._masks) if self.add_mask else None
    return out

  def _result(self, want_feats, want_pooled):
    total = int(self._valid.sum())
    return (self._boxes.copy(), self._labels.copy(), self._probs.copy(), self._valid.copy(),
            self._feats[:total].copy() if want_feats else None,
            self._pooled[:total].copy() if want_pooled else None)

  def _frames(self, frames):
    fr = np.ascontiguousarray(frames)
    if fr.dtype == np.uint8:
      dt = ODT_DTYPE_U8
    else:
      fr = np.ascontiguousarray(fr, dtype=np.float32)
      dt = ODT_DTYPE_F32
    assert fr.shape == (self.batch, self.src_height, self.src_width, 3), fr.shape
    return fr, dt

  # ---- conv_split_family = "auto" --------------------------------------------------------------------------------------
  AUTO_TAPS = ("p2", "p3", "p4", "p5", "p6", "rpn2", "rpn3", "rpn4", "rpn5", "rpn6")      # (readable after a forward in arena mode too)

  def _calibration_due(self):
    a = self._auto
    return a is not None and a["pending"] > 0 and a["chosen"] == 2

  def _auto_finish(self, incomplete=False):
    a = self._auto
    twin = a.pop("twin", None)
    if twin is not None:
      twin.close()
    if a["chosen"] == 3 or incomplete:
      a.pop("args", None)            # (nothing left to compare: the engine IS the bf16x3 engine, or the guard gave up)
    if incomplete:
      a["incomplete"] = True
      a["pending"] = 0

  def _auto_calibrate(self, run):
    """One calibration forward: `run(engine)` puts the caller's input through an engine (blocking).  The fp16x2 handle
    and a bf16x3-only twin (no range assumption at all) see the same input; if any pyramid / RPN tensor differs by more
    than the tolerance (relative to the tensor's |max|), this engine continues as the twin.  Returns True when the
    engine changed handles.

    Never while a ticket is outstanding: the blocking forward would overwrite the single device output buffers under
    the ticket's copy, and a ticket cannot follow the engine to another handle.  Such calls are skipped (counted in
    describe()); after 16 of them the guard gives up loudly in describe() instead of holding the twin forever."""
    a = self._auto
    if not self._calibration_due():
      return False
    if self._ticket_want:
      a["deferred"] += 1
      if a["deferred"] >= 16:
        self._auto_finish(incomplete=True)
      return False
    import copy
    lib, config, graph, batch, height, width, weights, device, num_class = a["args"]
    if "twin" not in a:
      cfg3 = copy.copy(config)
      cfg3.conv_split_family = 3
      a["twin"] = _Engine(lib, cfg3, graph, batch, height, width, weights, device, num_class)
      if (self.src_height, self.src_width) != (height, width):
        a["twin"].set_source_size(self.src_height, self.src_width)
    twin = a["twin"]
    run(self); run(twin)
    worst, where = 0.0, None
    for name in self.AUTO_TAPS:
      try:
        x, y = self.tap(name), twin.tap(name)
      except _lib.OdtError:
        continue
      d = float(np.abs(x - y).max() / max(1e-30, float(np.abs(y).max())))
      if not np.isfinite(d):
        d = float("inf")
      if d > worst:
        worst, where = d, name
    a["checks"].append({"max_rel_diff": worst, "tensor": where})
    a["pending"] -= 1
    swapped = False
    if worst > a["tolerance"]:
      # stay on bf16x3: this engine takes over the twin's handle.  State bound to the old handle: no tickets (checked
      # above); pinned ingest memory a caller may still hold a view of keeps the old handle alive until close();
      # the profiling switch follows.
      old = self.h
      self.h, twin.h = twin.h, None
      if self._ingest_views_out:
        self._retired.append(old)
      else:
        self.lib.dll.odt_destroy(old)
      self._ingest_views_out = False
      if self._profile_on:
        self.lib.check(self.lib.dll.odt_profile_enable(self.h, 1))
      a["chosen"] = 3
      swapped = True
    if a["chosen"] == 3 or a["pending"] <= 0:
      self._auto_finish()
      if a["chosen"] == 2 and a.pop("rebase", False):
        self.range_health(rebase=True)
    return swapped

  def range_health(self, rebase=False):
    """odt_range_health: the largest growth, over the plan's tensors, of the recorded |max| against the level last accepted
    (rebase=True accepts the current one), with the producing layer's name -- one or two forwards old, free to read."""
    f = C.c_double(); amax = C.c_double(); seen = C.c_longlong(); name = C.create_string_buffer(128)
    self.lib.check(self.lib.dll.odt_range_health(self.h, int(bool(rebase)), C.byref(f), name, 128, C.byref(amax), C.byref(seen)))
    return {"worst_growth": f.value, "tensor": name.value.decode(), "tensor_amax": amax.value, "tensors_seen": int(seen.value)}

  def _watch(self):
    """Continuous half of the "auto" guard: called after every forward / collect / synchronize while the engine runs the
    fp16x2 kernels."""
    a = self._auto
    if a is None or a["chosen"] != 2 or a["pending"] > 0 or a["incomplete"] or "args" not in a:
      return
    h = self.range_health()
    a["watch"] = h
    if h["worst_growth"] > a["watch_ratio"]:
      a["pending"] = a["frames"]     # the next forward also runs on a bf16x3 twin (rebuilt: _auto_calibrate)
      a["rearmed"] += 1
      a["deferred"] = 0
      a["rebase"] = True             # a comparison that keeps fp16x2 accepts the new maxima as the level to watch from

  def forward(self, frames, want_feats=True, want_pooled=False):
    """frames: [B,H,W,3] uint8/float32 BGR host array.  Returns fresh arrays."""
    fr, dt = self._frames(frames)
    if self._calibration_due():
      self._auto_calibrate(lambda e: e.lib.check(e.lib.dll.odt_forward(e.h, fr.ctypes.data_as(C.c_void_p), dt, 0, None,
                                                                       C.byref(e._outputs(False, False)))))
    out = self._outputs(want_feats, want_pooled)
    self.lib.check(self.lib.dll.odt_forward(self.h, fr.ctypes.data_as(C.c_void_p), dt, 0, None,
                                            C.byref(out)))
    self._watch()
    return self._result(want_feats, want_pooled)

  def read_outputs(self, want_feats=True, want_pooled=False):
    """The outputs of the most recently enqueued forward (odt_read_outputs): what :meth:`forward` would have
    returned for the frames of the last :meth:`forward_device_async`."""
    out = self._outputs(want_feats, want_pooled)
    self.lib.check(self.lib.dll.odt_read_outputs(self.h, C.byref(out)))
    return self._result(want_feats, want_pooled)

  def submit(self, frames, want_feats=True, want_pooled=True):
    """Pipelined ingest (odt_submit_ex): enqueue H2D + forward + D2H for one batch of host
    frames and return a ticket at once; at most two tickets may be outstanding.  Only what is
    asked for crosses PCIe on the way back (the [M,C,7,7] features are 40 MB per 8-frame batch,
    their 7x7 mean 0.8 MB)."""
    if frames is None:                # the frames were written into ingest_buffer(): no staging copy
      fr, dt = None, self._ingest_dtype
      view = self._ingest_view        # (None: nothing armed for this ticket -- odt_submit_ex refuses below)
      if view is not None and self._calibration_due():
        # the zero-copy path is guarded like the others: both handles read the armed pinned buffer (a blocking forward
        # does not touch the ingest slots); if the engine moves to the twin's handle, the frames move to ITS buffer
        src = view.ctypes.data_as(C.c_void_p)
        if self._auto_calibrate(lambda e: e.lib.check(e.lib.dll.odt_forward(e.h, src, dt, 0, None,
                                                                            C.byref(e._outputs(False, False))))):
          keep = view
          self.ingest_buffer(np.uint8 if dt == ODT_DTYPE_U8 else np.float32)[...] = keep
      self._ingest_view = None
    else:
      fr, dt = self._frames(frames)
      if self._calibration_due():
        self._auto_calibrate(lambda e: e.lib.check(e.lib.dll.odt_forward(e.h, fr.ctypes.data_as(C.c_void_p), dt, 0, None,
                                                                         C.byref(e._outputs(False, False)))))
    t = C.c_int()
    want = (1 if want_feats else 0) | (2 if want_pooled else 0) | (4 if self.add_mask else 0)
    self.lib.check(self.lib.dll.odt_submit_ex(self.h, fr.ctypes.data_as(C.c_void_p) if fr is not None else None, dt, want,
                                              C.byref(t)))
    self._ticket_want[t.value] = (bool(want_feats), bool(want_pooled))
    return t.value

  def ingest_buffer(self, dtype=np.uint8):
    """The NEXT ticket's pinned host input buffer as a numpy array [B,Hs,Ws,3] (odt_ingest_buffer): a decoder writes
    its frames straight into it and calls ``submit(None)`` -- the pageable -> pinned staging copy of
    ``submit(frames)`` disappears (what 8 co-hosted ranks contend on first is host memory bandwidth)."""
    dt = ODT_DTYPE_U8 if np.dtype(dtype) == np.uint8 else ODT_DTYPE_F32
    buf = C.c_void_p(); nbytes = C.c_size_t()
    self.lib.check(self.lib.dll.odt_ingest_buffer(self.h, dt, C.byref(buf), C.byref(nbytes)))
    self._ingest_dtype = dt
    ctype = C.c_uint8 if dt == ODT_DTYPE_U8 else C.c_float
    n = nbytes.value // C.sizeof(ctype)
    arr = np.ctypeslib.as_array(C.cast(buf, C.POINTER(ctype)), shape=(n,))
    self._ingest_view = arr.reshape(self.batch, self.src_height, self.src_width, 3)
    self._ingest_views_out = True
    return self._ingest_view

  def collect(self, ticket, want_feats=None, want_pooled=None):
    """Wait for a ticket of :meth:`submit`; same return value as :meth:`forward`.  By default what comes back is what
    the ticket was submitted with (features and / or their 7x7 mean); asking for more than that is an error."""
    sub = self._ticket_want.get(ticket, (True, False))
    want_feats = sub[0] if want_feats is None else want_feats
    want_pooled = sub[1] if want_pooled is None else want_pooled
    out = self._outputs(want_feats, want_pooled)
    self.lib.check(self.lib.dll.odt_collect(self.h, ticket, C.byref(out)))
    self._ticket_want.pop(ticket, None)
    self._watch()
    return self._result(want_feats, want_pooled)

  def forward_stream(self, batches, want_feats=False, want_pooled=True):
    """Generator over an iterable of frame batches with two batches in flight: the H2D of
    batch i+1 and the D2H of batch i-1 overlap the forward of batch i.  By default the appearance
    features come back pooled ([M,C]: what create_obj_infos makes of fpn_box_feat anyway,
    reference deep_sort/utils.py:27-28); want_feats=True returns the [M,C,7,7] tensor."""
    pending = []
    for fr in batches:
      pending.append(self.submit(fr, want_feats, want_pooled))
      if len(pending) == 2:
        yield self.collect(pending.pop(0), want_feats, want_pooled)
    while pending:
      yield self.collect(pending.pop(0), want_feats, want_pooled)

  def forward_device_async(self, dev_ptr, dtype, stream=None):
    """Enqueue one forward on frames already resident in HBM (bench path)."""
    if self._calibration_due():
      def run(e):
        e.lib.check(e.lib.dll.odt_forward_async(e.h, C.c_void_p(dev_ptr), dtype, 1, None))
        e.lib.check(e.lib.dll.odt_synchronize(e.h))
      self._auto_calibrate(run)
    self.lib.check(self.lib.dll.odt_forward_async(self.h, C.c_void_p(dev_ptr), dtype, 1,
                                                  C.c_void_p(stream) if stream else None))

  def synchronize(self):
    self.lib.check(self.lib.dll.odt_synchronize(self.h))
    self._watch()

  def describe(self):
    """What the handle runs (odt_describe): conv arithmetic mode, launches per kernel family, policy thresholds."""
    import json
    buf = C.create_string_buffer(16384)
    self.lib.check(self.lib.dll.odt_describe(self.h, buf, 16384))
    d = json.loads(buf.value.decode())
    auto = getattr(self, "_auto", None)      # (EfficientNetBackbone borrows this method: no auto state there)
    if auto is not None:
      d["conv_split_family_auto"] = {"chosen": "bf16x3 (family 3)" if auto["chosen"] == 3 else "fp16x2 (family 2)",
                                     "calibration_forwards_left": max(0, auto["pending"]) if auto["chosen"] == 2 else 0,
                                     "tolerance": auto["tolerance"], "checks": list(auto["checks"]),
                                     "calls_skipped_with_tickets_outstanding": auto["deferred"],
                                     "incomplete": bool(auto["incomplete"]),
                                     "watch": auto.get("watch"), "watch_ratio": auto.get("watch_ratio"),
                                     "rearmed": auto.get("rearmed", 0)}
      d["range_guard"] = ("conv_split_family = \"auto\" (default): fp16x2 kernels checked against a bf16x3-only twin handle on the "
                          "first forward(s), re-armed whenever a tensor's recorded |max| has grown past watch_ratio times the accepted level (odt_range_health)" + ("; GAVE UP: every call so far had tickets outstanding" if auto["incomplete"] else ""))
    elif hasattr(self, "_family_requested"):
      d["range_guard"] = "off (explicit conv_split_family = %r / conv_arith)" % (self._family_requested,)
    return d

  def range_report(self, names=None):
    """Debug handles (keep_taps): per stage tensor the |max| of the last forward and the share of its non-zero elements
    more than 2^17 below it -- what the fp16x2 kernels' per-tensor power of two leaves in the f16 subnormal grid (absolute
    instead of relative precision there).  {name: {"amax", "frac_nonzero_below_2^-17_amax", "elements"}}."""
    out = {}
    for name in (names or ("conv0", "pool0", "c2", "c3", "c4", "c5", "p2", "p3", "p4", "p5", "p6")):
      t = np.abs(self.tap(name))
      amax = float(t.max())
      nz = t > 0
      out[name] = {"amax": amax, "elements": int(t.size),
                   "frac_nonzero_below_2^-17_amax": float(((t < amax * 2.0 ** -17) & nz).sum() / max(1, int(nz.sum())))}
    return out

  def profile(self, enable):
    self.lib.check(self.lib.dll.odt_profile_enable(self.h, int(enable)))
    self._profile_on = bool(enable)

  def profile_read(self):
    ms = C.c_double(); fl = C.c_double(); n = C.c_int(); tot = C.c_double()
    self.lib.check(self.lib.dll.odt_profile_read(self.h, C.byref(ms), C.byref(fl), C.byref(n),
                                                 C.byref(tot)))
    return dict(conv_ms=ms.value, conv_flops=fl.value, conv_launches=n.value, total_ms=tot.value)

  def profile_layers(self):
    """[(name, flops, ms, (M, N, K))] for every conv launch of the plan."""
    cnt = C.c_int()
    self.lib.check(self.lib.dll.odt_profile_layer(self.h, -1, None, 0, None, None, None,
                                                  C.byref(cnt)))
    out = []
    for i in range(cnt.value):
      name = C.create_string_buffer(128); fl = C.c_double(); ms = C.c_double()
      mnk = (C.c_int64 * 3)()
      self.lib.check(self.lib.dll.odt_profile_layer(self.h, i, name, 128, C.byref(fl), C.byref(ms),
                                                    C.cast(mnk, c_i64_p), C.byref(cnt)))
      out.append((name.value.decode(), fl.value, ms.value, (mnk[0], mnk[1], mnk[2])))
    return out

  def tap(self, name):
    """Stage tensor in the device layout (NHWC), as numpy."""
    shape = (C.c_int64 * 4)(); rank = C.c_int()
    self.lib.check(self.lib.dll.odt_tap(self.h, name.encode(), None, 0, C.cast(shape, c_i64_p),
                                        C.byref(rank)))
    dims = [int(shape[i]) for i in range(rank.value)]
    out = np.zeros(dims, np.float32)
    self.lib.check(self.lib.dll.odt_tap(self.h, name.encode(), fptr(out), out.size,
                                        C.cast(shape, c_i64_p), C.byref(rank)))
    return out


class _DetectorBase(object):
  graph = ODT_GRAPH_SINGLE

  def __init__(self, config, gpuid=0, weights=None, lib=None):
    self.config = finalize_config(config)
    self.gpuid = gpuid
    self.lib = lib if lib is not None else _lib.get_lib()
    if weights is None:
      path = getattr(config, "load_from", None) if getattr(config, "is_load_from_pb", False) \
          else getattr(config, "model_path", None)
      path = path or getattr(config, "model_path", None)
      if path and str(path).endswith(".pb"):
        # frozen graph (reference --is_load_from_pb, models.py:198-263): Const payloads by name
        weights = load_frozen_pb(path)
      elif path and str(path).endswith(".npz"):
        weights = load_npz(path)
      elif path and (os.path.isdir(str(path)) or os.path.exists(str(path) + ".index") or
                     str(path).endswith(".index")):
        # TF checkpoint directory / prefix (reference obj_detect_tracking.py:404-416), read
        # without TensorFlow
        weights = load_checkpoint(str(path))
      else:
        raise ValueError("weights: pass a {name: array} dict or set config.model_path to a "
                         "Tensorpack-style .npz (reference obj_detect_tracking.py:417-435), a "
                         "TF checkpoint directory / prefix, or a frozen .pb (--is_load_from_pb)")
    unsupported = [f for f in ("use_se", "use_gn", "use_resnext", "use_deformable", "add_relation_nn",
                               "use_conv_frcnn_head", "use_att_frcnn_head",
                               "use_small_object_head", "use_cascade_rcnn")
                   if getattr(self.config, f, False)]
    if unsupported:
      raise NotImplementedError("graph variants not built on this path (SURVEY.md 8, out of scope): "
                                + ", ".join(unsupported))
    # --use_partial_classes (reference models.py:807-829): class-subset head
    self.head_num_class = int(self.config.num_class)
    if getattr(self.config, "use_frcnn_class_agnostic", False):
      weights = expand_class_agnostic_box(weights, self.head_num_class)
    if getattr(self.config, "use_partial_classes", False):
      ids = [self.config.classname2id[name] for name in self.config.partial_classes]
      weights = select_partial_classes(weights, ids, self.head_num_class)
      self.head_num_class = len(ids) + 1
    self.weights = weights
    self._engines = {}
    # the reference's tensor handles / placeholders (models.py:282-283, 965-973)
    self.image = TensorHandle(self, "image")
    self.is_train = TensorHandle(self, "is_train")
    self.final_boxes = TensorHandle(self, "final_boxes")
    self.final_labels = TensorHandle(self, "final_labels")
    self.final_probs = TensorHandle(self, "final_probs")
    self.fpn_box_feat = TensorHandle(self, "fpn_box_feat")
    self.final_valid_indices = TensorHandle(self, "final_valid_indices")
    if getattr(self.config, "add_mask", False):       # reference models.py:959-962
      if self.graph != ODT_GRAPH_SINGLE:
        raise NotImplementedError("--add_mask is built for the single-image graph (Mask_RCNN_FPN)")
      self.final_masks = TensorHandle(self, "final_masks")

  # static plans kept per model: a plan owns its activations, a device copy of the weights and pinned staging
  # (tens of GB at 8 x 1080p), so frames of ever-changing sizes (the reference's image-list drivers) must not
  # accumulate them: least-recently-used plans beyond this many are closed
  max_engines = 8

  def engine(self, batch, height, width, src_hw=None, replica=0, stream_set=False):
    """The static plan for frames of this size.  ``replica`` > 0: a further handle of the same plan (own weights copy, arena and
    streams) -- ``predict_stream`` keeps consecutive frames in flight on them."""
    key = (batch, height, width) if src_hw is None else (batch, height, width) + tuple(src_hw)
    if replica:
      key = key + ("replica", int(replica))
    cfg_e = self.config
    if stream_set:
      # handles of predict_stream: side by side, no side streams of their own (include/odt.h: tail_overlap)
      import copy
      key = key + ("stream",)
      cfg_e = copy.copy(self.config); cfg_e.tail_overlap = False
    e = self._engines.pop(key, None)
    if e is None:
      # evict least-recently-used plans beyond the cap -- but never one with tickets in flight (its pinned results
      # would be destroyed under the caller): those stay until collected, even if that exceeds the cap.  An engine
      # object a caller kept from an earlier engine() call is closed by its eviction; its next use raises OdtError
      # ("engine closed") -- hold at most `max_engines` plan sizes at a time, or raise the cap.
      while len(self._engines) >= max(1, int(self.max_engines)):
        old = next((k for k, v in self._engines.items() if not v._ticket_want), None)
        if old is None:
          break
        self._engines.pop(old).close()
      e = _Engine(self.lib, cfg_e, self.graph, batch, height, width,
                  self.weights, self.gpuid, num_class=self.head_num_class)
      if src_hw is not None:
        e.set_source_size(*src_hw)
    self._engines[key] = e           # (re)insert: dict order = recency
    return e

  def engine_for_raw(self, batch, src_height, src_width):
    """Engine for frames as they come off the decoder: (engine, scale) with the plan sized by
    the reference's rule (get_new_hw, nn.py:1548-1560, short_edge_size / max_size of the config)
    and the resize done on the device; ``scale`` is what the drivers divide boxes by
    (obj_detect_tracking.py:607-608)."""
    neww, newh = get_new_hw(src_height, src_width, self.config.short_edge_size, self.config.max_size)
    scale = (newh * 1.0 / src_height + neww * 1.0 / src_width) / 2.0
    if (newh, neww) == (src_height, src_width):
      return self.engine(batch, newh, neww), scale
    return self.engine(batch, newh, neww, src_hw=(src_height, src_width)), scale

  # reference models.py:1629-1636
  def get_feed_dict_forward(self, imgdata):
    return {self.image: imgdata, self.is_train: False}

  def _fetch(self, fetches, feed_dict):
    raise NotImplementedError

  def close(self):
    for e in self._engines.values():
      e.close()
    self._engines = {}


class Mask_RCNN_FPN(_DetectorBase):
  """b=1 graph (reference models.py:267-973)."""
  graph = ODT_GRAPH_SINGLE

  def predict(self, img, pooled=False):
    """One frame [H,W,3] BGR -> (final_boxes [R,4] f32, final_labels [R] i64,
    final_probs [R] f32, fpn_box_feat [R,256,7,7] f32 (or [R,256] if pooled))."""
    img = np.asarray(img)
    e = self.engine(1, img.shape[0], img.shape[1])
    boxes, labels, probs, valid, feats, pl = e.forward(img[None], want_feats=not pooled,
                                                       want_pooled=pooled)
    r = int(valid[0])
    self.last_masks = e._masks[:r].copy() if e.add_mask else None    # final_masks [R,28,28]
    return (boxes[0, :r].copy(), labels[0, :r].astype(np.int64), probs[0, :r].copy(),
            pl if pooled else feats)

  def predict_stream(self, frames, in_flight=2, pooled=False):
    """Frame-by-frame detection of a video with ``in_flight`` consecutive frames on the GPU at once (round 6): frame t runs on
    handle t mod in_flight, each handle on its own streams, results come back in frame order.  A single 1080p frame leaves most of
    the chip idle most of the time (each of its ~130 launches is latency-bound): 181 -> 220 frames/s with two frames in flight on
    `Mask_RCNN_FPN`.  Two is the robust default; from three on the handles are built without side streams of their own
    (odt_config.tail_overlap = -1) -- 223 / 236 frames/s with three / four in a process that holds no other handles, but 192 / 210
    next to an idle one (streams share hardware queues in creation order: profiles/r06_b1_stream_set.txt); the reference's loop (obj_detect_tracking.py:597-635) sees the same
    per-frame results, one frame later.  Yields what ``predict`` returns."""
    import collections
    n = max(1, int(in_flight))
    pending = collections.deque()

    def finish(item):
      e, t = item
      boxes, labels, probs, valid, feats, pl = e.collect(t)
      r = int(valid[0])
      return (boxes[0, :r].copy(), labels[0, :r].astype(np.int64), probs[0, :r].copy(), pl if pooled else feats)

    for k, img in enumerate(frames):
      img = np.asarray(img)
      if len(pending) == n:
        yield finish(pending.popleft())
      e = self.engine(1, img.shape[0], img.shape[1], replica=k % n, stream_set=n > 2)
      pending.append((e, e.submit(img[None], want_feats=not pooled, want_pooled=pooled)))
    while pending:
      yield finish(pending.popleft())

  def predict_raw(self, frame, pooled=False):
    """Decoder-sized frame [H0,W0,3] (uint8 or float32 BGR): the reference's
    ``resizeImage(frame.astype("float32"), short_edge_size, max_size)`` step
    (obj_detect_tracking.py:597-608) runs on the device.  Returns (boxes, labels, probs, feats,
    scale) with boxes in resized-image coordinates, exactly what ``sess.run`` returns after the
    host-side resize."""
    frame = np.asarray(frame)
    e, scale = self.engine_for_raw(1, frame.shape[0], frame.shape[1])
    boxes, labels, probs, valid, feats, pl = e.forward(frame[None], want_feats=not pooled,
                                                       want_pooled=pooled)
    r = int(valid[0])
    return (boxes[0, :r].copy(), labels[0, :r].astype(np.int64), probs[0, :r].copy(),
            pl if pooled else feats, scale)

  def _fetch(self, fetches, feed_dict):
    boxes, labels, probs, feats = self.predict(feed_dict[self.image])
    table = {"final_boxes": boxes, "final_labels": labels, "final_probs": probs,
             "fpn_box_feat": feats,
             "final_valid_indices": np.asarray([boxes.shape[0]], np.int32),
             "final_masks": self.last_masks}
    return [table[f.name] for f in fetches]


class Mask_RCNN_FPN_multi(_DetectorBase):
  """b=B graph (reference models.py:1969-2408)."""
  graph = ODT_GRAPH_MULTI

  # reference models.py:3301
  def get_feed_dict_forward_multi(self, imgdata_list):
    return {self.image: np.stack(imgdata_list, axis=0), self.is_train: False}

  def predict_batch(self, imgs, pooled=False):
    """[B,H,W,3] -> (final_boxes [B,100,4], final_labels [B,100] f32, final_probs [B,100],
    final_valid_indices [B] i32, fpn_box_feat [M,256,7,7]), M = sum(valid)."""
    imgs = np.asarray(imgs)
    e = self.engine(imgs.shape[0], imgs.shape[1], imgs.shape[2])
    boxes, labels, probs, valid, feats, pl = e.forward(imgs, want_feats=not pooled,
                                                       want_pooled=pooled)
    return boxes, labels.astype(np.float32), probs, valid, (pl if pooled else feats)

  def predict_batch_raw(self, frames, pooled=False):
    """[B,H0,W0,3] decoder-sized frames; device-side resize (see Mask_RCNN_FPN.predict_raw).
    Returns predict_batch's tuple + scale."""
    frames = np.asarray(frames)
    e, scale = self.engine_for_raw(frames.shape[0], frames.shape[1], frames.shape[2])
    boxes, labels, probs, valid, feats, pl = e.forward(frames, want_feats=not pooled,
                                                       want_pooled=pooled)
    return boxes, labels.astype(np.float32), probs, valid, (pl if pooled else feats), scale

  def _fetch(self, fetches, feed_dict):
    boxes, labels, probs, valid, feats = self.predict_batch(feed_dict[self.image])
    table = {"final_boxes": boxes, "final_labels": labels, "final_probs": probs,
             "final_valid_indices": valid, "fpn_box_feat": feats}
    return [table[f.name] for f in fetches]


class Session(object):
  """Shim for the ``tf.Session`` the reference drivers use: ``run(fetches,
  feed_dict)`` triggers ONE forward and returns numpy arrays in fetch order
  (reference obj_detect_tracking.py:632-635)."""

  def __init__(self, config=None):
    pass

  def __enter__(self):
    return self

  def __exit__(self, *a):
    return False

  def run(self, fetches, feed_dict=None):
    """Fetches / feed keys are tensor handles or TF tensor names ("model_0/final_boxes:0", or "final_boxes:0"
    when one frozen model is imported): the reference's frozen route addresses everything by name."""
    single = isinstance(fetches, (TensorHandle, str))
    fl = [fetches] if single else list(fetches)
    fl = [_DEFAULT_GRAPH.get_tensor_by_name(f) if isinstance(f, str) else f for f in fl]
    feed = {(_DEFAULT_GRAPH.get_tensor_by_name(k) if isinstance(k, str) else k): v
            for k, v in (feed_dict or {}).items()}
    model = fl[0].model
    out = model._fetch(fl, feed)
    return out[0] if single else out


def config_from_weights(weights, add_mask=False, is_multi=False, **overrides):
  """Architecture of a frozen model from its tensors (the reference's Mask_RCNN_FPN_frozen needs no config: the
  graph is in the file; here the file is a weight container, so what the graph would say is read off the shapes):
  bottlenecks per stage, classes, FPN / head widths, class-agnostic box head.  Everything that is a graph constant
  rather than a tensor (rpn_test_post_nms_topk, thresholds, frame size) keeps the reference's defaults unless
  overridden."""
  from .config import make_config
  blocks = [0, 0, 0, 0]
  for k in weights:
    parts = k.split("/")
    if len(parts) >= 3 and parts[0].startswith("group") and parts[1].startswith("block") and parts[2] == "conv1":
      g, b = int(parts[0][5:]), int(parts[1][5:])
      blocks[g] = max(blocks[g], b + 1)
  num_class = int(np.asarray(weights["fastrcnn/outputs/class/W"]).shape[-1])
  nbox = int(np.asarray(weights["fastrcnn/outputs/box/W"]).shape[-1])
  kw = dict(resnet_num_block=blocks, num_class=num_class, add_mask=bool(add_mask),
            use_frcnn_class_agnostic=(nbox == 4 or nbox == 8) and num_class > 2,
            im_batch_size=2 if is_multi else 1)
  kw.update(overrides)
  cfg = make_config(**kw)
  cfg.fpn_num_channel = int(np.asarray(weights["fpn/lateral_1x1_c2/W"]).shape[-1])
  cfg.fpn_frcnn_fc_head_dim = int(np.asarray(weights["fastrcnn/fc6/W"]).shape[-1])
  return cfg


class Mask_RCNN_FPN_frozen(object):
  """``Mask_RCNN_FPN_frozen(modelpath, gpuid, add_mask, is_multi)`` (reference models.py:198-263): a frozen ``.pb``
  imported under the name scope ``model_<gpuid>`` whose placeholders and outputs are looked up BY NAME --
  ``model_0/image:0`` in, ``model_0/final_boxes:0``, ``final_labels:0``, ``final_probs:0``, ``fpn_box_feat:0``
  (``final_valid_indices:0`` with is_multi, ``final_masks:0`` with add_mask) out -- and run with
  ``sess.run([...], feed_dict=model.get_feed_dict_forward(img))``.  Same attributes, same feed-dict methods; the
  tensors are also reachable through ``get_default_graph().get_tensor_by_name`` and as plain strings in
  ``Session.run``.  The file is read without TensorFlow (frozen_pb.py); pass the driver's ``config`` for what is
  a graph constant in the file (rpn_test_post_nms_topk ...), else config_from_weights() fills in the reference's
  defaults."""

  def __init__(self, modelpath, gpuid, add_mask=False, is_multi=False, config=None, lib=None):
    self.graph = get_default_graph()
    self.is_multi = is_multi
    self.var_prefix = "model_%s" % gpuid
    weights = load_frozen_pb(modelpath)
    if config is None:
      config = config_from_weights(weights, add_mask=add_mask, is_multi=is_multi)
    cls = Mask_RCNN_FPN_multi if is_multi else Mask_RCNN_FPN
    self._model = cls(config, gpuid=gpuid, weights=weights, lib=lib)
    self.config = self._model.config
    names = ["image", "final_boxes", "final_labels", "final_probs", "fpn_box_feat"]
    if is_multi:
      names.append("final_valid_indices")
    if add_mask:
      names.append("final_masks")
    self.graph._unregister(self.var_prefix)             # re-import under the same scope replaces it
    for n in names:
      h = getattr(self._model, n)
      self.graph._register(self.var_prefix, h)
      setattr(self, n, self.graph.get_tensor_by_name("%s/%s:0" % (self.var_prefix, n)))

  def get_feed_dict_forward(self, imgdata):              # reference models.py:241-255
    if self.is_multi:
      return {self.image: np.stack(imgdata, axis=0)}
    return {self.image: imgdata}

  def get_feed_dict_forward_multi(self, imgs):           # reference models.py:257-264
    return {self.image: np.stack(imgs, axis=0)}

  def predict(self, *a, **k):
    return self._model.predict(*a, **k)

  def predict_batch(self, *a, **k):
    return self._model.predict_batch(*a, **k)

  def engine(self, *a, **k):
    return self._model.engine(*a, **k)

  def close(self):
    self.graph._unregister(self.var_prefix)
    self._model.close()


def get_model(config, gpuid=0, task=0, controller="/cpu:0", is_multi=False, weights=None,
              lib=None):
  """reference models.py:97-119.  Dispatch is the reference's; frozen ``.pb`` files are read
  without TensorFlow; the EfficientDet branch is a "next" row."""
  # is_load_from_pb (reference models.py:102-108 -> Mask_RCNN_FPN_frozen): the frozen file is read
  # as a weight container (frozen_pb.load_frozen_pb), the architecture comes from the config
  if getattr(config, "is_efficientdet", False):        # reference models.py:103-104,112-113
    from .efficientdet import EfficientDet
    return EfficientDet(config, gpuid=gpuid, weights=weights, lib=lib)
  if weights is None and getattr(config, "is_load_from_pb", False):   # reference models.py:102-108
    path = getattr(config, "load_from", None) or getattr(config, "model_path", None)
    return Mask_RCNN_FPN_frozen(path, gpuid, add_mask=getattr(config, "add_mask", False), is_multi=is_multi,
                                config=config, lib=lib)
  cls = Mask_RCNN_FPN_multi if is_multi else Mask_RCNN_FPN
  return cls(config, gpuid=gpuid, weights=weights, lib=lib)


def initialize(config, sess):
  """reference obj_detect_tracking.py:392-448: weights are loaded when the model is
  built, so this is a no-op kept for drop-in compatibility."""
  return None
