"""Drop-in surface: the reference's driver code talks to `get_model()` handles through
`sess.run(fetches, feed_dict)` and hands the results to deep_sort (reference
obj_detect_tracking.py:505-517, :577-695).  These tests drive that exact call pattern.

BASELINE config #1 ("plumbing"): the reference ships no test video and cv2/av are not
installed, so the frame loop is driven by a synthetic in-memory reader exposing the
cv2.VideoCapture methods the loop uses.  When /root/reference is present (build
container) the UNMODIFIED reference Tracker consumes our Detections and our HIP-backed metric.
"""
import os
import sys
import types

import numpy as np
import pytest

from object_detection_tracking_amd._lib import OdtError

from common import small_config, weights_for
from object_detection_tracking_amd import models
from object_detection_tracking_amd.application_util import preprocessing
from object_detection_tracking_amd.config import ACTEV_CLASSES
from object_detection_tracking_amd.deep_sort import (Detection, NearestNeighborDistanceMetric,
                                                     create_obj_infos)
from object_detection_tracking_amd.weights import synthetic_frames

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REF = "/root/reference"


def _reference_tracker_cls():
  if not os.path.isdir(os.path.join(REF, "deep_sort")):
    return None
  np.float = float; np.int = int          # numpy-2 removed the aliases the reference uses
  sys.modules.setdefault("cv2", types.ModuleType("cv2"))
  if REF not in sys.path:
    sys.path.append(REF)
  from deep_sort.tracker import Tracker
  return Tracker


class SyntheticCapture(object):
  """cv2.VideoCapture look-alike over an in-memory frame array."""

  def __init__(self, frames):
    self.frames, self.i = frames, 0

  def isOpened(self):
    return True

  def get(self, prop):
    return float(len(self.frames))

  def read(self):
    if self.i >= len(self.frames):
      return False, None
    f = self.frames[self.i]; self.i += 1
    return True, f


def test_session_shim_single(backend):
  name, lib = backend
  cfg = small_config(resnet_num_block=[1, 1, 1, 1])
  m = models.get_model(cfg, 0, controller="/cpu:0", weights=weights_for(cfg), lib=lib)
  try:
    hw = (64, 96) if name == "emu" else (96, 128)
    img = synthetic_frames(1, *hw)[0].astype("float32")         # the reference feeds float32
    with models.Session() as sess:
      models.initialize(cfg, sess)
      feed = m.get_feed_dict_forward(img)
      assert feed[m.is_train] is False
      boxes, labels, probs, feats = sess.run(
          [m.final_boxes, m.final_labels, m.final_probs, m.fpn_box_feat], feed_dict=feed)
      only = sess.run(m.final_probs, feed_dict=feed)
    assert boxes.dtype == np.float32 and boxes.shape[1] == 4 and boxes.shape[0] <= 100
    assert labels.dtype == np.int64 and probs.dtype == np.float32
    assert feats.shape == (len(boxes), 256, 7, 7) and len(feats) == len(boxes)
    assert np.array_equal(only, probs)
    boxes[:, 2] -= boxes[:, 0]                                   # caller mutates in place
    u8 = m.predict(synthetic_frames(1, *hw)[0])                  # uint8 feed is bit-identical
    assert np.array_equal(u8[2], probs)
    pooled = m.predict(synthetic_frames(1, *hw)[0], pooled=True)[3]
    np.testing.assert_allclose(pooled, feats.mean(axis=(2, 3)), atol=2e-6)
  finally:
    m.close()


def test_session_shim_multi(backend):
  name, lib = backend
  cfg = small_config(resnet_num_block=[1, 1, 1, 1], im_batch_size=2, rpn_test_post_nms_topk=32)
  m = models.get_model(cfg, 0, weights=weights_for(cfg), lib=lib, is_multi=True)
  try:
    fr = synthetic_frames(2, 96, 128)
    feed = m.get_feed_dict_forward_multi([fr[0].astype("float32"), fr[1].astype("float32")])
    sess = models.Session()
    boxes, labels, probs, valid, feats = sess.run(
        [m.final_boxes, m.final_labels, m.final_probs, m.final_valid_indices, m.fpn_box_feat],
        feed_dict=feed)
    assert boxes.shape == (2, 100, 4) and labels.shape == (2, 100) and labels.dtype == np.float32
    assert valid.dtype == np.int32 and feats.shape[0] == valid.sum()   # obj_detect_tracking_multi.py:467
  finally:
    m.close()


def test_plumbing_video_loop_with_tracker(backend):
  """config #1: frame loop (frame_gap) -> sess.run -> create_obj_infos -> tracker NMS ->
  Tracker.predict/update, exactly the call sequence of obj_detect_tracking.py:577-695 -- on the simulator in the CPU
  suite and on the product library under -m gpu (detector, cosine metric, native DeepSORT and TMOT cores all on hip)."""
  emu_lib = backend[1]
  Tracker = _reference_tracker_cls()
  cfg = small_config(resnet_num_block=[1, 1, 1, 1], rpn_test_post_nms_topk=32)
  m = models.get_model(cfg, 0, weights=weights_for(cfg), lib=emu_lib)
  id2class = {i: n for i, n in enumerate(ACTEV_CLASSES)}
  base = synthetic_frames(1, 96, 128)[0]
  video = np.stack([np.roll(base, 2 * i, axis=1) for i in range(9)])
  vcap = SyntheticCapture(video)
  frame_gap, min_conf = 4, 0.02           # random-init weights: use a reachable confidence
  metric = NearestNeighborDistanceMetric("cosine", 0.5, 5, lib=emu_lib)
  tracker = Tracker(metric, max_iou_distance=0.5) if Tracker else None
  # the native trackers on the same detections: DeepSORT core and the TMOT / JDE core
  # (obj_detect_tracking_multi_queuer_tmot.py:543-583: (tlwh, conf, feature) triples)
  from object_detection_tracking_amd.deep_sort import Tracker as NativeTracker
  from object_detection_tracking_amd.tmot import BaseTrack, JDETracker
  native = NativeTracker(NearestNeighborDistanceMetric("cosine", 0.5, 5, lib=emu_lib), max_iou_distance=0.5,
                         lib=emu_lib)
  BaseTrack._count = 0
  jde = JDETracker(min_conf_jde := 0.02, frame_gap=4., lib=emu_lib)
  n_jde = 0
  sess = models.Session()
  cur_frame, n_runs, n_det = 0, 0, 0
  try:
    while cur_frame < int(vcap.get(7)):
      suc, frame = vcap.read()
      assert suc
      if cur_frame % frame_gap != 0:
        cur_frame += 1
        continue
      im = frame.astype("float32")
      scale = 1.0
      boxes, labels, probs, feats = sess.run(
          [m.final_boxes, m.final_labels, m.final_probs, m.fpn_box_feat],
          feed_dict=m.get_feed_dict_forward(im))
      assert len(feats) == len(boxes)
      n_runs += 1
      objs = [id2class[int(l)] for l in labels[:1]]        # class of the top detection
      for obj in objs:
        dets = create_obj_infos(cur_frame, boxes, probs, labels, feats, id2class, [obj], min_conf,
                                0, scale)
        keep = preprocessing.non_max_suppression(np.array([d.tlwh for d in dets]).reshape(-1, 4),
                                                 0.85, np.array([d.confidence for d in dets]))
        dets = [dets[i] for i in keep]
        n_det += len(dets)
        assert all(isinstance(d, Detection) and d.feature.shape == (256,) for d in dets)
        if tracker is not None and obj == objs[0]:
          tracker.predict()
          tracker.update(dets)
        if obj == objs[0]:
          native.predict(); native.update(dets)
          out = jde.update([(d.tlwh.copy(), d.confidence, d.feature.copy()) for d in dets])
          n_jde += len(out)
          if tracker is not None:              # native DeepSORT core == reference tracker on live detections
            assert [t.track_id for t in tracker.tracks] == [t.track_id for t in native.tracks]
      cur_frame += 1
  finally:
    m.close()
  assert n_runs == 3
  assert n_det > 0
  if tracker is not None:
    assert len(tracker.tracks) > 0
  assert len(native.tracks) > 0
  assert n_jde > 0 and all(t.track_id >= 1 for t in jde.tracked_stracks)


def test_reference_tracker_with_hip_metric_reproduces_golden_tracks(emu_lib):
  """The unmodified reference Tracker + our metric class == the reference end to end."""
  Tracker = _reference_tracker_cls()
  if Tracker is None:
    pytest.skip("/root/reference not present (GPU box)")
  g = np.load(os.path.join(G, "deep_sort_ref.npz"))
  tracker = Tracker(NearestNeighborDistanceMetric("cosine", 0.5, budget=5, lib=emu_lib),
                    max_iou_distance=0.5)
  o = t = 0
  for fr, (n, nt) in enumerate(zip(g["seq_n"], g["seq_tracks_n"])):
    dets = [Detection(g["seq_tlwh"][o + i], 0.95, g["seq_feat"][o + i]) for i in range(n)]
    o += n
    tracker.predict(); tracker.update(dets)
    got = np.asarray([[tr.track_id] + list(tr.to_tlwh()) for tr in tracker.tracks
                      if tr.is_confirmed() and tr.time_since_update <= 1]).reshape(-1, 5)
    want = g["seq_tracks"][t:t + nt]; t += nt
    assert got.shape == want.shape, fr
    assert np.array_equal(got[:, 0], want[:, 0]), fr
    np.testing.assert_allclose(got[:, 1:], want[:, 1:], rtol=1e-9, atol=1e-6)


def test_metric_class_bookkeeping():
  """partial_fit budget trimming / active-target pruning like nn_matching.py:137-154."""
  met = NearestNeighborDistanceMetric("cosine", 0.5, budget=2, lib=object())
  f = np.eye(4, dtype=np.float32)
  met.partial_fit(f, [1, 1, 1, 2], [1, 2])
  assert len(met.samples[1]) == 2 and np.array_equal(met.samples[1][-1], f[2])
  met.partial_fit(f[:1], [3], [3])
  assert list(met.samples) == [3]
  assert met.distance(np.zeros((0, 4), np.float32), [3]).shape == (1, 0)
  with pytest.raises(ValueError):
    NearestNeighborDistanceMetric("euclidean", 0.5)


def test_resize_helpers():
  """reference nn.py:1540-1560: target size arithmetic and the identity at native 1080p."""
  from object_detection_tracking_amd.nn import get_new_hw, resizeImage
  assert get_new_hw(1080, 1920, 1080, 1920) == (1920, 1080)
  assert get_new_hw(720, 1280, 1080, 1920) == (1920, 1080)
  assert get_new_hw(1080, 1440, 1080, 1920) == (1440, 1080)
  assert get_new_hw(2160, 3840, 1080, 1920) == (1920, 1080)
  assert get_new_hw(1000, 3000, 1080, 1920) == (1920, 640)      # long edge capped
  im = np.random.default_rng(0).uniform(0, 255, (1080, 1920, 3)).astype("float32")
  assert resizeImage(im, 1080, 1920) is im                       # no copy, like the reference
  small = np.arange(4 * 6 * 3, dtype="float32").reshape(4, 6, 3)
  up = resizeImage(small, 8, 12)
  assert up.shape == (8, 12, 3)
  # linear ramps stay linear under bilinear resampling away from the clamped border
  np.testing.assert_allclose(np.diff(up[2:6, 3, 0]), np.diff(up[2:6, 3, 0])[0], rtol=1e-5)
  np.testing.assert_allclose(up[0, 0], small[0, 0]); np.testing.assert_allclose(up[-1, -1], small[-1, -1])


# ---- frozen .pb route (reference --is_load_from_pb, models.py:102-108,198-263) -----------------
def test_frozen_pb_reader_roundtrip(tmp_path):
  from object_detection_tracking_amd.frozen_pb import load_frozen_pb, write_frozen_pb
  rng = np.random.default_rng(3)
  w = {"conv0/W": rng.standard_normal((7, 7, 3, 64)).astype(np.float32),
       "conv0/bn/gamma": rng.standard_normal((64,)).astype(np.float32),
       "conv0/bn/mean/EMA": rng.standard_normal((64,)).astype(np.float32),
       "fastrcnn/fc6/W": rng.standard_normal((96, 40)).astype(np.float32),
       "fpn/lateral_1x1_c2/b": np.zeros((16,), np.float32)}
  path = str(tmp_path / "frozen.pb")
  write_frozen_pb(path, w, float_val_names=("conv0/bn/gamma",), half_names=("fastrcnn/fc6/W",))
  got = load_frozen_pb(path)
  assert set(got) == set(w)                       # int / scalar Consts, Identity, Placeholder skipped
  for k in w:
    want = w[k].astype(np.float16).astype(np.float32) if k == "fastrcnn/fc6/W" else w[k]
    assert got[k].dtype == np.float32 and np.array_equal(got[k], want), k
  with pytest.raises(ValueError):
    open(path, "wb").write(b"\x0a\x02\x0a\x00")    # a GraphDef without any Const
    load_frozen_pb(path)


def test_get_model_from_frozen_pb_and_checkpoint_dir(emu_lib, tmp_path):
  """obj_detect_tracking.py --is_load_from_pb --model_path x.pb and --model_path <checkpoint dir>:
  same detections as the same weights passed directly."""
  from object_detection_tracking_amd.frozen_pb import write_frozen_pb
  from object_detection_tracking_amd.tf_checkpoint import write_checkpoint
  cfg = small_config(resnet_num_block=[1, 1, 1, 1])
  w = weights_for(cfg)
  pb = str(tmp_path / "obj_v3.pb")
  write_frozen_pb(pb, w)
  ck = tmp_path / "obj_v3_model"; ck.mkdir()
  write_checkpoint(str(ck / "model-77"), w)
  fr = synthetic_frames(1, 64, 96)[0]
  m0 = models.get_model(cfg, 0, weights=w, lib=emu_lib)
  want = m0.predict(fr); m0.close()
  for kw in (dict(is_load_from_pb=True, model_path=pb, load_from=pb), dict(model_path=str(ck))):
    m1 = models.get_model(small_config(resnet_num_block=[1, 1, 1, 1], **kw), 0, lib=emu_lib)
    try:
      for a_, b_ in zip(want, m1.predict(fr)):
        assert np.array_equal(a_, b_)
    finally:
      m1.close()


def test_frozen_model_named_tensor_contract(emu_lib, tmp_path):
  """Mask_RCNN_FPN_frozen (reference models.py:198-263): the frozen file is imported under ``model_<gpuid>`` and
  every placeholder / output is addressed BY NAME -- get_tensor_by_name("model_0/final_boxes:0"), sess.run with
  handles or names, feed through get_feed_dict_forward[_multi]."""
  from object_detection_tracking_amd.frozen_pb import write_frozen_pb
  cfg = small_config(resnet_num_block=[1, 1, 1, 1])
  w = weights_for(cfg)
  pb = str(tmp_path / "obj_v3.pb")
  write_frozen_pb(pb, w)
  fr = synthetic_frames(2, 64, 96)
  m0 = models.get_model(cfg, 0, weights=w, lib=emu_lib)
  want = m0.predict(fr[0]); m0.close()
  m = models.Mask_RCNN_FPN_frozen(pb, 0, add_mask=False, is_multi=False, config=cfg, lib=emu_lib)
  try:
    g = models.get_default_graph()
    assert m.var_prefix == "model_0" and m.image is g.get_tensor_by_name("model_0/image:0")
    assert m.final_boxes is g.get_tensor_by_name("model_0/final_boxes:0")
    with pytest.raises(KeyError):
      g.get_tensor_by_name("model_0/final_valid_indices:0")        # single-image import does not carry it
    sess = models.Session()
    got = sess.run([m.final_boxes, m.final_labels, m.final_probs, m.fpn_box_feat],
                   feed_dict=m.get_feed_dict_forward(fr[0]))
    for a_, b_ in zip(want, got):
      assert np.array_equal(a_, b_)
    # by name, scoped and (one imported model) unscoped; a single fetch returns the array itself
    got = sess.run(["model_0/final_boxes:0", "final_probs:0"], feed_dict={"model_0/image:0": fr[0]})
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[2])
    assert np.array_equal(sess.run("final_labels:0", feed_dict={"image:0": fr[0]}), want[1])
  finally:
    m.close()
  with pytest.raises(KeyError):
    models.get_default_graph().get_tensor_by_name("model_0/final_boxes:0")    # closed models leave the graph
  # the architecture read off the file when no config is given (reference: the graph is in the .pb)
  c2 = models.config_from_weights(w)
  assert list(c2.resnet_num_block) == [1, 1, 1, 1] and c2.num_class == cfg.num_class
  # batched import: model_0/final_valid_indices:0 exists, feed through get_feed_dict_forward_multi
  cfgm = small_config(resnet_num_block=[1, 1, 1, 1], im_batch_size=2, rpn_test_post_nms_topk=32)
  mm = models.Mask_RCNN_FPN_frozen(pb, 0, is_multi=True, config=cfgm, lib=emu_lib)
  try:
    boxes, valid = models.Session().run(["model_0/final_boxes:0", "model_0/final_valid_indices:0"],
                                        feed_dict=mm.get_feed_dict_forward_multi([fr[0], fr[1]]))
    assert boxes.shape == (2, cfgm.result_per_im, 4) and valid.shape == (2,) and valid.dtype == np.int32
  finally:
    mm.close()


def test_engine_cache_is_bounded(emu_lib):
  """Frames of ever-changing sizes must not accumulate static plans (each owns activations + a weight copy)."""
  cfg = small_config(resnet_num_block=[1, 1, 1, 1], rpn_test_post_nms_topk=16)
  m = models.get_model(cfg, 0, weights=weights_for(cfg), lib=emu_lib)
  try:
    m.max_engines = 2
    e1 = m.engine(1, 64, 96); e2 = m.engine(1, 64, 128)
    assert m.engine(1, 64, 96) is e1                       # hit: refreshed
    e3 = m.engine(1, 96, 96)                               # evicts the least recently used: e2
    assert len(m._engines) == 2 and e2.h is None and e1.h is not None and e3.h is not None
    # an evicted engine a caller still holds fails loudly, not with a null handle inside the library
    with pytest.raises(OdtError, match="engine closed"):
      e2.forward(synthetic_frames(1, 64, 128))
    # an engine with a ticket in flight is never evicted (its pinned results were destroyed under the caller)
    fr = synthetic_frames(1, 64, 96)
    want = e1.forward(fr, want_feats=False, want_pooled=True)
    t = e1.submit(fr, want_feats=False, want_pooled=True)
    m.engine(1, 96, 96)                                    # e3 most recent, e1 least recent -- but in flight
    e4 = m.engine(1, 64, 160)                              # evicts e3 instead
    assert e1.h is not None and e3.h is None and e4.h is not None
    got = e1.collect(t)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[5], want[5])
    m.engine(1, 64, 192)                                   # nothing in flight any more: plain LRU again
    assert len(m._engines) == 2
  finally:
    m.close()


# ---- TF checkpoint route (reference obj_detect_tracking.py:404-416; obj_v3_model.tgz is one) ----
def test_tf_checkpoint_reader_roundtrip(tmp_path):
  from object_detection_tracking_amd.tf_checkpoint import load_checkpoint, read_index, write_checkpoint
  rng = np.random.default_rng(4)
  v = {"group%d/block%d/conv%d/W" % (g, b, c): rng.standard_normal((1, 1, 8, 4)).astype(np.float32)
       for g in range(3) for b in range(3) for c in (1, 2, 3)}          # shared prefixes, many blocks
  v["conv0/bn/variance/EMA"] = rng.uniform(0.5, 1.5, (64,)).astype(np.float32)
  v["fastrcnn/fc6/W"] = rng.standard_normal((96, 40)).astype(np.float16)
  v["global_step"] = np.asarray(1234, np.int64)
  v["conv0/W/Momentum"] = np.zeros((7, 7, 3, 64), np.float32)
  prefix = str(tmp_path / "model-1234")
  write_checkpoint(prefix, v)
  assert set(read_index(prefix + ".index")) == set(v)
  for where in (prefix, prefix + ".index", str(tmp_path)):              # prefix, index file, directory
    got = load_checkpoint(where)
    assert set(got) == {k for k in v if k not in ("global_step", "conv0/W/Momentum")}
    for k, a in got.items():
      assert a.dtype == np.float32 and np.array_equal(a, v[k].astype(np.float32)), k
  open(prefix + ".index", "ab").write(b"x")
  with pytest.raises(ValueError):
    load_checkpoint(prefix)


def test_frozen_pb_written_by_the_protobuf_runtime(emu_lib, tmp_path):
  """The .pb reader on a GraphDef serialised by Google's protobuf runtime (tests/tf_protos.py declares TensorFlow's
  messages from their published field numbers) instead of this repository's own writer: Const nodes in every payload
  form TF uses (tensor_content, packed float_val / double_val / half_val, a splat constant), between nodes a real
  frozen graph also holds (Placeholder, Conv2D with list / string / shape attributes, an int32 Const) -- and the
  same file through get_model(load_from_pb): outputs identical to the .npz route."""
  import tf_protos as T
  from object_detection_tracking_amd.frozen_pb import load_frozen_pb
  cfg = small_config(resnet_num_block=[1, 1, 1, 1])
  w = weights_for(cfg)
  g = T.MESSAGES["GraphDef"]()
  g.versions.producer = 134; g.versions.min_consumer = 12
  ph = g.node.add(); ph.name = "image"; ph.op = "Placeholder"
  ph.attr["dtype"].type = T.DT_FLOAT
  ph.attr["shape"].shape.dim.add().size = -1
  ph.attr["shape"].shape.dim.add().size = -1
  ph.attr["shape"].shape.dim.add().size = 3
  ic = g.node.add(); ic.name = "some/int/constant"; ic.op = "Const"
  ic.attr["dtype"].type = T.DT_INT32
  ic.attr["value"].tensor.dtype = T.DT_INT32
  ic.attr["value"].tensor.tensor_shape.dim.add().size = 2
  ic.attr["value"].tensor.int_val.extend([3, 4])
  names = sorted(w)
  forms = {}
  for i, k in enumerate(names):
    a = w[k]
    if k == "conv0/bn/gamma":
      T.const_node(g, k, a, T.DT_FLOAT, "vals"); forms[k] = "float_val"
    elif k == "fastrcnn/fc6/W":
      T.const_node(g, k, a.astype(np.float16), T.DT_HALF, "vals"); forms[k] = "half_val"
    elif k == "fastrcnn/fc7/W":
      T.const_node(g, k, a.astype(np.float16), T.DT_HALF, "content"); forms[k] = "half content"
    elif k == "rpn/conv0/b":
      T.const_node(g, k, a.astype(np.float64), T.DT_DOUBLE, "vals"); forms[k] = "double_val"
    else:
      T.const_node(g, k, a, T.DT_FLOAT, "content")
    if i == 3:                                  # an op node in the middle, with the attribute kinds Conv2D carries
      cv = g.node.add(); cv.name = "conv0/Conv2D"; cv.op = "Conv2D"
      cv.input.extend(["image", "conv0/W"]); cv.device = "/device:GPU:0"
      cv.attr["T"].type = T.DT_FLOAT
      cv.attr["strides"].list.i.extend([1, 1, 2, 2])
      cv.attr["padding"].s = b"VALID"
      cv.attr["data_format"].s = b"NCHW"
      cv.attr["use_cudnn_on_gpu"].b = True
  splat = np.full((5, 7), 0.25, np.float32)
  T.const_node(g, "some/splat", splat, T.DT_FLOAT, "splat")
  path = str(tmp_path / "by_protobuf.pb")
  with open(path, "wb") as fh:
    fh.write(g.SerializeToString())
  got = load_frozen_pb(path)
  assert set(got) == set(names) | {"some/splat"}                     # the int32 Const and the op nodes are not weights
  assert np.array_equal(got["some/splat"], splat)
  for k in names:
    want = w[k]
    if forms.get(k, "").startswith("half"):
      want = want.astype(np.float16).astype(np.float32)
    assert got[k].dtype == np.float32 and got[k].shape == want.shape, k
    assert np.array_equal(got[k], want), (k, forms.get(k))
  # ... and as a model: the half-precision entries replaced by exact ones so that both routes see the same numbers
  g2 = T.MESSAGES["GraphDef"]()
  for k in names:
    T.const_node(g2, k, w[k], T.DT_FLOAT, "vals" if k.endswith("/b") else "content")
  path2 = str(tmp_path / "model_by_protobuf.pb")
  with open(path2, "wb") as fh:
    fh.write(g2.SerializeToString())
  fr = synthetic_frames(1, 64, 96)[0]
  cfg_pb = small_config(resnet_num_block=[1, 1, 1, 1], is_load_from_pb=True, model_path=path2, load_from=path2)
  m1 = models.get_model(cfg, 0, weights=w, lib=emu_lib)
  m2 = models.get_model(cfg_pb, 0, lib=emu_lib)
  try:
    o1, o2 = m1.predict(fr), m2.predict(fr)
    for a, b in zip(o1, o2):
      assert np.array_equal(a, b)
    assert len(o1[0]) > 0
  finally:
    m1.close(); m2.close()


def test_checkpoint_index_values_by_the_protobuf_runtime(tmp_path):
  """V2 checkpoint whose .index VALUES (BundleHeaderProto / BundleEntryProto) come from Google's protobuf runtime
  (tests/tf_protos.py) -- offset 0, shard 0 and empty shapes are then omitted as the official encoder omits defaults,
  crc32c is a fixed32 -- inside this repository's table writer (prefix-compressed keys, several blocks)."""
  import tf_protos as T
  from object_detection_tracking_amd.tf_checkpoint import load_checkpoint, write_checkpoint
  rng = np.random.default_rng(9)
  names = ["group0/block0/conv1/W", "group0/block0/conv1/bn/gamma", "group0/block0/conv1/bn/beta", "group0/block0/conv2/W",
           "conv0/W", "fastrcnn/fc6/W", "fastrcnn/fc6/b", "fpn/lateral_1x1_c2/W", "scalar_like"]
  w = {k: rng.standard_normal((3, 4, 5)[:1 + i % 3]).astype(np.float32) for i, k in enumerate(names)}
  w["scalar_like"] = np.asarray(rng.standard_normal((1,)), np.float32)
  w["global_step"] = np.asarray([77], np.int64)                       # skipped by name
  def entry(dt, shape, offset, size):
    e = T.MESSAGES["BundleEntryProto"]()
    e.dtype = dt; e.offset = offset; e.size = size; e.crc32c = 0x9a3b5c01
    for d in shape:
      e.shape.dim.add().size = d
    return e.SerializeToString()
  def header():
    h = T.MESSAGES["BundleHeaderProto"]()
    h.num_shards = 1; h.version.producer = 1
    return h.SerializeToString()
  ck = tmp_path / "ck"; ck.mkdir()
  write_checkpoint(str(ck / "model-5"), w, per_block=3, encode_entry=entry, encode_header=header)
  got = load_checkpoint(str(ck))
  assert set(got) == set(names)
  for k in names:
    assert got[k].dtype == np.float32 and np.array_equal(got[k], w[k]), k


@pytest.mark.gpu
def test_frozen_pb_to_hip_forward_vs_oracle(hip_lib, tmp_path):
  """The frozen route end to end on the product library (it ran on the simulator only): weights written
  as a frozen GraphDef, read back by frozen_pb.py, architecture taken from the file, forward on the MI355X through the
  named-tensor surface -- against the oracle on the ORIGINAL weights (so a reader that dropped or permuted a tensor
  shows up as a parity failure, not as self-consistency)."""
  from common import match_detections
  from object_detection_tracking_amd.frozen_pb import write_frozen_pb
  from oracle.graph import OracleModel
  cfg = small_config(resnet_num_block=[1, 2, 2, 1], rpn_test_post_nms_topk=100, max_size=448, short_edge_size=256)
  w = weights_for(cfg)
  pb = str(tmp_path / "obj_v3_frozen.pb")
  write_frozen_pb(pb, w)
  fr = synthetic_frames(1, 256, 448, seed=11)[0]
  ref = OracleModel(cfg, w).forward(fr)
  m = models.Mask_RCNN_FPN_frozen(pb, 0, add_mask=False, is_multi=False, config=cfg, lib=hip_lib)
  try:
    boxes, labels, probs, feats = models.Session().run(
        ["model_0/final_boxes:0", "model_0/final_labels:0", "model_0/final_probs:0", "model_0/fpn_box_feat:0"],
        feed_dict=m.get_feed_dict_forward(fr))
    assert m.engine(1, 256, 448).describe()["memory"]["keep_taps"] == 0            # the production (arena) handle
    miss, extra = match_detections(boxes, labels, probs, ref["final_boxes"], ref["final_labels"], ref["final_probs"], 1e-3, 1e-4)
    assert miss == 0 and extra == 0 and len(boxes) > 3, (miss, extra, len(boxes))
    assert feats.shape == (len(boxes), 256, 7, 7)
  finally:
    m.close()
  # and with the architecture read off the file (no config: the reference's Mask_RCNN_FPN_frozen takes none)
  m2 = models.Mask_RCNN_FPN_frozen(pb, 0, lib=hip_lib)
  try:
    assert list(m2.config.resnet_num_block) == [1, 2, 2, 1]
  finally:
    m2.close()


@pytest.mark.gpu
def test_product_default_engine_on_the_drop_in_and_ingest_paths(hip_lib):
  """The suite-wide make_config (tests/common.py) pins conv_split_family = 0; this runs the surfaces a user touches -- the
  session shim on both graphs, predict, the pipelined ingest with two streams' worth of tickets -- on the PRODUCT default
  (conv_split_family = "auto": guarded, handle swap machinery armed, continuous watch) and requires the unguarded handle's
  results bit for bit (ordinary weights: the guard stays on fp16x2)."""
  import copy
  base = small_config(resnet_num_block=[1, 1, 2, 3], im_batch_size=2, rpn_test_post_nms_topk=64)
  w = weights_for(base)
  fr = synthetic_frames(2, 160, 224, seed=21)
  out = {}
  for fam in (0, "auto"):
    cfg = copy.copy(base); cfg.conv_split_family = fam
    m1 = models.get_model(cfg, 0, weights=w, lib=hip_lib)
    mm = models.get_model(cfg, 0, weights=w, lib=hip_lib, is_multi=True)
    try:
      with models.Session() as sess:
        single = sess.run([m1.final_boxes, m1.final_labels, m1.final_probs, m1.fpn_box_feat], feed_dict=m1.get_feed_dict_forward(fr[0].astype("float32")))
        multi = sess.run([mm.final_boxes, mm.final_labels, mm.final_probs, mm.final_valid_indices, mm.fpn_box_feat],
                         feed_dict=mm.get_feed_dict_forward_multi([fr[0].astype("float32"), fr[1].astype("float32")]))
      pooled = m1.predict(fr[1], pooled=True)
      e = mm.engine(2, 160, 224)
      stream = list(e.forward_stream([fr, fr[::-1].copy(), fr]))
      d = e.describe()
      if fam == "auto":
        a = d["conv_split_family_auto"]
        assert a["chosen"].startswith("fp16x2") and a["calibration_forwards_left"] == 0 and a["watch"] is not None, a
      else:
        assert d["range_guard"].startswith("off"), d
      out[fam] = (single, multi, pooled, stream)
    finally:
      m1.close(); mm.close()
  def same(a, b):
    if isinstance(a, (list, tuple)):
      assert len(a) == len(b)
      for x, y in zip(a, b):
        same(x, y)
    else:
      assert (a is None and b is None) or np.array_equal(a, b)
  same(out[0], out["auto"])


def test_predict_stream_keeps_frames_in_flight_on_replica_handles(backend):
  """models.predict_stream (round 6): consecutive frames on replica handles (frame t on handle t mod in_flight), results in frame
  order and equal to predict() frame by frame; the replicas are separate handles of the same plan."""
  name, lib = backend
  cfg = small_config(resnet_num_block=[1, 1, 1, 1], rpn_test_post_nms_topk=32)
  m = models.get_model(cfg, 0, weights=weights_for(cfg), lib=lib)
  try:
    hw = (64, 96) if name == "emu" else (96, 128)
    frames = [synthetic_frames(1, *hw, seed=30 + i)[0] for i in range(5)]
    want = [m.predict(f, pooled=True) for f in frames]
    for n in (2, 3):
      got = list(m.predict_stream(frames, in_flight=n, pooled=True))
      assert len(got) == len(frames)
      for g, w in zip(got, want):
        for a, b in zip(g, w):
          assert np.array_equal(a, b)
    e0, e1 = m.engine(1, *hw), m.engine(1, *hw, replica=1)
    assert e0 is not e1 and e0.h != e1.h and m.engine(1, *hw, replica=1) is e1
  finally:
    m.close()
