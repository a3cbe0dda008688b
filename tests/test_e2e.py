"""End-to-end parity of the whole forward (libodt via the reference's model surface) against
the oracle, stage by stage through the debug taps, for both reference graphs.

Tolerances: the network is ~105 sequential fp32 conv/FC layers accumulated in a different order
than torch/TF, so stage tensors are compared with a relative tolerance on their own scale and
boxes with an ABSOLUTE 1e-3 px at every size up to 1920 x 1080 (the north_star's bound; measured worst case
4.9e-4 px at 8 x 1080p, profiles/r03_parity.json).  Selection stages (top-k / NMS / level assignment) are discontinuous, so final
detections are compared as matched sets with a reported mismatch count; the bit-exact index
tests on identical inputs live in test_ops.py.
"""
import numpy as np
import pytest

from common import assert_same_detections, match_detections, small_config, weights_for
from object_detection_tracking_amd import models
from common import make_config
from object_detection_tracking_amd.weights import synthetic_frames
from oracle.graph import OracleModel


def _rel(a, b):
  return float(np.abs(a - b).max() / max(1e-6, np.abs(b).max()))


def _with_taps(cfg):
  import copy
  c = copy.copy(cfg)
  c.keep_taps = True
  return c


def _check_trunk(e, ref, tol):
  for name in ["conv0", "pool0", "c2", "c3", "c4", "c5"]:
    assert _rel(e.tap(name).transpose(0, 3, 1, 2), ref[name]) < tol, name
  for l in range(2, 7):
    r = ref["p%d" % l]
    t = e.tap("p%d" % l).transpose(0, 3, 1, 2)[:, :, :r.shape[2], :r.shape[3]]
    assert _rel(t, r) < tol, "p%d" % l
    rp = e.tap("rpn%d" % l)
    assert _rel(rp[..., :3], ref["rpn_logits%d" % l]) < tol
    assert _rel(rp[..., 3:15].reshape(rp.shape[:3] + (3, 4)), ref["rpn_deltas%d" % l]) < tol


def _run_single(lib, cfg, H, W, tol=2e-5, box_tol=None, budget=0):
  w = weights_for(cfg)
  fr = synthetic_frames(1, H, W)
  ref = OracleModel(cfg, w).forward(fr[0])
  # the production handle (activations in the liveness-planned arena) gives the outputs; a debug handle with
  # keep_taps (a dedicated buffer per stage tensor) gives the stage taps -- and must agree with it bit for bit
  m0 = models.get_model(cfg, 0, weights=w, lib=lib)
  try:
    prod = m0.predict(fr[0])
    assert m0.engine(1, H, W).describe()["memory"]["keep_taps"] == 0
    with pytest.raises(Exception, match="keep_taps"):
      m0.engine(1, H, W).tap("c3")
  finally:
    m0.close()
  m = models.get_model(_with_taps(cfg), 0, weights=w, lib=lib)
  try:
    boxes, labels, probs, feats = m.predict(fr[0])
    for a, b in zip(prod, (boxes, labels, probs, feats)):
      assert np.array_equal(a, b), "arena and keep_taps handles disagree"
    e = m.engine(1, H, W)
    _check_trunk(e, ref, tol)
    box_tol = box_tol or 1e-3
    assert boxes.dtype == np.float32 and labels.dtype == np.int64 and probs.dtype == np.float32
    assert feats.shape == (boxes.shape[0], 256, 7, 7)
    n = int(e.tap("nproposals")[0])
    assert n == ref["proposals"].shape[0]
    pm, rm = match_detections(e.tap("proposals")[0, 0, :n], np.zeros(n), np.zeros(n),
                              ref["proposals"], np.zeros(n), np.zeros(n), box_tol, 1)
    # mismatch budget: 0 -- every full-size configuration measured 0 unmatched proposals / detections on the MI355X in
    # both arithmetic modes (profiles/r02_parity.json)
    assert pm + rm <= budget, "proposal sets differ: %d/%d of %d" % (pm, rm, n)
    miss, extra = match_detections(boxes, labels, probs, ref["final_boxes"], ref["final_labels"],
                                   ref["final_probs"], box_tol, 1e-4)
    assert miss + extra <= budget, (miss, extra)
    if miss + extra == 0:
      # not only as sets: pair by pair boxes / scores / appearance features, and the order up to score ties
      assert_same_detections(boxes, labels, probs, feats, ref["final_boxes"], ref["final_labels"], ref["final_probs"],
                             ref["fpn_box_feat"], box_tol, 1e-4, 10 * tol)
    return miss, extra
  finally:
    m.close()


def test_forward_single_small(backend):
  name, lib = backend
  cfg = small_config(resnet_num_block=[1, 1, 1, 1] if name == "emu" else [1, 1, 2, 3])
  miss, extra = _run_single(lib, cfg, 96, 128)
  assert miss == 0 and extra == 0


def _run_multi(lib, cfg, B, H, W, tol=2e-5, w=None, info=None, budget=0):
  w = weights_for(cfg) if w is None else w
  fr = synthetic_frames(B, H, W)
  ref = OracleModel(cfg, w).forward_multi(fr)
  m0 = models.get_model(cfg, 0, weights=w, lib=lib, is_multi=True)      # production handle: arena
  try:
    prod = m0.predict_batch(fr)
  finally:
    m0.close()
  m = models.get_model(_with_taps(cfg), 0, weights=w, lib=lib, is_multi=True)
  try:
    boxes, labels, probs, valid, feats = m.predict_batch(fr)
    for a, b in zip(prod, (boxes, labels, probs, valid, feats)):
      assert np.array_equal(a, b), "arena and keep_taps handles disagree"
    e = m.engine(B, H, W)
    _check_trunk(e, ref, tol)
    assert labels.dtype == np.float32 and valid.dtype == np.int32
    assert boxes.shape == (B, cfg.result_per_im, 4)
    assert np.array_equal(valid, ref["final_valid_indices"])
    assert feats.shape[0] == valid.sum()
    if info is not None:
      info["nproposals"] = e.tap("nproposals").reshape(-1).copy()
    box_tol = 1e-3
    tot = 0
    for b in range(B):
      v = valid[b]
      miss, extra = match_detections(boxes[b, :v], labels[b, :v], probs[b, :v],
                                     ref["final_boxes"][b, :v], ref["final_labels"][b, :v],
                                     ref["final_probs"][b, :v], box_tol, 1e-4)
      tot += miss + extra
    assert tot <= budget, tot
    if tot == 0:
      off = 0
      for b in range(B):
        v = int(valid[b])
        assert_same_detections(boxes[b, :v], labels[b, :v], probs[b, :v], feats[off:off + v],
                               ref["final_boxes"][b, :v], ref["final_labels"][b, :v], ref["final_probs"][b, :v],
                               ref["fpn_box_feat"][off:off + v], box_tol, 1e-4, 10 * tol)
        off += v
    return tot
  finally:
    m.close()


def test_forward_multi_small(backend):
  name, lib = backend
  cfg = small_config(resnet_num_block=[1, 1, 1, 1], im_batch_size=2, rpn_test_post_nms_topk=48)
  assert _run_multi(lib, cfg, 2, 96, 128) == 0


def test_forward_multi_fewer_proposals_than_k(backend):
  """Trained RPNs score most anchors negative, so an image keeps fewer than K proposals (the
  zero-padded NMS slots outrank negative logits and are dropped by the area > 0 filter,
  models.py:2487-2520).  The box head's rows must stay at b * K + j for the images behind such an
  image (a packed ROIAlign output shifted them: ADVICE round 1)."""
  name, lib = backend
  cfg = small_config(resnet_num_block=[1, 1, 1, 1], im_batch_size=2, rpn_test_post_nms_topk=48)
  w = dict(weights_for(cfg))
  w["rpn/class/b"] = (w["rpn/class/b"] - 3.0).astype(np.float32)
  info = {}
  assert _run_multi(lib, cfg, 2, 96, 128, w=w, info=info) == 0
  n = info["nproposals"]
  assert 0 < n[0] < 48 and 0 < n[1], n     # the case under test: a short first image, a live second one


@pytest.mark.gpu
def test_forward_single_r101_256x448(hip_lib):
  cfg = make_config(rpn_test_post_nms_topk=300, max_size=448, short_edge_size=256)
  _run_single(hip_lib, cfg, 256, 448)


@pytest.mark.gpu
def test_forward_single_r101_odd_size(hip_lib):
  """Non multiple-of-32 frame: exercises pad-to-32, sliced P2..P4 and anchor slicing."""
  cfg = make_config(rpn_test_post_nms_topk=200, max_size=400, short_edge_size=230)
  _run_single(hip_lib, cfg, 230, 394)


@pytest.mark.gpu
def test_forward_single_r101_k1000(hip_lib):
  """The reference script's default rpn_test_post_nms_topk = 1000 (obj_detect_tracking.py:132)."""
  cfg = make_config(rpn_test_post_nms_topk=1000, max_size=640, short_edge_size=384)
  _run_single(hip_lib, cfg, 384, 640)


@pytest.mark.gpu
def test_forward_single_r101_1080p_k1000(hip_lib):
  """BASELINE config #2's frame with the script's own default K = 1000 (obj_detect_tracking.py:132): 1000 RoIs through
  ROIAlign / box head / per-class NMS at 1920x1080."""
  cfg = make_config(rpn_test_post_nms_topk=1000)
  _run_single(hip_lib, cfg, 1080, 1920, tol=1e-5)


@pytest.mark.gpu
def test_forward_single_r101_k2000(hip_lib):
  """--rpn_test_post_nms_topk above 1024 (the reference takes any value): the selection kernels' K > 1024 path --
  top-k in rounds of 1024, paneled NMS walk (csrc/select_device.hpp block_nms_paneled), per-class NMS over 2000 RoIs."""
  cfg = make_config(rpn_test_post_nms_topk=2000, max_size=640, short_edge_size=384)
  _run_single(hip_lib, cfg, 384, 640)


@pytest.mark.gpu
def test_forward_multi_r101_b2_k2000(hip_lib):
  cfg = make_config(rpn_test_post_nms_topk=2000, max_size=448, short_edge_size=256, im_batch_size=2)
  _run_multi(hip_lib, cfg, 2, 256, 448)


def test_forward_k_above_1024_small(backend):
  """K = 1100 on a small frame (P2 alone has 2304 anchors): the K > 1024 selection kernels end to end -- the single-image
  graph with 4 classes on the simulator (1024 fibres per workgroup are slow there), both graphs with 15 on the GPU."""
  name, lib = backend
  kw = dict(num_class=4) if name == "emu" else {}
  cfg = small_config(resnet_num_block=[1, 1, 1, 1], rpn_test_post_nms_topk=1100, **kw)
  miss, extra = _run_single(lib, cfg, 96, 128)
  assert miss == 0 and extra == 0
  if name == "hip":
    cfg = small_config(resnet_num_block=[1, 1, 1, 1], im_batch_size=2, rpn_test_post_nms_topk=1100)
    assert _run_multi(lib, cfg, 2, 96, 128) == 0


@pytest.mark.gpu
def test_forward_multi_r101_b2_256x448(hip_lib):
  cfg = make_config(rpn_test_post_nms_topk=300, max_size=448, short_edge_size=256,
                    im_batch_size=2)
  _run_multi(hip_lib, cfg, 2, 256, 448)


@pytest.mark.gpu
def test_forward_single_r101_1080p(hip_lib):
  """BASELINE config #2: 1920x1080, b=1, K=300."""
  cfg = make_config(rpn_test_post_nms_topk=300)
  _run_single(hip_lib, cfg, 1080, 1920, tol=1e-5)


@pytest.mark.gpu
def test_forward_multi_r101_b8_1080p(hip_lib):
  """BASELINE config #3, the benchmark workload itself: 8 frames of 1920x1080 through the
  batched graph, every stage tap and the final detections against the oracle."""
  cfg = make_config(rpn_test_post_nms_topk=300, im_batch_size=8)
  _run_multi(hip_lib, cfg, 8, 1080, 1920, tol=1e-5)


@pytest.mark.gpu
def test_forward_r50_coco_partial_1280x720(hip_lib):
  """The README's released frozen model obj_coco_resnet50_partial_*_1280x720_rpn300: ResNet-50
  FPN (blocks 3,4,6,3), version 2 (no dilation), 81 COCO classes reduced to a class subset."""
  names = ["BG"] + ["c%d" % i for i in range(1, 81)]
  part = ["c1", "c2", "c3", "c4", "c6", "c8", "c25", "c27"]
  cfg = make_config(rpn_test_post_nms_topk=300, version=2, use_dilations=False, num_class=81,
                    resnet_num_block=[3, 4, 6, 3], is_coco_model=True, use_partial_classes=True,
                    partial_classes=part, classname2id={n: i for i, n in enumerate(names)},
                    max_size=1280, short_edge_size=720)
  _run_single(hip_lib, cfg, 720, 1280, tol=5e-5)


@pytest.mark.gpu
def test_determinism_and_size_independent_properties(hip_lib):
  """Full-size properties that need no oracle: run-to-run bit-identical outputs; boxes inside
  the frame; probs sorted within (0,1]; labels in range; features finite."""
  cfg = make_config(rpn_test_post_nms_topk=300, im_batch_size=2)
  w = weights_for(cfg)
  fr = synthetic_frames(2, 1080, 1920, seed=7)
  m = models.get_model(cfg, 0, weights=w, lib=hip_lib, is_multi=True)
  try:
    a = m.predict_batch(fr)
    b = m.predict_batch(fr)
    for x, y in zip(a, b):
      assert np.array_equal(x, y)
    boxes, labels, probs, valid, feats = a
    assert np.all(valid == cfg.result_per_im)
    assert boxes.min() >= 0 and boxes[..., 2].max() <= 1920 and boxes[..., 3].max() <= 1080
    assert np.all(np.diff(probs, axis=1) <= 0) and probs.max() <= 1 and probs.min() >= 0
    assert labels.min() >= 1 and labels.max() <= cfg.num_class - 1
    assert np.isfinite(feats).all() and feats.shape == (int(valid.sum()), 256, 7, 7)
  finally:
    m.close()


def test_forward_coco_v2_partial_classes(backend):
  """Model-zoo variant (reference obj_detect_tracking.py:233-239,263-290): version 2 = no
  dilations, 81 COCO classes, --use_partial_classes keeps a class subset of the head
  (models.py:807-829).  The oracle gathers logits as the graph does; the product gathers the
  weight columns once at load time."""
  name, lib = backend
  names = ["BG"] + ["c%d" % i for i in range(1, 81)]
  part = ["c1", "c3", "c4", "c6", "c8", "c17", "c80"]
  cfg = small_config(resnet_num_block=[1, 1, 1, 1], version=2, use_dilations=False, num_class=81,
                     is_coco_model=True, use_partial_classes=True, partial_classes=part,
                     classname2id={n: i for i, n in enumerate(names)})
  miss, extra = _run_single(lib, cfg, 96, 128)
  assert miss == 0 and extra == 0
  m = models.get_model(cfg, 0, weights=weights_for(cfg), lib=lib)
  try:
    assert m.head_num_class == 8
    _, labels, _, _ = m.predict(synthetic_frames(1, 96, 128)[0])
    assert labels.size and labels.min() >= 1 and labels.max() <= 7   # 1..num_partial (ref :637-641)
  finally:
    m.close()


def test_forward_version5_class_agnostic(backend):
  """obj_v4 / obj_v5 models (reference obj_detect_tracking.py:272-277): dilated res5 +
  class-agnostic box regression (models.py:1126-1170, :798-802: one [K,1,4] regression tiled over
  the foreground classes).  Product: the four box columns tiled at load time."""
  name, lib = backend
  cfg = small_config(resnet_num_block=[1, 1, 1, 1], version=5)
  assert cfg.use_frcnn_class_agnostic and cfg.use_dilations
  assert weights_for(cfg)["fastrcnn/outputs/box/W"].shape[1] == 4
  miss, extra = _run_single(lib, cfg, 96, 128)
  assert miss == 0 and extra == 0
  with pytest.raises(NotImplementedError):
    models.get_model(small_config(version=6), 0, weights=weights_for(cfg), lib=lib)   # SE-ResNet


def test_forward_single_with_mask_head(backend):
  """--add_mask (reference models.py:932-962, 1173-1199): final_masks [R,28,28] next to the usual
  outputs; fetched through the Session shim like obj_detect_tracking.py:626-631."""
  name, lib = backend
  small = dict(result_per_im=6, mrcnn_head_dim=64) if name == "emu" else {}   # simulator cost
  cfg = small_config(resnet_num_block=[1, 1, 1, 1], add_mask=True, rpn_test_post_nms_topk=32, **small)
  w = weights_for(cfg)
  fr = synthetic_frames(1, 96, 128)[0]
  ref = OracleModel(cfg, w).forward(fr)
  m = models.get_model(cfg, 0, weights=w, lib=lib)
  try:
    sess = models.Session()
    boxes, labels, probs, feats, masks = sess.run(
        [m.final_boxes, m.final_labels, m.final_probs, m.fpn_box_feat, m.final_masks],
        feed_dict=m.get_feed_dict_forward(fr))
    assert masks.shape == (boxes.shape[0], 28, 28) and masks.dtype == np.float32
    assert np.array_equal(labels, ref["final_labels"])
    np.testing.assert_allclose(boxes, ref["final_boxes"], rtol=0, atol=1e-3)
    assert masks.min() >= 0 and masks.max() <= 1 and masks.std() > 1e-3
    np.testing.assert_allclose(masks, ref["final_masks"], rtol=0, atol=2e-5)
  finally:
    m.close()
  with pytest.raises(NotImplementedError):
    models.get_model(small_config(add_mask=True, im_batch_size=2), 0, weights=w, lib=lib, is_multi=True)


def test_arithmetic_modes_agree(backend, monkeypatch):
  """The two arithmetic modes of the conv path on the same frame: exact-f32 MFMA everywhere
  (ODT_CONV_SPLIT=0) against every eligible layer on the bf16x3 split kernel (forced with
  MINTILES=1; at 1080p the library picks it by itself).  Stage tensors agree at f32 rounding
  level, detections as matched sets; both are separately checked against the oracle elsewhere."""
  name, lib = backend
  cfg = small_config(resnet_num_block=[1, 1, 1, 1] if name == "emu" else [1, 1, 2, 3])
  w = weights_for(cfg)
  H, W = (64, 96) if name == "emu" else (96, 128)
  fr = synthetic_frames(1, H, W, seed=5)
  out = {}
  for mode in ("0", "1"):
    monkeypatch.setenv("ODT_CONV_SPLIT", mode)
    monkeypatch.setenv("ODT_CONV_SPLIT_MINTILES", "1")
    m = models.get_model(_with_taps(cfg), 0, weights=w, lib=lib)
    try:
      boxes, labels, probs, feats = m.predict(fr[0])
      e = m.engine(1, H, W)
      nsplit = sum(1 for nm, _, _, _ in e.profile_layers() if nm.endswith("[bf16x3]"))
      out[mode] = (boxes, labels, probs, {k: e.tap(k) for k in ("c2", "c3", "c4", "c5", "p2", "p5", "rpn2")}, nsplit)
    finally:
      m.close()
  assert out["0"][4] == 0 and out["1"][4] > 20, (out["0"][4], out["1"][4])
  for k, t in out["0"][3].items():
    assert _rel(out["1"][3][k], t) < 2e-5, k
  miss, extra = match_detections(out["1"][0], out["1"][1], out["1"][2], out["0"][0], out["0"][1], out["0"][2],
                                 1e-3, 1e-4)
  assert miss + extra <= 2, (miss, extra)


@pytest.mark.gpu
def test_arena_four_coresident_handles_b8_1080p(hip_lib):
  """The reference runs several videos per GPU (SPEED.md:61); with the activations planned into an arena a b=8 @1080p
  handle is ~5 GB (a dedicated buffer per stage tensor: ~25 GB), so four of them fit beside each other.  All four
  give the same bits, interleaved on their own streams, and the handle reports what it allocated."""
  import torch
  from object_detection_tracking_amd._lib import ODT_DTYPE_U8
  cfg = make_config(rpn_test_post_nms_topk=300, im_batch_size=8)
  w = weights_for(cfg)
  fr = synthetic_frames(8, 1080, 1920, seed=21)
  ms = [models.get_model(cfg, 0, weights=w, lib=hip_lib, is_multi=True) for _ in range(4)]
  try:
    es = [m.engine(8, 1080, 1920) for m in ms]
    mem = es[0].describe()["memory"]
    assert mem["keep_taps"] == 0 and mem["device_bytes"] <= 8e9, mem
    assert mem["activation_arena_bytes"][0] + mem["activation_arena_bytes"][1] < 0.25 * mem["arena_tensor_bytes_unshared"], mem
    want = es[0].forward(fr, want_feats=False, want_pooled=True)
    d = torch.from_numpy(fr).cuda(0)
    for rep in range(3):                       # forwards of the four handles in flight together, twice over
      for e in es:
        e.forward_device_async(d.data_ptr(), ODT_DTYPE_U8)
    for e in es:
      got = e.read_outputs(want_feats=False, want_pooled=True)
      for a, b in zip(got[:4], want[:4]):
        assert np.array_equal(a, b)
      assert np.array_equal(got[5], want[5])
  finally:
    for m in ms:
      m.close()


def test_rpn_head_fused_into_conv_epilogue(backend, monkeypatch):
  """rpn/head@pL (1x1, 256 -> 3 logits || 12 deltas) evaluated inside the epilogue of rpn/conv0@pL where that conv runs on
  a conv_split3 kernel with the whole Cout in one n-tile (at 1080p: P2, P3, P4; here forced by the tile thresholds):
  the RPN outputs agree with the two-launch form at f32 rounding level, the whole forward with the oracle, and the
  handle reports the folded launches."""
  name, lib = backend
  monkeypatch.setenv("ODT_CONV_SPLIT_MINTILES", "1"); monkeypatch.setenv("ODT_CONV_SPLIT3_MINTILES", "1")
  cfg = small_config(resnet_num_block=[1, 1, 1, 1])
  w = weights_for(cfg)
  H, W = (64, 96) if name == "emu" else (160, 224)
  fr = synthetic_frames(1, H, W, seed=5)
  out = {}
  for mode in ("0", "1"):
    monkeypatch.setenv("ODT_FUSE_RPN_HEAD", mode)
    m = models.get_model(_with_taps(cfg), 0, weights=w, lib=lib)
    try:
      det = m.predict(fr[0])
      e = m.engine(1, H, W)
      out[mode] = (det, {l: e.tap("rpn%d" % l) for l in range(2, 7)}, e.describe())
    finally:
      m.close()
  assert out["0"][2]["convs_fused_into_epilogues"] == 0 and out["1"][2]["convs_fused_into_epilogues"] >= 2, out["1"][2]
  assert out["1"][2]["conv_launches"] + out["1"][2]["convs_fused_into_epilogues"] == out["0"][2]["conv_launches"]
  for l in range(2, 7):
    a, b = out["1"][1][l], out["0"][1][l]
    assert np.all(a[..., 15] == 0)
    assert _rel(a, b) < 1e-5, l
  miss, extra = match_detections(out["1"][0][0], out["1"][0][1], out["1"][0][2], out["0"][0][0], out["0"][0][1], out["0"][0][2], 1e-3, 1e-4)
  assert miss + extra == 0
  if name == "hip":                                    # (simulator: the two-launch form is what the oracle tests cover; fused == two-launch above)
    monkeypatch.setenv("ODT_FUSE_RPN_HEAD", "1")
    miss, extra = _run_single(lib, cfg, H, W)         # fused form against the oracle, arena + taps handles
    assert miss == 0 and extra == 0



def test_bottleneck_conv3_fused_into_conv2_kernel(backend, monkeypatch):
  """block/conv3 (1x1 + BN + shortcut + ReLU) evaluated from block/conv2's accumulators inside conv_h2k_kernel where conv2
  runs there with its whole Cout in one 256-wide n-tile (at 1080p, b = 8: the 22 identity blocks of res4; here forced by
  the tile thresholds on a 256 x 256 frame: res4 has 16 x 16 pixels): stage tensors agree with the two-launch form at f32
  rounding level, detections as sets, the handle reports the folded launches, and (GPU) the fused form agrees with the
  oracle through both the arena and the keep_taps handle."""
  name, lib = backend
  monkeypatch.setenv("ODT_CONV_SPLIT_MINTILES", "1"); monkeypatch.setenv("ODT_CONV_SPLIT3_MINTILES", "1")
  monkeypatch.setenv("ODT_CONV_SPLIT3_BM", "256")
  # (res2's identity blocks are 64 wide: the 256 x 64 tile's tail, K halves of 32; res3's 128 wide, K halves of 64; res4's 256 wide)
  cfg = small_config(resnet_num_block=[2, 2, 2, 1] if name == "emu" else [2, 2, 4, 3], max_size=256, short_edge_size=256)
  w = weights_for(cfg)
  H, W = 256, 256
  fr = synthetic_frames(1, H, W, seed=5)
  out = {}
  for mode in ("0", "3"):
    monkeypatch.setenv("ODT_FUSE_BOTTLENECK", mode)
    m = models.get_model(_with_taps(cfg), 0, weights=w, lib=lib)
    try:
      det = m.predict(fr[0])
      e = m.engine(1, H, W)
      out[mode] = (det, {k: e.tap(k) for k in ("c2", "c3", "c4", "c5", "p4", "rpn4")}, e.describe(), [nm for nm, _, _, _ in e.profile_layers()])
    finally:
      m.close()
  nf = out["3"][2]["bottleneck_tails_fused"]
  assert out["0"][2]["bottleneck_tails_fused"] == 0 and nf == (3 if name == "emu" else 5), out["3"][2]
  assert out["3"][2]["conv_launches"] + nf == out["0"][2]["conv_launches"]
  assert sum(1 for nm in out["3"][3] if "conv2+conv3[fp16x2]" in nm) == nf, out["3"][3]
  for k in ("c2", "c3", "c4", "c5", "p4", "rpn4"):
    assert _rel(out["3"][1][k], out["0"][1][k]) < 1e-5, k
  miss, extra = match_detections(out["3"][0][0], out["3"][0][1], out["3"][0][2], out["0"][0][0], out["0"][0][1], out["0"][0][2], 1e-3, 1e-4)
  assert miss + extra == 0
  if name == "hip":
    monkeypatch.setenv("ODT_FUSE_BOTTLENECK", "3")
    miss, extra = _run_single(lib, cfg, H, W)
    assert miss == 0 and extra == 0


@pytest.mark.parametrize("tiles", ["256", "128/k2"])
def test_fp16x2_family_agrees_with_bf16x3(backend, tiles, monkeypatch):
  """conv_split_family = 2: the layers with 256-row tiles at least 128 columns wide and a recorded input range run on the
  fp16x2 kernels (conv_h2.hip: three exact f16 products per MAC, operands scaled by the |max| the producing kernel recorded),
  the rest stays on bf16x3.  Stage tensors agree with the bf16x3 handle at f32 rounding level, detections as matched sets;
  the fp16x2 handle is also checked against the oracle."""
  name, lib = backend
  if name == "emu" and tiles == "256":
    pytest.skip("simulator: the 256-row fp16x2 kernels are covered per op (tests/test_ops.py pipes 2/256...); the whole-model "
                "run of them takes two minutes there and runs on the GPU")
  monkeypatch.setenv("ODT_CONV_SPLIT_MINTILES", "1"); monkeypatch.setenv("ODT_CONV_SPLIT3_MINTILES", "1")
  # "256": 256-row tiles (conv_h2k_kernel / conv_h2_kernel<., 4>); "128/k2": the 128 x 128 4-wave tile with the reduction cut
  # in two (what b = 1 runs below res3): the ranges' combine pass applies the scales and records the output's range
  monkeypatch.setenv("ODT_CONV_SPLIT3_BM", tiles.split("/")[0])
  if "/" in tiles:
    monkeypatch.setenv("ODT_CONV_SPLIT3_FORCE_SPLITK", tiles.split("/")[1][1:])
  cfg = small_config(resnet_num_block=[1, 1, 1, 1] if name == "emu" else [1, 1, 2, 3])
  w = weights_for(cfg)
  H, W = (64, 96) if name == "emu" else (160, 224)
  fr = synthetic_frames(1, H, W, seed=5)
  out = {}
  for fam in (3, 2):
    c = _with_taps(cfg); c.conv_split_family = fam
    m = models.get_model(c, 0, weights=w, lib=lib)
    try:
      boxes, labels, probs, feats = m.predict(fr[0])
      e = m.engine(1, H, W)
      out[fam] = (boxes, labels, probs, {k: e.tap(k) for k in ("c2", "c3", "c4", "c5", "p2", "p5", "rpn2", "rpn4")}, e.describe())
      names = [nm for nm, _, _, _ in e.profile_layers()]
      if fam == 2:
        # conv0 reads the range the preprocess kernel records; P6 (max-pooled P5) inherits P5's; the 64-wide res2 layers
        for want in ("conv0", "rpn/conv0@p6", "group0/block0/conv2"):
          assert any(nm.startswith(want) and nm.endswith("[fp16x2]") for nm in names), (want, names[:12])
      if fam == 2:          # a second frame through the same handle: the range slots start every forward at zero
        b2, l2, p2, _ = m.predict(fr[0])
        assert np.array_equal(b2, boxes) and np.array_equal(p2, probs)
    finally:
      m.close()
  assert out[3][4]["fp16x2_split_launches"] == 0 and out[2][4]["fp16x2_split_launches"] >= 10, out[2][4]
  for k, t in out[3][3].items():
    assert _rel(out[2][3][k], t) < 2e-5, k
  miss, extra = match_detections(out[2][0], out[2][1], out[2][2], out[3][0], out[3][1], out[3][2], 1e-3, 1e-4)
  assert miss + extra <= 2, (miss, extra)
  if name == "hip":
    c = small_config(resnet_num_block=[1, 1, 2, 3]); c.conv_split_family = 2
    miss, extra = _run_single(lib, c, H, W)
    assert miss == 0 and extra == 0


def test_cache_hints_and_record_granularity_do_not_change_results(backend, monkeypatch):
  """The non-temporal load hints (ODT_CONV_NT), the K-slice rotation being on, and the range record per workgroup / per wave
  are performance knobs: the forward's outputs are bit-identical with them off (the rotation changes the summation order of
  the 1x1 tiles and is therefore NOT part of this list)."""
  name, lib = backend
  if name == "emu":
    pytest.skip("simulator: cache hints do not exist there and both record paths run in the other e2e tests; a GPU test")
  monkeypatch.setenv("ODT_CONV_SPLIT_MINTILES", "1"); monkeypatch.setenv("ODT_CONV_SPLIT3_MINTILES", "1")
  cfg = small_config(resnet_num_block=[1, 1, 1, 1])
  w = weights_for(cfg)
  H, W = 160, 224
  fr = synthetic_frames(1, H, W, seed=7)
  out = []
  for env in ({}, {"ODT_CONV_NT": "0", "ODT_AMAX_PER_WAVE": "1"}):
    for k in ("ODT_CONV_NT", "ODT_AMAX_PER_WAVE"):
      monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
      monkeypatch.setenv(k, v)
    m = models.get_model(cfg, 0, weights=w, lib=lib)
    try:
      out.append(m.predict(fr[0]))
    finally:
      m.close()
  for a, b in zip(out[0], out[1]):
    assert np.array_equal(a, b)


@pytest.mark.parametrize("shape", [(1, 96, 160), (2, 130, 200)])
def test_stem_conv0_pool0_in_one_kernel(backend, shape, monkeypatch):
  """conv0 + BN + ReLU + pool0 as ONE launch of conv_stem_kernel (a patch of the padded frame per 8 x 7 pooled pixels, split
  once into LDS; the conv map never written): same products in the same order as conv_h2_kernel + maxpool3x3s2_kernel, so
  every output of the forward is BIT-IDENTICAL to the two-launch form (tiles partial in x, tiles at the frame's borders,
  several images, more workgroups than tiles and fewer); a keep_taps handle -- which exposes conv0 -- keeps the two launches."""
  name, lib = backend
  B, H, W = shape
  if name == "hip":
    H, W = 4 * H, 4 * W                     # (more tiles than CUs in the second shape: 2 x 17 x 29 tiles of 8 x 7 pooled pixels)
  elif B == 1:
    pytest.skip("simulator: the two-image shape covers it (a forward of the whole model takes half a minute there)")
  else:
    H, W = 64, 80                            # (simulator: 2 x 2 x 4 tiles, the last of a row partial, three workgroups)
  monkeypatch.setenv("ODT_CONV_SPLIT_MINTILES", "1"); monkeypatch.setenv("ODT_CONV_SPLIT3_MINTILES", "1")
  cfg = small_config(resnet_num_block=[1, 1, 1, 1], im_batch_size=B, rpn_test_post_nms_topk=32, max_size=max(H, W), short_edge_size=min(H, W))
  w = weights_for(cfg)
  fr = synthetic_frames(B, H, W, seed=11)
  out = {}
  if name == "emu":
    monkeypatch.setenv("ODT_STEM_GRID", "3")     # (the simulator's 256 CUs: a persistent workgroup would see one tile)
  for mode in ("0", "1"):
    monkeypatch.setenv("ODT_FUSE_STEM", mode)
    m = models.get_model(cfg, 0, weights=w, lib=lib, is_multi=B > 1)
    try:
      res = m.predict_batch(fr) if B > 1 else m.predict(fr[0])
      e = m.engine(B, H, W)
      out[mode] = (res, e.describe(), [nm for nm, _, _, _ in e.profile_layers()])
    finally:
      m.close()
  assert out["0"][1]["stem_fused"] == 0 and out["1"][1]["stem_fused"] == 1, (out["0"][1], out["1"][1])
  assert "conv0+pool0[fp16x2]" in out["1"][2] and "conv0[fp16x2]" in out["0"][2], out["1"][2][:3]
  for a, b in zip(out["0"][0], out["1"][0]):
    assert np.array_equal(a, b)
  monkeypatch.setenv("ODT_FUSE_STEM", "1")
  m = models.get_model(_with_taps(cfg), 0, weights=w, lib=lib, is_multi=B > 1)
  try:
    if name == "hip":
      res = m.predict_batch(fr) if B > 1 else m.predict(fr[0])
      for a, b in zip(out["1"][0], res):
        assert np.array_equal(a, b)
    assert m.engine(B, H, W).describe()["stem_fused"] == 0
  finally:
    m.close()


def test_convs_cut_into_batch_ranges_are_bit_identical(backend, monkeypatch):
  """A conv whose tensors would reach 2 GiB (32-bit buffer offsets; b = 16 @1080p) runs as several launches over batch
  ranges.  With the limit lowered (test knob) a small batched plan takes that path for most layers: same bits out."""
  name, lib = backend
  B, H, W = (2, 64, 96) if name == "emu" else (4, 160, 224)
  cfg = small_config(resnet_num_block=[1, 1, 1, 1], im_batch_size=B, rpn_test_post_nms_topk=32)
  w = weights_for(cfg)
  fr = synthetic_frames(B, H, W, seed=4)
  outs = []
  for limit in (None, "400000"):
    if limit is None:
      monkeypatch.delenv("ODT_CONV_CHUNK_BYTES", raising=False)
    else:
      monkeypatch.setenv("ODT_CONV_CHUNK_BYTES", limit)
    m = models.get_model(cfg, 0, weights=w, lib=lib, is_multi=True)
    try:
      outs.append((m.predict_batch(fr), m.engine(B, H, W).describe()["convs_cut_into_batch_ranges"]))
    finally:
      m.close()
  assert outs[0][1] == 0 and outs[1][1] >= 5, (outs[0][1], outs[1][1])
  for a, b in zip(outs[0][0], outs[1][0]):
    assert np.array_equal(a, b)


@pytest.mark.gpu
def test_forward_multi_b24_1080p(hip_lib):
  """b = 24 @1080p: conv0's output, the res2 tensors and the P2-level tensors pass 2 GiB (b = 16 is just below it) -- the
  plan cuts those convs into two batch ranges.  Frames 0..7 give the detections of the b = 8 plan (the box head's
  split-K choice differs with the row count, so compared as matched sets within the e2e tolerance)."""
  cfgb = make_config(rpn_test_post_nms_topk=300, im_batch_size=24)
  cfg8 = make_config(rpn_test_post_nms_topk=300, im_batch_size=8)
  w = weights_for(cfgb)
  fr8 = synthetic_frames(8, 1080, 1920, seed=3)
  fr = np.concatenate([fr8, fr8[::-1], fr8], 0)
  m8 = models.get_model(cfg8, 0, weights=w, lib=hip_lib, is_multi=True)
  try:
    ref = m8.predict_batch(fr8)
  finally:
    m8.close()
  m = models.get_model(cfgb, 0, weights=w, lib=hip_lib, is_multi=True)
  try:
    boxes, labels, probs, valid, feats = m.predict_batch(fr)
    d = m.engine(24, 1080, 1920).describe()
    assert d["convs_cut_into_batch_ranges"] >= 10 and d["memory"]["device_bytes"] < 16e9, d
    assert np.all(valid > 0) and np.isfinite(feats).all()
    for b, rb in [(i, i) for i in range(8)] + [(8 + i, 7 - i) for i in range(8)] + [(16 + i, i) for i in range(8)]:
      v = valid[b]
      assert v == ref[3][rb], (b, v, ref[3][rb])
      miss, extra = match_detections(boxes[b, :v], labels[b, :v], probs[b, :v], ref[0][rb, :v], ref[1][rb, :v], ref[2][rb, :v], 1e-2, 1e-4)
      assert miss + extra == 0, (b, miss, extra)
  finally:
    m.close()


# ---- the fp16x2 default's range assumption: guard (conv_split_family = "auto") and out-of-domain behaviour -----------------
def _split_multi_ref(ref, B):
  """Per image (boxes, labels, probs) of the oracle's batched outputs."""
  out = []
  for b in range(B):
    v = int(ref["final_valid_indices"][b])
    out.append((ref["final_boxes"][b, :v], ref["final_labels"][b, :v], ref["final_probs"][b, :v]))
  return out


def _outlier_weights(cfg, exp=30):
  """Synthetic weights whose conv0 output carries ONE channel 2^exp above the rest -- and nothing downstream reads it
  (zero rows in the convs that consume pool0): every useful value of that tensor sits 2^exp below the tensor's |max|,
  outside what one power of two per tensor leaves an f16 pair."""
  w = {k: np.array(v, copy=True) for k, v in weights_for(cfg).items()}
  c = 5
  w["conv0/bn/gamma"][c] *= np.float32(2.0 ** exp)
  w["conv0/bn/beta"][c] = 0
  w["group0/block0/conv1/W"][:, :, c, :] = 0
  w["group0/block0/convshortcut/W"][:, :, c, :] = 0
  return w


def test_auto_family_guards_the_fp16x2_range_assumption(backend, monkeypatch):
  """conv_split_family = "auto": the first forward also runs on a bf16x3-only twin handle; ordinary weights stay on the
  fp16x2 kernels (pyramid / RPN tensors agree at f32 rounding level), weights with an outlier channel 2^30 above the useful
  content of its tensor make the fp16x2 handle deviate by orders of magnitude more -- the engine continues on bf16x3 and
  agrees with the oracle, which a forced fp16x2 handle does not."""
  name, lib = backend
  monkeypatch.setenv("ODT_CONV_SPLIT_MINTILES", "1"); monkeypatch.setenv("ODT_CONV_SPLIT3_MINTILES", "1")
  cfg = small_config(resnet_num_block=[1, 1, 1, 1])
  H, W = (64, 96) if name == "emu" else (160, 224)
  fr = synthetic_frames(1, H, W, seed=5)
  import copy
  # (simulator: the outlier case only -- ordinary weights on both families there: test_fp16x2_family_agrees_with_bf16x3)
  for kind in (("outlier",) if name == "emu" else ("ordinary", "outlier")):
    w = weights_for(cfg) if kind == "ordinary" else _outlier_weights(cfg)
    ref = OracleModel(cfg, w).forward(fr[0])
    ca = copy.copy(cfg); ca.conv_split_family = "auto"
    m = models.get_model(ca, 0, weights=w, lib=lib)
    try:
      boxes, labels, probs, feats = m.predict(fr[0])
      e = m.engine(1, H, W)
      d = e.describe()
      auto = d["conv_split_family_auto"]
      assert len(auto["checks"]) == 1 and auto["calibration_forwards_left"] == 0, auto
      p2 = e.tap("p2").transpose(0, 3, 1, 2)[:, :, :ref["p2"].shape[2], :ref["p2"].shape[3]]
      if kind == "ordinary":
        assert auto["chosen"].startswith("fp16x2") and auto["checks"][0]["max_rel_diff"] < 2e-5, auto
        assert d["fp16x2_split_launches"] >= 10, d
      else:
        assert auto["chosen"].startswith("bf16x3") and auto["checks"][0]["max_rel_diff"] > 1e-4, auto
        assert d["fp16x2_split_launches"] == 0, d
      assert _rel(p2, ref["p2"]) < 2e-5, kind
      miss, extra = match_detections(boxes, labels, probs, ref["final_boxes"], ref["final_labels"], ref["final_probs"], 1e-3, 1e-4)
      assert miss + extra == 0, (kind, miss, extra)
      if name == "hip":
        b2, l2, p2b, _ = m.predict(fr[0])                # (the second forward runs on the chosen handle alone)
        assert np.array_equal(b2, boxes) and np.array_equal(p2b, probs)
    finally:
      m.close()
    if kind == "outlier" and name == "hip":
      # what the guard is for: the same weights on a handle FORCED to the fp16x2 family leave f32 level
      cf = _with_taps(cfg); cf.conv_split_family = 2
      m = models.get_model(cf, 0, weights=w, lib=lib)
      try:
        m.predict(fr[0])
        e = m.engine(1, H, W)
        p2f = e.tap("p2").transpose(0, 3, 1, 2)[:, :, :ref["p2"].shape[2], :ref["p2"].shape[3]]
        assert _rel(p2f, ref["p2"]) > 1e-4
        rep = e.range_report(("conv0", "pool0", "c2"))
        assert rep["pool0"]["frac_nonzero_below_2^-17_amax"] > 0.9 and rep["c2"]["frac_nonzero_below_2^-17_amax"] < 0.05, rep
      finally:
        m.close()


def test_default_engine_is_guarded_on_every_entry_path(backend, monkeypatch):
  """The PRODUCT default (the package's make_config, or an args object without the field) is conv_split_family = "auto":
  describe() names the guard; the first forward calibrates whichever way the frames arrive -- forward(), submit(frames)
  and the zero-copy ingest path ingest_buffer() + submit(None) (that path used to skip the guard and
  keep the weights dict alive for ever); on the ingest path an engine that moves to the bf16x3 twin takes the armed
  frames along, the caller's old buffer view stays valid memory until close(), and the profiling switch follows; a
  call that finds tickets outstanding never calibrates (it would clobber the ticket's device outputs) and is counted."""
  from object_detection_tracking_amd.config import make_config as product_make_config
  name, lib = backend
  assert product_make_config().conv_split_family == "auto"
  H, W = (64, 96) if name == "emu" else (160, 224)
  kw = dict(resnet_num_block=[1, 1, 1, 1], rpn_test_post_nms_topk=64, max_size=256, short_edge_size=96, im_batch_size=1)
  if name == "hip":
    monkeypatch.setenv("ODT_CONV_SPLIT_MINTILES", "1"); monkeypatch.setenv("ODT_CONV_SPLIT3_MINTILES", "1")
  else:
    # simulator: forced split kernels cost ~1 min per forward there, so the plan keeps its exact-f32 kernels at this size (both
    # handles then agree bit for bit) and a NEGATIVE tolerance makes the guard change handles all the same -- the host logic
    # under test (ingest buffer hand-over, retired handle, profiling switch, released twin) is what runs; the arithmetic
    # side of the guard is test_auto_family_guards_the_fp16x2_range_assumption's
    kw["conv_split_auto_tol"] = -1.0
  cfg = product_make_config(**kw)
  class Args(object):                      # a reference-style args object: no conv_split_family attribute at all
    pass
  bare = Args(); bare.__dict__.update({k: v for k, v in cfg.__dict__.items() if k != "conv_split_family"})
  fr = synthetic_frames(1, H, W, seed=5)
  cfg3 = product_make_config(conv_split_family=3, **kw)
  # (simulator: the handle-changing case through the ingest path only)
  for kind in (("outlier",) if name == "emu" else ("ordinary", "outlier")):
    w = weights_for(cfg) if kind == "ordinary" else _outlier_weights(cfg)
    m3 = models.get_model(cfg3, 0, weights=w, lib=lib, is_multi=True)
    try:
      want3 = m3.engine(1, H, W).forward(fr, want_feats=False, want_pooled=True)
      assert m3.engine(1, H, W).describe()["range_guard"].startswith("off")
    finally:
      m3.close()
    for path in (("ingest",) if name == "emu" else ("forward", "submit", "ingest")):
      m = models.get_model(bare if path != "submit" else cfg, 0, weights=w, lib=lib, is_multi=True)
      try:
        e = m.engine(1, H, W)
        d = e.describe()
        assert "auto" in d["range_guard"] and d["conv_split_family_auto"]["calibration_forwards_left"] == 1, d
        e.profile(True)
        if path == "forward":
          got = e.forward(fr, want_feats=False, want_pooled=True)
        elif path == "submit":
          got = e.collect(e.submit(fr, want_feats=False, want_pooled=True))
        else:
          view = e.ingest_buffer(np.uint8)
          np.copyto(view, fr)
          got = e.collect(e.submit(None, want_feats=False, want_pooled=True))
          assert np.array_equal(view, fr)             # the caller's view is still readable memory, whatever the guard chose
        auto = e.describe()["conv_split_family_auto"]
        assert len(auto["checks"]) == 1 and auto["calibration_forwards_left"] == 0 and not auto["incomplete"], auto
        assert "twin" not in e._auto                                      # the twin handle is released (the weights dict is the model's own:
        assert ("args" in e._auto) == (kind == "ordinary")                # kept while the guard watches an fp16x2 engine)
        assert e.profile_read()["conv_launches"] > 0                      # profiling survived a handle change
        if kind == "ordinary":
          assert auto["chosen"].startswith("fp16x2"), auto
        else:
          assert auto["chosen"].startswith("bf16x3"), auto
          for a, b in zip(got, want3):                                    # the engine IS the bf16x3 engine now
            assert (a is None and b is None) or np.array_equal(a, b)
          assert len(e._retired) == (1 if path == "ingest" else 0)
        if path == "ingest":                                              # the next ticket uses the live handle's buffer
          np.copyto(e.ingest_buffer(np.uint8), fr)
          again = e.collect(e.submit(None, want_feats=False, want_pooled=True))
          assert np.array_equal(again[0], got[0]) and np.array_equal(again[5], got[5])
      finally:
        m.close()
  # tickets outstanding at every call that could calibrate: skipped, counted, and given up after 16
  monkeypatch.delenv("ODT_CONV_SPLIT_MINTILES", raising=False); monkeypatch.delenv("ODT_CONV_SPLIT3_MINTILES", raising=False)
  kw.pop("conv_split_auto_tol", None)
  cfg2 = product_make_config(conv_split_auto_frames=2, **kw)
  m = models.get_model(cfg2, 0, weights=weights_for(cfg), lib=lib, is_multi=True)
  try:
    e = m.engine(1, H, W)
    t0 = e.submit(fr, want_feats=False, want_pooled=True)               # calibrates (no ticket yet): 1 of 2 done
    t1 = e.submit(fr, want_feats=False, want_pooled=True)               # t0 outstanding: skipped
    a = e.describe()["conv_split_family_auto"]
    assert len(a["checks"]) == 1 and a["calls_skipped_with_tickets_outstanding"] == 1 and a["calibration_forwards_left"] == 1, a
    r0, r1 = e.collect(t0), e.collect(t1)
    assert np.array_equal(r0[0], r1[0])
    e.forward(fr)                                                         # nothing outstanding: the second check runs
    a = e.describe()["conv_split_family_auto"]
    assert len(a["checks"]) == 2 and a["calibration_forwards_left"] == 0 and "twin" not in e._auto, a
  finally:
    m.close()


@pytest.mark.gpu
def test_mixed_exposure_batch_b8_1080p_and_batch_independence(hip_lib):
  """Config #3's batch with frames 0-1 under-exposed (pixel values 0 .. 15), 2-3 over-exposed (240 .. 255), 4-7 ordinary
  (exactly constant frames make every score a tie: nothing to compare): the fp16x2 kernels take ONE power of two per
  activation tensor ACROSS the batch, so a frame's arithmetic depends on its batch mates' range.  (i) the batch agrees with
  the oracle at the usual budgets (0 unmatched, boxes within 1e-3 px); (ii) every frame's pyramid features and detections
  from the b = 8 forward agree with the same frame alone (b = 1, same graph) at f32 rounding level."""
  cfg = make_config(rpn_test_post_nms_topk=300, im_batch_size=8)
  w = weights_for(cfg)
  H, W = 1080, 1920
  fr = synthetic_frames(8, H, W, seed=11)
  fr[0:2] = fr[0:2] // 16
  fr[2:4] = 255 - fr[2:4] // 16
  ref = OracleModel(cfg, w).forward_multi(fr)
  m = models.get_model(cfg, 0, weights=w, is_multi=True, lib=hip_lib)
  try:
    e = m.engine(8, H, W)
    boxes, labels, probs, valid, feats, _ = e.forward(fr)
    assert e.describe()["fp16x2_split_launches"] > 80
    p3 = e.tap("p3")
    got = (boxes, labels, probs, valid)
  finally:
    m.close()
  # (i) against the oracle, image by image
  per_ref = _split_multi_ref(ref, 8)
  for b in range(8):
    n = int(got[3][b])
    rb, rl, rp = per_ref[b]
    assert n == len(rb), (b, n, len(rb))
    miss, extra = match_detections(got[0][b, :n], got[1][b, :n], got[2][b, :n], rb, rl, rp, 1e-3, 1e-4)
    assert miss + extra == 0, (b, miss, extra)
  # (ii) every frame alone
  cfg1 = make_config(rpn_test_post_nms_topk=300, im_batch_size=1)
  m1 = models.get_model(cfg1, 0, weights=w, is_multi=True, lib=hip_lib)
  try:
    e1 = m1.engine(1, H, W)
    for b in range(8):
      b1, l1, p1, v1, _, _ = e1.forward(fr[b:b + 1])
      assert _rel(e1.tap("p3")[0], p3[b]) < 1e-5, b
      n = int(got[3][b])
      assert int(v1[0]) == n, (b, int(v1[0]), n)
      miss, extra = match_detections(got[0][b, :n], got[1][b, :n], got[2][b, :n], b1[0, :n], l1[0, :n], p1[0, :n], 1e-3, 1e-4)
      assert miss + extra == 0, (b, miss, extra)
  finally:
    m1.close()


def _heavy_tailed_weights(cfg, base):
  """The same network FUNCTION with heavy-tailed activations: a few BN channels of four tensors scaled by 2^7 ... 2^12
  (gamma and beta), the rows of the one conv that reads each tensor divided by the same power of two (ReLU and max-pool
  commute with a positive scale; powers of two commute with f32 rounding: the oracle's outputs do not change by a bit) --
  tensors whose |max| sits 2^7 ... 2^12 above what it was, i.e. the bulk of their content that much further below it."""
  w = {k: np.array(v, copy=True) for k, v in base.items()}
  rng = np.random.default_rng(4)
  def scale(prod, consumers, exp):
    idx = rng.choice(w[prod + "/bn/gamma"].shape[0], 3, replace=False)
    f = np.float32(2.0 ** exp)
    w[prod + "/bn/gamma"][idx] *= f; w[prod + "/bn/beta"][idx] *= f
    for c in consumers:
      w[c + "/W"][:, :, idx, :] /= f
  scale("conv0", ("group0/block0/conv1", "group0/block0/convshortcut"), 7)
  scale("group1/block1/conv1", ("group1/block1/conv2",), 10)
  scale("group2/block5/conv2", ("group2/block5/conv3",), 12)        # (inside a fused conv2 -> conv3 kernel at this size)
  scale("group2/block7/conv1", ("group2/block7/conv2",), 7)
  return w


def _trained_like_weights(cfg, base, seed=11):
  """BatchNorm statistics with the shape of a TRAINED checkpoint instead of the near-identity synthetic ones: moving
  variances over four decades (gamma following, so gamma / sigma keeps its level), on top of that a per-channel
  gamma / sigma factor spanning 10^3 (log-uniform 10^-1.5 ... 10^1.5) on every BN of the backbone, a few channels another
  10^2 up -- and READ by their consumers --, 3 % dead channels (gamma = 0, beta < 0: zero after ReLU) behind every
  BN + ReLU.  The per-channel factors are compensated in the rows of the convs that read the tensor (as training does:
  what one layer's BN scales up, the next layer's weights scale down), so the network still detects things; the
  activation TENSORS -- conv0, every conv1 / conv2 output and the residual trunk of every group -- now carry channel
  scales over five decades (2^17), which is what one power of two per tensor has to serve."""
  w = {k: np.array(v, copy=True) for k, v in base.items()}
  rng = np.random.default_rng(seed)
  blocks = list(cfg.resnet_num_block)
  def factors(n):
    f = (10.0 ** rng.uniform(-1.5, 1.5, n)).astype(np.float32)
    f[rng.choice(n, max(2, n // 64), replace=False)] *= np.float32(100.0)          # consumed outliers
    return f
  def rescale(producers, consumers, dead):
    n = w[producers[0] + "/bn/gamma"].shape[0]
    f = factors(n)
    for pr in producers:
      v = (10.0 ** rng.uniform(-2, 2, n)).astype(np.float32)
      w[pr + "/bn/variance/EMA"] *= v
      w[pr + "/bn/gamma"] *= np.sqrt(v) * f
      w[pr + "/bn/beta"] *= f
      w[pr + "/bn/mean/EMA"] *= np.sqrt(v)
      if dead:
        idx = rng.choice(n, max(1, n * 3 // 100), replace=False)
        w[pr + "/bn/gamma"][idx] = 0
        w[pr + "/bn/beta"][idx] = -np.abs(w[pr + "/bn/beta"][idx]) - np.float32(0.01)
    for c in consumers:
      w[c + "/W"] /= f[None, None, :, None]
  rescale(["conv0"], ["group0/block0/conv1", "group0/block0/convshortcut"], True)
  for g, cnt in enumerate(blocks):
    for i in range(cnt):
      pre = "group%d/block%d" % (g, i)
      rescale([pre + "/conv1"], [pre + "/conv2"], True)
      rescale([pre + "/conv2"], [pre + "/conv3"], True)
    # the group's residual trunk: every block writes it (conv3 + shortcut), every later block, the next group and the
    # FPN lateral read it
    prod = ["group%d/block%d/conv3" % (g, i) for i in range(cnt)] + ["group%d/block0/convshortcut" % g]
    cons = ["group%d/block%d/conv1" % (g, i) for i in range(1, cnt)] + ["fpn/lateral_1x1_c%d" % (g + 2)]
    if g + 1 < len(blocks):
      cons += ["group%d/block0/conv1" % (g + 1), "group%d/block0/convshortcut" % (g + 1)]
    rescale(prod, cons, False)
  return w


@pytest.mark.gpu
def test_trained_like_bn_statistics_b8_1080p_default_engine(hip_lib):
  """the DEFAULT engine (the package's make_config: conv_split_family = "auto") on weights
  with trained-checkpoint-shaped BatchNorm statistics (_trained_like_weights) at BASELINE config #3's size.  Whatever
  the guard chooses must agree with the oracle at the usual budgets (every detection matched, boxes within 1e-3 px);
  the choice and the measured fp16x2-vs-bf16x3 difference are reported by describe()."""
  from object_detection_tracking_amd.config import make_config as product_make_config
  cfg = product_make_config(rpn_test_post_nms_topk=300, im_batch_size=8)
  assert cfg.conv_split_family == "auto"
  H, W, B = 1080, 1920, 8
  fr = synthetic_frames(B, H, W, seed=21)
  w = _trained_like_weights(cfg, weights_for(cfg))
  ref = OracleModel(cfg, w).forward_multi(fr)
  per_ref = _split_multi_ref(ref, B)
  assert sum(len(r[0]) for r in per_ref) >= 4 * B, "the rescaled network no longer detects anything: the test would be empty"
  m = models.get_model(cfg, 0, weights=w, is_multi=True, lib=hip_lib)
  try:
    e = m.engine(B, H, W)
    boxes, labels, probs, valid, _, _ = e.forward(fr)
    d = e.describe()
    auto = d["conv_split_family_auto"]
    print("trained-like BN statistics: guard chose", auto["chosen"], auto["checks"])
    assert "auto" in d["range_guard"] and len(auto["checks"]) == 1 and auto["calibration_forwards_left"] == 0, d
    worst = 0.0
    for b in range(B):
      n = int(valid[b]); rb, rl, rp = per_ref[b]
      assert n == len(rb), (b, n, len(rb))
      miss, extra = match_detections(boxes[b, :n], labels[b, :n], probs[b, :n], rb, rl, rp, 1e-3, 1e-4)
      assert miss + extra == 0, (b, miss, extra)
    # the pyramid itself, not only what survives selection
    from object_detection_tracking_amd._lib import OdtError
    seen = 0
    for l in range(2, 7):
      r = ref["p%d" % l]
      try:
        t = e.tap("p%d" % l).transpose(0, 3, 1, 2)[:, :, :r.shape[2], :r.shape[3]]
      except OdtError:
        continue                       # (production handle: a level whose memory the arena has reused by the end of the forward)
      worst = max(worst, _rel(t, r)); seen += 1
    assert seen >= 3 and worst < 2e-5, (seen, worst)
  finally:
    m.close()


@pytest.mark.gpu
def test_heavy_tailed_bn_gamma_1080p_auto_family(hip_lib):
  """Full size (b = 2 @1080p), activations with outlier channels 2^7 ... 2^12 above the rest (the regime of trained
  checkpoints; see _heavy_tailed_weights): within the fp16x2 kernels' domain -- conv_split_family = "auto" keeps them and
  the forward agrees with the oracle at the usual budgets; with ONE channel 2^30 up and nothing reading it the auto
  engine leaves fp16x2 and still agrees."""
  import copy
  cfg = make_config(rpn_test_post_nms_topk=300, im_batch_size=2)
  H, W = 1080, 1920
  fr = synthetic_frames(2, H, W, seed=3)
  base = weights_for(cfg)
  for tag, w, want in (("heavy-tailed", _heavy_tailed_weights(cfg, base), "fp16x2"), ("outlier 2^30", _outlier_weights(cfg), "bf16x3")):
    ref = OracleModel(cfg, w).forward_multi(fr)
    per_ref = _split_multi_ref(ref, 2)
    ca = copy.copy(cfg); ca.conv_split_family = "auto"
    m = models.get_model(ca, 0, weights=w, is_multi=True, lib=hip_lib)
    try:
      e = m.engine(2, H, W)
      boxes, labels, probs, valid, _, _ = e.forward(fr)
      auto = e.describe()["conv_split_family_auto"]
      assert auto["chosen"].startswith(want), (tag, auto)
      for b in range(2):
        n = int(valid[b]); rb, rl, rp = per_ref[b]
        assert n == len(rb), (tag, b, n, len(rb))
        miss, extra = match_detections(boxes[b, :n], labels[b, :n], probs[b, :n], rb, rl, rp, 1e-3, 1e-4)
        assert miss + extra == 0, (tag, b, miss, extra)
    finally:
      m.close()


def _exposure_outlier_weights(cfg, exp=30):
  """_outlier_weights whose outlier channel only fires on SATURATED frames: conv0 channel 5 averages its 7 x 7 x 3 patch of the
  normalised frame against a BatchNorm mean of 2.0 -- frames that stay below 200 / 255 (normalised <= 1.68) leave it at zero
  behind the ReLU, a patch of 255s (2.25 .. 2.64) gives 0.44 x 2^exp; nothing downstream reads the channel.  A stream that
  switches from ordinary to over-exposed frames moves pool0's |max| 2^exp above its useful content MID-RUN."""
  w = {k: np.array(v, copy=True) for k, v in weights_for(cfg).items()}
  c = 5
  w["conv0/W"][:, :, :, c] = np.float32(1.0 / 147.0)
  w["conv0/bn/gamma"][c] = np.float32(2.0 ** exp)
  w["conv0/bn/beta"][c] = 0
  w["conv0/bn/mean/EMA"][c] = 2.0
  w["conv0/bn/variance/EMA"][c] = 1.0
  w["group0/block0/conv1/W"][:, :, c, :] = 0
  w["group0/block0/convshortcut/W"][:, :, c, :] = 0
  return w


def _exposure_stream(B, H, W, seed):
  """(ordinary, saturated): the same scene clipped to <= 200, and with a quarter of the frame blown out to 255."""
  fr = np.minimum(synthetic_frames(B, H, W, seed=seed), 200).astype(np.uint8)
  hot = fr.copy()
  hot[:, : H // 2, : W // 2] = 255
  return fr, hot


def _run_exposure_switch(lib, cfg, B, H, W, multi, ordinary_frames=3, max_frames_after_cut=4):
  import copy
  w = _exposure_outlier_weights(cfg)
  fr, hot = _exposure_stream(B, H, W, seed=9)
  ca = copy.copy(cfg); ca.conv_split_family = "auto"
  c3 = copy.copy(cfg); c3.conv_split_family = 3
  m3 = models.get_model(c3, 0, weights=w, lib=lib, is_multi=multi)
  try:
    want_hot = m3.engine(B, H, W).forward(hot, want_feats=False, want_pooled=True)
  finally:
    m3.close()
  m = models.get_model(ca, 0, weights=w, lib=lib, is_multi=multi)
  try:
    e = m.engine(B, H, W)
    for k in range(ordinary_frames):
      e.forward(fr, want_feats=False, want_pooled=True)
    d = e.describe()["conv_split_family_auto"]
    # ordinary frames: the first-forward comparison keeps fp16x2, the watch sees nothing remarkable, nothing is re-armed
    assert d["chosen"].startswith("fp16x2") and len(d["checks"]) == 1 and d["rearmed"] == 0, d
    assert d["watch"] is not None and d["watch"]["tensors_seen"] > 0 and d["watch"]["worst_growth"] < 4.0, d
    left = None
    for k in range(max_frames_after_cut):
      got = e.forward(hot, want_feats=False, want_pooled=True)
      d = e.describe()["conv_split_family_auto"]
      if d["chosen"].startswith("bf16x3"):
        left = k + 1
        break
    # the scene cut: pool0's recorded |max| jumps 2^30 above the level the first comparison accepted (visible to the host one
    # forward later), the comparison is re-armed, the twin disagrees, the engine leaves fp16x2 -- and IS the bf16x3 engine from
    # that frame on
    assert left is not None and left <= max_frames_after_cut, d
    assert d["rearmed"] >= 1 and len(d["checks"]) >= 2 and d["checks"][-1]["max_rel_diff"] > d["tolerance"], d
    for a, b in zip(got, want_hot):
      assert (a is None and b is None) or np.array_equal(a, b)
    return left, d
  finally:
    m.close()


@pytest.mark.gpu
def test_continuous_range_guard_leaves_fp16x2_after_a_scene_cut(hip_lib, monkeypatch):
  """Round 6: the "auto" guard is continuous.  Small size, split kernels forced (the simulator run of this takes eight minutes:
  the CPU suite covers the host logic with a mocked statistic below, the device counters run on the GPU)."""
  monkeypatch.setenv("ODT_CONV_SPLIT_MINTILES", "1"); monkeypatch.setenv("ODT_CONV_SPLIT3_MINTILES", "1")
  cfg = small_config(resnet_num_block=[1, 1, 1, 1])
  left, d = _run_exposure_switch(hip_lib, cfg, 1, 160, 224, multi=False)
  assert left <= 4, (left, d)


def test_continuous_range_guard_host_logic(emu_lib, monkeypatch):
  """The engine's half of the continuous guard on the simulator: a recorded |max| that has grown past watch_ratio re-arms the
  fp16x2-vs-bf16x3 comparison for the next forward (twin rebuilt from the model's weights), a comparison that keeps fp16x2
  accepts the new level (rebase) so that the same stream does not re-arm again, and odt_range_health answers on a handle
  without fp16x2 launches."""
  import copy
  cfg = small_config(resnet_num_block=[1, 1, 1, 1], rpn_test_post_nms_topk=16)
  ca = copy.copy(cfg); ca.conv_split_family = "auto"
  fr = synthetic_frames(1, 64, 96, seed=5)
  m = models.get_model(ca, 0, weights=weights_for(cfg), lib=emu_lib)
  try:
    e = m.engine(1, 64, 96)
    e.forward(fr, want_feats=False, want_pooled=True)
    h = e.range_health()
    assert set(h) == {"worst_growth", "tensor", "tensor_amax", "tensors_seen"} and h["worst_growth"] <= 1.0, h   # (exact-f32 plan at this size: the preprocessed frames' record at most)
    a = e.describe()["conv_split_family_auto"]
    assert a["chosen"].startswith("fp16x2") and len(a["checks"]) == 1 and a["rearmed"] == 0 and "twin" not in e._auto and "args" in e._auto, a
    state = {"growth": 1000.0, "rebased": 0}
    def fake(self, rebase=False):
      out = {"worst_growth": state["growth"], "tensor": "pool0", "tensor_amax": 1e9, "tensors_seen": 7}
      if rebase:
        state["growth"] = 1.0; state["rebased"] += 1
      return out
    monkeypatch.setattr(type(e), "range_health", fake)
    e.forward(fr, want_feats=False, want_pooled=True)             # the watch behind this forward sees the signal ...
    a = e.describe()["conv_split_family_auto"]
    assert a["rearmed"] == 1 and a["calibration_forwards_left"] == 1 and a["watch"]["tensor"] == "pool0", a
    e.forward(fr, want_feats=False, want_pooled=True)             # ... this one runs on a fresh twin as well
    a = e.describe()["conv_split_family_auto"]
    assert len(a["checks"]) == 2 and a["calibration_forwards_left"] == 0 and a["chosen"].startswith("fp16x2"), a
    assert state["rebased"] == 1 and "twin" not in e._auto        # fp16x2 kept: the new maxima are the level to watch from
    e.forward(fr, want_feats=False, want_pooled=True)             # same maxima again: growth 1, no re-arm
    assert e.describe()["conv_split_family_auto"]["rearmed"] == 1
  finally:
    m.close()


@pytest.mark.gpu
def test_continuous_range_guard_1080p_stream_b2(hip_lib):
  from object_detection_tracking_amd.config import make_config as product_make_config
  """The same on the product default at 1080p (multi graph, b = 2): an over-exposed cut in the middle of a stream moves the
  engine to the bf16x3 handle within four frames."""
  cfg = product_make_config(rpn_test_post_nms_topk=300, im_batch_size=2, max_size=1920, short_edge_size=1080)
  left, d = _run_exposure_switch(hip_lib, cfg, 2, 1080, 1920, multi=True)
  assert left <= 4, (left, d)
