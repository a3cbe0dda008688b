"""EfficientDet path, backbone (detector half -- in progress)."""
import numpy as np
import pytest

from object_detection_tracking_amd.efficientdet import arch


def _params_with_head(name):
  shapes = arch.backbone_variable_shapes(name)
  n = sum(int(np.prod(s)) for k, s in shapes.items() if "moving" not in k)
  width, _ = arch.efficientnet_params(name)
  last = arch.backbone_spec(name)["blocks"][-1]["cout"]
  head = arch.round_filters(1280, width)
  return n + last * head + 2 * head + head * 1000 + 1000        # head conv + BN + FC(1000)


def test_architecture_arithmetic_matches_published_parameter_counts():
  """The published EfficientNet sizes (Tan & Le 2019, table 2 / the official model cards: B0 5.3M,
  B1 7.8M, B2 9.2M, B3 12M, B4 19M, B5 30M, B6 43M, B7 66M parameters) pin filter rounding, block
  repeats, SE widths and variable shapes of backbone_variable_shapes()."""
  want = {"efficientnet-b0": 5288548, "efficientnet-b1": 7794184, "efficientnet-b2": 9109994,
          "efficientnet-b3": 12233232, "efficientnet-b4": 19341616, "efficientnet-b5": 30389784,
          "efficientnet-b6": 43040704, "efficientnet-b7": 66347960}     # exact counts of the official models
  for name, n in want.items():
    assert _params_with_head(name) == n, (name, _params_with_head(name), n)


def test_reduction_levels_and_strides():
  sp = arch.backbone_spec("efficientnet-b6")
  red = {b["reduction"]: b for b in sp["blocks"] if b["reduction"]}
  assert sorted(red) == [1, 2, 3, 4, 5]
  assert [red[l]["cout"] for l in (3, 4, 5)] == [72, 200, 576]     # P3..P5 inputs of EfficientDet-D6/D7
  assert sp["stem"] == 56 and len(sp["blocks"]) == 45


def _backbone_parity(lib, name, H, W, B=1, tol=2e-5):
  from object_detection_tracking_amd.efficientdet import EfficientNetBackbone, synthetic_backbone_weights
  from object_detection_tracking_amd.weights import synthetic_frames
  from oracle import effnet
  w = synthetic_backbone_weights(name, 0)
  fr = synthetic_frames(B, H, W, seed=5)
  taps = {}
  ref = effnet.backbone_forward(name, w, effnet.preprocess(fr), taps)
  net = EfficientNetBackbone(name, w, B, H, W, lib=lib)
  try:
    got = net.features(fr)
    stem = net.tap("stem")
    rs = taps["stem"].transpose(0, 2, 3, 1)
    assert np.abs(stem[..., :rs.shape[-1]] - rs).max() <= tol * max(1.0, np.abs(rs).max())
    assert np.all(stem[..., rs.shape[-1]:] == 0)                       # pad channels stay zero
    for lvl in (1, 2, 3, 4, 5):
      r = ref[lvl].transpose(0, 2, 3, 1)
      assert got[lvl].shape == r.shape, (lvl, got[lvl].shape, r.shape)
      err = np.abs(got[lvl] - r).max() / max(1e-6, np.abs(r).max())
      assert err < tol * 10, (lvl, err)
    got2 = net.features(fr)                                            # bit-deterministic
    for lvl in got:
      assert np.array_equal(got[lvl], got2[lvl])
  finally:
    net.close()


def test_backbone_b0_parity(backend):
  name, lib = backend
  if name == "emu":
    _backbone_parity(lib, "efficientnet-b0", 64, 96)
  else:
    _backbone_parity(lib, "efficientnet-b0", 250, 333, B=2)           # odd sizes: SAME pads both ways


def test_backbone_mbconv_expand_dw_in_one_kernel(backend, monkeypatch):
  """MBConv's expand 1x1 + BN + swish -> depthwise + BN + swish (+ squeeze sums) as ONE kernel with the expanded tensor in
  LDS (csrc/effnet_mbconv.hip; reference efficientnet_model.py:162-330), forced onto every block (ODT_EFFDET_FUSE_MB=2:
  k = 3 / 5, stride 1 / 2, tiles cut by the image border, maps smaller than one 16 x 16 patch): against the oracle at the
  unfused path's tolerance, and against the unfused handle itself."""
  from object_detection_tracking_amd.efficientdet import EfficientNetBackbone, synthetic_backbone_weights
  from object_detection_tracking_amd.weights import synthetic_frames
  name, lib = backend
  net_name, B, H, W = ("efficientnet-b0", 1, 70, 100) if name == "emu" else ("efficientnet-b0", 2, 250, 333)
  w = synthetic_backbone_weights(net_name, 0)
  fr = synthetic_frames(B, H, W, seed=5)
  feats = {}
  for mode in ("2", "0"):
    monkeypatch.setenv("ODT_EFFDET_FUSE_MB", mode)
    if mode == "2":
      _backbone_parity(lib, net_name, H, W, B=B)
    net = EfficientNetBackbone(net_name, w, B, H, W, lib=lib)
    try:
      d = net.describe()
      assert (d["mbconv_expand_dw_fused"] >= 10) == (mode == "2"), d
      feats[mode] = net.features(fr)
    finally:
      net.close()
  for lvl in feats["0"]:
    a, b = feats["2"][lvl], feats["0"][lvl]
    assert np.abs(a - b).max() <= 2e-5 * max(1e-6, np.abs(b).max()), lvl
  if name == "hip":
    monkeypatch.setenv("ODT_EFFDET_FUSE_MB", "2")
    _backbone_parity(lib, "efficientnet-b6", 384, 512)


@pytest.mark.gpu
def test_backbone_b6_parity_512(hip_lib):
  """The backbone of EfficientDet-D6/D7 (efficientdet_wrapper.py:566-587)."""
  _backbone_parity(hip_lib, "efficientnet-b6", 384, 512)


def _det_parity(lib, model, H, W, tol=3e-5):
  """Feature network + class / box nets against the oracle (per-level logits)."""
  import torch
  from object_detection_tracking_amd.efficientdet import EfficientNetBackbone
  from object_detection_tracking_amd.weights import synthetic_frames
  from oracle import effnet
  c = arch.det_config(model)
  w = arch.synthetic_det_weights(model, 0)
  fr = synthetic_frames(1, H, W, seed=9)
  red = effnet.backbone_forward(c["backbone"], w, effnet.preprocess(fr))
  taps = {}
  fpn = effnet.feature_network(model, w, {l: torch.from_numpy(red[l]) for l in (3, 4, 5)}, (H, W), taps)
  ref = effnet.class_box_nets(model, w, fpn)
  from object_detection_tracking_amd.efficientdet import generate_anchors
  wa = dict(w); wa["effdet/anchors"] = generate_anchors(H, W, c["anchor_scale"])
  net = EfficientNetBackbone(c["backbone"], wa, 1, H, W, lib=lib, det=model, topk=100)
  try:
    net.forward_async(fr); net.synchronize()
    F_ = c["fpn_num_filters"]
    def rel(a, b):
      return np.abs(a - b).max() / max(1e-6, np.abs(b).max())
    n0 = net.tap("cell0_fnode0")[..., :F_]
    assert rel(n0, taps["cell0_fnode0"].transpose(0, 2, 3, 1)) < tol * 10
    for lvl in range(3, 8):
      f = net.tap("fpn_%d" % lvl)[..., :F_]
      assert rel(f, fpn[lvl].numpy().transpose(0, 2, 3, 1)) < tol * 30, lvl
      cl = net.tap("class_%d" % lvl); bx = net.tap("box_%d" % lvl)
      assert rel(cl[..., :ref[lvl][0].shape[-1]], ref[lvl][0]) < tol * 30, lvl
      assert rel(bx[..., :36], ref[lvl][1]) < tol * 30, lvl
  finally:
    net.close()


def test_efficientdet_architecture_matches_published_sizes():
  """EfficientDet-D0 3.9M / D7 52M parameters (Tan et al. 2020, table 1) pin the feature-network and
  head variable tables."""
  def count(model):
    c = arch.det_config(model)
    shapes = dict(arch.backbone_variable_shapes(c["backbone"])); shapes.update(arch.det_variable_shapes(model))
    return sum(int(np.prod(s)) for k, s in shapes.items() if "moving" not in k) / 1e6
  assert abs(count("efficientdet-d0") - 3.9) < 0.05
  assert abs(count("efficientdet-d7") - 52.0) < 0.5


def test_efficientdet_d0_nets_parity(backend):
  name, lib = backend
  # every pyramid level must shrink in both dimensions (efficientdet_arch.py:196-199 raises otherwise)
  _det_parity(lib, "efficientdet-d0", 136, 152 if name == "emu" else 200)


@pytest.mark.gpu
def test_efficientdet_d1_nets_parity_odd_size(hip_lib):
  """88 filters (channel padding to 96) and sizes that are not multiples of 128 (nearest resize
  with a non-integer ratio, asymmetric 'SAME' pads)."""
  _det_parity(hip_lib, "efficientdet-d1", 270, 350)


def _det_e2e(lib, model, H, W, topk, score_thr=0.02, tol_box=2e-2, src_hw=None, partial=None, wmod=None):
  """Full EfficientDet forward through get_model / Session.run against the oracle."""
  import torch
  from object_detection_tracking_amd import models
  from common import make_config
  from object_detection_tracking_amd.weights import synthetic_frames
  from oracle import effnet
  c = arch.det_config(model)
  w = arch.synthetic_det_weights(model, 0)
  if wmod is not None:
    wmod(w)
  if src_hw is None:
    fr = synthetic_frames(1, H, W, seed=13)[0]
    x, scale = effnet.preprocess(fr[None]), 1.0
  else:                                         # frame of another size: device-side resize + pad
    fr = synthetic_frames(1, src_hw[0], src_hw[1], seed=13)[0]
    x, scale = effnet.preprocess_resized(fr, (H, W))
  red = effnet.backbone_forward(c["backbone"], w, x)
  fpn = effnet.feature_network(model, w, {l: torch.from_numpy(red[l]) for l in (3, 4, 5)}, (H, W))
  cb = effnet.class_box_nets(model, w, fpn)
  rb, rs, rc, rl, dbg = effnet.detect(model, cb, (H, W), image_scale=scale, topk=topk, score_thr=score_thr,
                                      partial_class_idxs=partial)
  cfg = make_config(is_efficientdet=True, efficientdet_modelname=model, efficientdet_max_detection_topk=topk,
                    short_edge_size=H, max_size=W, threshold_conf=score_thr)
  cfg.max_size = W; cfg.result_score_thres = score_thr
  if partial:
    cfg.use_partial_classes = True; cfg.partial_class_idxs = list(partial)
  m = models.get_model(cfg, 0, weights=w, lib=lib)
  try:
    boxes, labels, probs, feats = models.Session().run(
        [m.final_boxes, m.final_labels, m.final_probs, m.fpn_box_feat], feed_dict=m.get_feed_dict_forward(fr))
    assert len(boxes) == len(rb) and len(boxes) > 3, (len(boxes), len(rb))
    # near-equal scores of random-init heads may swap two candidates (1e-6 logit differences):
    # compare as sets first, element-wise when the order is identical
    from common import match_detections, tie_swaps
    miss, extra = match_detections(boxes, labels, probs, rb, rc, rs, tol_box, 2e-5)
    # a random-init class head puts whole grids of overlapping candidates on one score plateau (differences of 1e-7):
    # which of two such NMS competitors survives is below the f32 noise of ANY implementation; those swaps are counted
    # apart, everything else keeps the budget
    ties = tie_swaps(boxes, labels, probs, rb, rc, rs, tol_box, 2e-5)
    assert (miss - ties) + (extra - ties) <= max(2, len(rb) // 25), (miss, extra, ties)
    if miss + extra > 0 or not np.array_equal(labels, rc):
      return
    if np.abs(boxes - rb).max() > tol_box:       # same set, two near-equal scores in swapped order
      return
    np.testing.assert_allclose(probs, rs, rtol=0, atol=2e-5)
    # fpn_box_feat: ROIAlign mean on each box's own level (restated with the oracle's crop_and_resize)
    from oracle import graph as og
    want = np.zeros((len(rb), c["fpn_num_filters"]), np.float32)
    for i in range(len(rb)):
      f = fpn[int(rl[i])].numpy()
      bf = (rb[i:i + 1] * np.float32(1.0 / 2 ** int(rl[i]))).astype(np.float32)
      want[i] = og.roi_align(f, bf, np.zeros((1,), np.int32), 7).mean(axis=(2, 3))[0]
    assert feats.shape == want.shape
    np.testing.assert_allclose(feats, want, rtol=0, atol=5e-5 * max(1.0, np.abs(want).max()))
  finally:
    m.close()


def test_efficientdet_d0_end_to_end(backend):
  name, lib = backend
  _det_e2e(lib, "efficientdet-d0", 136, 152 if name == "emu" else 200, topk=300 if name == "emu" else 1000)


@pytest.mark.parametrize("mode", ["split_forced", "f32_32ch", "gate_pass"])
def test_efficientdet_d0_arithmetic_and_stride_modes(backend, mode, monkeypatch):
  """The two ways the EfficientDet plan can run its 1x1 convs: every one of them forced onto the bf16x3 split kernels
  (channel counts that are not multiples of 64 -- 40, 72, 144, 240, 432 ... -- go through the padded n-tile: zero weight
  rows, zero bias, 64-channel tensor strides), and the exact-f32 kernel with 32-channel strides (ODT_EFFDET_SPLIT=0)."""
  name, lib = backend
  if mode == "split_forced":
    monkeypatch.setenv("ODT_CONV_SPLIT_MINTILES", "1"); monkeypatch.setenv("ODT_CONV_SPLIT3_MINTILES", "1")
  elif mode == "gate_pass":         # squeeze-excite gate as a pass over the activations instead of folded into the weights
    monkeypatch.setenv("ODT_EFFDET_WSCALE", "0")
  else:
    monkeypatch.setenv("ODT_EFFDET_SPLIT", "0")
  _backbone_parity(lib, "efficientnet-b0", 64, 96)
  _det_e2e(lib, "efficientdet-d0", 136, 152 if name == "emu" else 200, topk=300 if name == "emu" else 1000)


def test_efficientdet_d0_topk_threshold_ties(backend):
  """Exact score ties at the top-k threshold: with a zero class-predict kernel every anchor position carries the same
  logit per (anchor shape, class) channel, so the k-th score is shared by hundreds of candidates and the selection has to
  fall back on the index order (tf.nn.top_k: lowest index first) -- the radix select's index passes, which the other
  tests (distinct scores: the passes leave at once) never run."""
  name, lib = backend
  def zero_class_kernel(w):
    for k in ("class_net/class-predict/pointwise_kernel", "class_net/class-predict/depthwise_kernel"):
      w[k] = np.zeros_like(w[k])
    rng = np.random.default_rng(3)
    w["class_net/class-predict/bias"] = rng.uniform(-4.0, -1.0, w["class_net/class-predict/bias"].shape).astype(np.float32)
  _det_e2e(lib, "efficientdet-d0", 136, 152, topk=300, wmod=zero_class_kernel)


def test_efficientdet_d0_partial_classes(emu_lib):
  """--use_partial_classes on the EfficientDet path (efficientdet_wrapper.py:243-250, 402-410):
  labels are 1..len(partial) in the order of partial_class_idxs."""
  _det_e2e(emu_lib, "efficientdet-d0", 136, 152, topk=200, partial=[0, 2, 7, 44, 89])


def test_efficientdet_d0_resized_input(backend):
  """A 16:9 frame into a square network input: scale 0.8 down / 1.25 up, zero padding at the bottom,
  boxes multiplied by image_scale_to_original (efficientdet_wrapper.py:45-60)."""
  name, lib = backend
  _det_e2e(lib, "efficientdet-d0", 144, 144, topk=300, src_hw=(108, 180), tol_box=4e-2)
  if name == "hip":
    _det_e2e(lib, "efficientdet-d0", 256, 256, topk=1000, src_hw=(135, 200), tol_box=4e-2)   # upscaling


@pytest.mark.gpu
def test_efficientdet_d0_end_to_end_512(hip_lib):
  """EfficientDet-D0 at its native 512x512 with the reference's top-k of 5000."""
  _det_e2e(hip_lib, "efficientdet-d0", 512, 512, topk=5000)


@pytest.mark.gpu
def test_efficientdet_d2_end_to_end_odd(hip_lib):
  _det_e2e(hip_lib, "efficientdet-d2", 300, 420, topk=2000)


def test_efficientdet_flops_match_published():
  """Tan et al. 2020, table 1: EfficientDet-D0 2.5 B, D7 325 B multiply-adds at 512 / 1536 pixels --
  pins the spatial sizes and operator list of the traffic / FLOP model used by the bench."""
  _, f0 = arch.algorithmic_traffic_and_flops("efficientdet-d0", 512, 512)
  _, f7 = arch.algorithmic_traffic_and_flops("efficientdet-d7", 1536, 1536)
  assert abs(f0 / 2e9 - 2.5) < 0.1 and abs(f7 / 2e9 - 325) < 5


def test_anchor_known_answers():
  """EfficientDet-D0 @512: 49 104 anchors (the count every port of the model quotes); the first
  anchor is the 32-pixel square centred on the first stride-8 cell; product generator == oracle."""
  from object_detection_tracking_amd.efficientdet import generate_anchors
  from oracle import effnet
  a = generate_anchors(512, 512, 4.0)
  assert a.shape == (49104, 4) and a.dtype == np.float32
  assert np.array_equal(a[0], np.array([-12, -12, 20, 20], np.float32))
  np.testing.assert_allclose(a[1], [4 - 32 * 0.7 / 2, 4 - 32 * 1.4 / 2, 4 + 32 * 0.7 / 2, 4 + 32 * 1.4 / 2], rtol=1e-6)
  assert np.array_equal(a, effnet.generate_anchors((512, 512), 4.0))
  assert generate_anchors(1536, 1536, 5.0).shape == (441936, 4)          # D7


def _det_full_parity(lib, model, H, W, topk, thr, gain, stem_tol=2e-5, stage_tol=1e-3):
  """Stage taps (stem, first BiFPN node, pyramid levels, per-level class / box logits) + the final detections as a matched
  set, of one EfficientDet forward through get_model().predict(), against ONE pass of oracle.effnet."""
  import torch
  from common import match_detections, tie_swaps
  from object_detection_tracking_amd import models
  from common import make_config
  from object_detection_tracking_amd.weights import synthetic_frames
  from oracle import effnet
  c = arch.det_config(model)
  w = arch.synthetic_det_weights(model, 0, gain=gain)
  fr = synthetic_frames(1, H, W, seed=13)[0]
  btaps, ftaps = {}, {}
  red = effnet.backbone_forward(c["backbone"], w, effnet.preprocess(fr[None]), btaps)
  fpn = effnet.feature_network(model, w, {l: torch.from_numpy(red[l]) for l in (3, 4, 5)}, (H, W), ftaps)
  cb = effnet.class_box_nets(model, w, fpn)
  rb, rs, rc, rl, _ = effnet.detect(model, cb, (H, W), image_scale=1.0, topk=topk, score_thr=thr)
  cfg = make_config(is_efficientdet=True, efficientdet_modelname=model, efficientdet_max_detection_topk=topk,
                    short_edge_size=H, max_size=W, threshold_conf=thr)
  cfg.max_size = W; cfg.result_score_thres = thr
  m0 = models.get_model(cfg, 0, weights=w, lib=lib)            # production handle: activations in the arena
  try:
    prod = m0.predict(fr)
  finally:
    m0.close()
  cfg.keep_taps = True                                         # debug handle: stage taps; must agree bit for bit
  m = models.get_model(cfg, 0, weights=w, lib=lib)
  try:
    boxes, labels, probs, feats = m.predict(fr)
    for a, b in zip(prod, (boxes, labels, probs, feats)):
      assert np.array_equal(a, b), "arena and keep_taps handles disagree"
    e = m.engine((H, W))
    F_ = c["fpn_num_filters"]
    def rel(a, b):
      return float(np.abs(a - b).max() / max(1e-6, np.abs(b).max()))
    errs = {}
    rs_ = btaps["stem"].transpose(0, 2, 3, 1)
    errs["stem"] = rel(e.tap("stem")[..., :rs_.shape[-1]], rs_)
    n0 = ftaps["cell0_fnode0"].transpose(0, 2, 3, 1)
    errs["cell0_fnode0"] = rel(e.tap("cell0_fnode0")[..., :F_], n0)
    for lvl in range(3, 8):
      errs["fpn_%d" % lvl] = rel(e.tap("fpn_%d" % lvl)[..., :F_], fpn[lvl].numpy().transpose(0, 2, 3, 1))
      errs["class_%d" % lvl] = rel(e.tap("class_%d" % lvl)[..., :cb[lvl][0].shape[-1]], cb[lvl][0])
      errs["box_%d" % lvl] = rel(e.tap("box_%d" % lvl)[..., :36], cb[lvl][1])
    # 45 MBConv blocks + 8 BiFPN cells + 5-layer heads of f32 arithmetic in two summation orders: stage errors grow with
    # depth; the bound is 30x the per-layer tolerance of the small-model tests (3e-5), as _det_parity uses for D0 / D1
    assert errs["stem"] < stem_tol, errs
    assert max(errs.values()) < stage_tol, errs
    assert len(boxes) == len(rb) and len(boxes) > 3, (len(boxes), len(rb))
    assert np.isfinite(boxes).all() and np.isfinite(feats).all()
    miss, extra = match_detections(boxes, labels, probs, rb, rc, rs, 5e-2, 5e-5)
    ties = tie_swaps(boxes, labels, probs, rb, rc, rs, 5e-2, 5e-5)
    assert (miss - ties) + (extra - ties) <= max(2, len(rb) // 25), (miss, extra, ties, errs)
    print("%s @%dx%d parity:" % (model, H, W) + " stage rel errors %s; detections %d, unmatched %d/%d (%d tie swaps)" %
          ({k: "%.1e" % v for k, v in errs.items()}, len(boxes), miss, extra, ties))
  finally:
    m.close()


def test_efficientdet_full_parity_d0_small(emu_lib):
  """The D7 test's checker on a size the simulator finishes."""
  _det_full_parity(emu_lib, "efficientdet-d0", 136, 152, topk=300, thr=0.02, gain=1.0)


@pytest.mark.gpu
def test_efficientdet_d7_1536_parity(hip_lib):
  """BASELINE config #5 at its own size: EfficientDet-D7 (EfficientNet-B6 backbone, 384 filters, 8 BiFPN cells of 'sum'
  nodes, 5-layer heads) at 1536 x 1536, the reference's top-k of 5000, on the weights bench.py runs (VERDICT round 2:
  the tests stopped at D2 / B6-512)."""
  _det_full_parity(hip_lib, "efficientdet-d7", 1536, 1536, topk=5000, thr=0.02, gain=arch.bench_gain("efficientdet-d7"))


def test_efficientdet_levels_merged_into_one_launch(backend, monkeypatch):
  """Class / box nets with the five pyramid levels of a layer in ONE depthwise and ONE pointwise launch (batch 1, layers
  on the conv_split3 kernels: D7 by itself, here D1 -- 88 filters padded to 128 -- with the tile threshold lowered): the
  per-level BatchNorm is applied as per-row-range scale / bias in the conv epilogue.  Against the oracle (stage taps +
  detections) and against the one-launch-per-level plan."""
  name, lib = backend
  monkeypatch.setenv("ODT_CONV_SPLIT_MINTILES", "1"); monkeypatch.setenv("ODT_CONV_SPLIT3_MINTILES", "1")
  H, W = (136, 152) if name == "emu" else (264, 328)
  _det_full_parity(lib, "efficientdet-d1", H, W, topk=300 if name == "emu" else 1000, thr=0.02, gain=1.0)
  # launch counts: merged vs per level
  from object_detection_tracking_amd import models
  from common import make_config
  from object_detection_tracking_amd.weights import synthetic_frames
  w = arch.synthetic_det_weights("efficientdet-d1", 0)
  fr = synthetic_frames(1, H, W, seed=13)[0]
  res = {}
  for mode in ("1", "0"):
    monkeypatch.setenv("ODT_EFFDET_MERGE_LEVELS", mode)
    cfg = make_config(is_efficientdet=True, efficientdet_modelname="efficientdet-d1", efficientdet_max_detection_topk=300,
                      short_edge_size=H, max_size=W, threshold_conf=0.02)
    cfg.max_size = W; cfg.result_score_thres = 0.02
    m = models.get_model(cfg, 0, weights=w, lib=lib)
    try:
      det = m.predict(fr)
      from object_detection_tracking_amd.models import _Engine
      res[mode] = (det, _Engine.describe(m.engine((H, W)))["conv_launches"])
    finally:
      m.close()
  assert res["0"][1] - res["1"][1] == 4 * 2 * (3 + 1), (res["0"][1], res["1"][1])     # 5 -> 1 launches per layer: 3 repeats + predict, two nets
  from common import match_detections
  miss, extra = match_detections(res["1"][0][0], res["1"][0][1], res["1"][0][2], res["0"][0][0], res["0"][0][1], res["0"][0][2], 5e-2, 5e-5)
  assert miss + extra <= 2, (miss, extra)


def test_efficientdet_predict_stream_frames_in_flight(backend):
  """EfficientDet.predict_stream (round 6): consecutive frames on replica handles, results in frame order and equal to predict()
  frame by frame."""
  name, lib = backend
  from object_detection_tracking_amd import models
  from object_detection_tracking_amd.config import make_config
  from object_detection_tracking_amd.weights import synthetic_frames
  model = "efficientdet-d0"
  S = 128 if name == "emu" else 256
  cfg = make_config(is_efficientdet=True, efficientdet_modelname=model, efficientdet_max_detection_topk=200, short_edge_size=S, max_size=S)
  cfg.max_size = S
  m = models.get_model(cfg, 0, weights=arch.synthetic_det_weights(model, 0, gain=arch.bench_gain(model)), lib=lib)
  try:
    frames = [synthetic_frames(1, S, S, seed=40 + i)[0] for i in range(4)]
    want = [m.predict(f) for f in frames]
    got = list(m.predict_stream(frames, in_flight=3))
    assert len(got) == len(want)
    for g, w in zip(got, want):
      for a, b in zip(g, w):
        assert np.array_equal(a, b)
    assert m.engine((S, S), replica=1) is not m.engine((S, S))
  finally:
    m.close()
