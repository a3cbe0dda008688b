"""Staged parity tests: every kernel through the C ABI (odt_op_*) against the oracle /
a plain torch fp32 reference on identical seeded inputs.

Each test runs on two builds of the same kernel sources: the product libodt_hip.so on a real
MI355X (`-m gpu`) and the HIP-on-CPU simulator build (CPU suite).  Integer / index results are
compared bit-exactly; floating point within the tolerance written in the test.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as TF

from common import torch_conv_nhwc
from object_detection_tracking_amd import ops
from object_detection_tracking_amd._lib import ODT_GRAPH_MULTI, ODT_GRAPH_SINGLE
from oracle import graph as og
from oracle import tfops
from oracle.anchors import all_anchors

F = np.float32

CONV_CASES = [
    # B, H, W, Cin, Cout, k, stride, dil, pad_t, pad_l, Ho, Wo, relu
    (1, 9, 11, 32, 64, 1, 1, 1, 0, 0, 9, 11, True),          # 1x1
    (2, 10, 13, 64, 96, 3, 1, 1, 1, 1, 10, 13, False),       # 3x3 SAME, batch
    (1, 12, 14, 32, 160, 3, 2, 1, 1, 1, 6, 7, True),         # pad T/L 1 + 3x3 s2 VALID
    (1, 13, 15, 32, 32, 3, 2, 2, 1, 1, 5, 6, True),          # dilated + strided (res5 block0)
    (1, 8, 9, 64, 15, 1, 1, 1, 0, 0, 8, 9, False),           # 15-channel RPN head
    (1, 20, 20, 32, 130, 3, 1, 2, 2, 2, 20, 20, False),      # dil 2 SAME
    (1, 11, 13, 64, 128, 1, 2, 1, 0, 0, 5, 6, False),        # cropped shortcut 1x1 s2 VALID
]
CONV_CASES_GPU = [
    (2, 68, 120, 256, 256, 3, 1, 1, 1, 1, 68, 120, True),    # res4 conv2
    (1, 68, 120, 1024, 256, 1, 1, 1, 0, 0, 68, 120, True),   # res4 conv1
    (1, 135, 240, 256, 256, 3, 1, 1, 1, 1, 135, 240, False), # FPN posthoc / RPN conv0 @P3
    (1, 272, 480, 64, 64, 3, 1, 1, 1, 1, 272, 480, True),    # res2 conv2 (N = 64 tile)
    (1, 1, 300, 12544, 1024, 1, 1, 1, 0, 0, 1, 300, True),   # fc6 as a 1x1 conv over RoIs
]


def _run_conv(lib, case, rng, tile_env=None):
  B, H, W, Cin, Cout, k, s, d, pt, pl, Ho, Wo, relu = case
  x = rng.standard_normal((B, H, W, Cin)).astype(F)
  w = (rng.standard_normal((k, k, Cin, Cout)) * np.sqrt(2.0 / (k * k * Cin))).astype(F)
  b = rng.standard_normal(Cout).astype(F)
  y = ops.conv2d(x, w, b, s, d, pt, pl, (Ho, Wo), relu=relu, lib=lib)
  # against float64, relative to the magnitude the sum was formed from: |y - y64| <= 5e-6 * (sum |a||w| + |bias|) -- what
  # the f64 tests of the split kernels justify (test_parity_report.py: 1.6e-7 ... 4.2e-7 measured); ReLU is 1-Lipschitz
  r = torch_conv_nhwc(x, w, b, s, d, pt, pl, Ho, Wo, dtype=np.float64)
  mag = torch_conv_nhwc(np.abs(x), np.abs(w), np.abs(b), s, d, pt, pl, Ho, Wo, dtype=np.float64)
  if relu:
    r = np.maximum(r, 0)
  err = np.abs(y.astype(np.float64) - r) / mag
  assert err.max() <= 5e-6, err.max()


@pytest.mark.parametrize("case", CONV_CASES)
@pytest.mark.parametrize("tile", ["0", "1", "2", "3"])
def test_conv2d(backend, case, tile, monkeypatch):
  name, lib = backend
  monkeypatch.setenv("ODT_CONV_TILE", tile)
  _run_conv(lib, case, np.random.default_rng(1))


@pytest.mark.gpu
@pytest.mark.parametrize("case", CONV_CASES_GPU)
def test_conv2d_model_shapes(hip_lib, case):
  _run_conv(hip_lib, case, np.random.default_rng(2))


def test_conv2d_residual_and_offset(backend):
  name, lib = backend
  rng = np.random.default_rng(3)
  x = rng.standard_normal((1, 10, 12, 32)).astype(F)
  w = (rng.standard_normal((1, 1, 32, 96)) * 0.2).astype(F)
  b = rng.standard_normal(96).astype(F)
  res = rng.standard_normal((1, 10, 12, 96)).astype(F)
  y = ops.conv2d(x, w, b, res=res, res_mode=1, relu=True, lib=lib)
  r = np.maximum(torch_conv_nhwc(x, w, b, 1, 1, 0, 0, 10, 12) + res, 0)
  np.testing.assert_allclose(y, r, rtol=1e-4, atol=1e-4)
  # FPN top-down: nearest-2x upsample of the coarser level added in the epilogue (nn.py:949-1004)
  up = rng.standard_normal((1, 5, 6, 96)).astype(F)
  y = ops.conv2d(x, w, b, res=up, res_mode=2, lib=lib)
  r = torch_conv_nhwc(x, w, b, 1, 1, 0, 0, 10, 12) + up.repeat(2, 1).repeat(2, 2)
  np.testing.assert_allclose(y, r, rtol=1e-4, atol=1e-4)
  # output written at offset (1,1) into a zeroed buffer (nn.py:493-497 pad after ReLU)
  y = ops.conv2d(x, w, b, out_hw=(10, 12), out_off=(1, 1), relu=True, lib=lib)
  r = np.maximum(torch_conv_nhwc(x, w, b, 1, 1, 0, 0, 10, 12), 0)
  assert np.all(y[:, 0] == 0) and np.all(y[:, :, 0] == 0)
  np.testing.assert_allclose(y[:, 1:, 1:], r, rtol=1e-4, atol=1e-4)


def test_preprocess_bit_exact(backend):
  name, lib = backend
  rng = np.random.default_rng(4)
  fr = rng.integers(0, 256, (2, 21, 37, 3), dtype=np.uint8)
  Hp, Wp = 3 + 21 + 2 + 11, 3 + 37 + 2 + 5
  for frames in (fr, fr.astype(F)):
    got = ops.preprocess(frames, 3, 3, Hp, Wp, lib=lib)
    ref = og.preprocess(frames).numpy().transpose(0, 2, 3, 1)      # NHWC
    assert np.array_equal(got[:, 3:3 + 21, 3:3 + 37, :3], ref)     # identical op order -> exact
    pad = got.copy(); pad[:, 3:24, 3:40, :3] = 0
    assert not pad.any()


def test_maxpool_bit_exact(backend):
  name, lib = backend
  import torch
  import torch.nn.functional as TF
  rng = np.random.default_rng(5)
  x = rng.standard_normal((2, 16, 22, 64)).astype(F)
  got = ops.maxpool3x3s2(x, lib=lib)
  xt = torch.from_numpy(x).permute(0, 3, 1, 2)
  ref = TF.max_pool2d(og.pad_tl(xt), 3, 2).permute(0, 2, 3, 1).numpy()
  assert np.array_equal(got, ref)


@pytest.mark.parametrize("n,k", [(10, 10), (100, 7), (5000, 64), (70000, 300), (1530, 1000), (6120, 2000), (4096, 4096),
                                 (388800, 3000)])
def test_topk_bit_exact(backend, n, k):
  name, lib = backend
  if name == "emu" and n > 6200:
    pytest.skip("large n only on the GPU")
  rng = np.random.default_rng(n + k)
  s = rng.standard_normal(n).astype(F)
  s[rng.integers(0, n, n // 4)] = s[rng.integers(0, n, n // 4)]   # exact ties
  assert np.array_equal(ops.top_k(s, k, lib=lib), tfops.top_k(s, k).astype(np.int32))


def test_topk_ties_straddling_k(backend):
  """All-equal and mostly-equal scores: the k-th value is tied; lower indices must win."""
  name, lib = backend
  for s in (np.zeros(300, F), np.r_[np.ones(5, F), np.zeros(200, F), -np.zeros(50, F)],
            np.r_[np.full(90, -1.5, F), np.full(10, 2.0, F)]):
    for k in (1, 7, 64):
      assert np.array_equal(ops.top_k(s, k, lib=lib), tfops.top_k(s, k).astype(np.int32))


def _random_boxes(rng, n, size=200.0, clustered=True):
  if clustered:
    ctr = rng.uniform(0, size, (max(1, n // 8), 2))
    c = ctr[rng.integers(0, ctr.shape[0], n)] + rng.normal(0, 6, (n, 2))
  else:
    c = rng.uniform(0, size, (n, 2))
  wh = rng.uniform(4, 60, (n, 2))
  return np.concatenate([c - wh / 2, c + wh / 2], 1).astype(F)


@pytest.mark.parametrize("n,max_out,thr", [(1, 5, 0.5), (50, 100, 0.7), (300, 300, 0.7),
                                           (1000, 100, 0.5), (1024, 1024, 0.7),
                                           # more than 1024 candidates: the paneled walk (panels of 512, select_device.hpp)
                                           (1025, 1025, 0.7), (1400, 100, 0.5), (2000, 2000, 0.7), (4096, 4096, 0.7),
                                           (3000, 3000, 0.95)])
def test_nms_indices_bit_exact(backend, n, max_out, thr):
  name, lib = backend
  if name == "emu" and n > 300 and n not in (1025, 1400):
    pytest.skip("large n only on the GPU")
  rng = np.random.default_rng(n)
  b = _random_boxes(rng, n)
  s = rng.uniform(0, 1, n).astype(F)
  s[rng.integers(0, n, n // 5 + 1)] = F(0.5)                      # score ties
  got = ops.nms(b, s, max_out, thr, lib=lib)
  ref = tfops.non_max_suppression(b, s, max_out, thr)
  assert np.array_equal(got, ref.astype(np.int32))


def test_nms_edge_cases(backend):
  """zero-area boxes (IoU := 0), flipped corners, identical boxes, touching boxes."""
  name, lib = backend
  b = np.array([[0, 0, 10, 10], [0, 0, 10, 10], [10, 10, 0, 0], [5, 5, 5, 9], [0, 0, 0, 0],
                [10, 0, 20, 10], [1, 1, 9, 9], [0, 0, 10, 5]], F)
  s = np.array([0.9, 0.9, 0.8, 0.7, 0.7, 0.6, 0.5, 0.4], F)
  for thr in (0.0, 0.3, 0.5, 0.64, 0.7):
    got = ops.nms(b, s, 8, thr, lib=lib)
    ref = tfops.non_max_suppression(b, s, 8, thr)
    assert np.array_equal(got, ref.astype(np.int32)), thr
  assert ops.nms(np.zeros((0, 4), F), np.zeros((0,), F), 5, 0.5, lib=lib).size == 0


def _proposal_inputs(rng, B, hw_levels, img_hw, neg_shift=0.0):
  strides, sizes = (4, 8, 16, 32, 64), (32, 64, 128, 256, 512)
  rpn, anchors, logits, deltas = [], [], [], []
  for (h, w), st, sz in zip(hw_levels, strides, sizes):
    lg = (rng.standard_normal((B, h, w, 3)) * 1.5 - neg_shift).astype(F)
    dl = (rng.standard_normal((B, h, w, 3, 4)) * 0.4).astype(F)
    an = all_anchors(st, [sz], (0.5, 1, 2), max(img_hw) + 64)
    rpn.append(ops.pack_rpn(lg, dl)); anchors.append(an); logits.append(lg); deltas.append(dl)
  return rpn, anchors, logits, deltas


def _oracle_proposals_single(logits, deltas, anchors, img_hw, K, thr, clip):
  ab, asc = [], []
  for lg, dl, an in zip(logits, deltas, anchors):
    h, w = lg.shape[1:3]
    dec = og.decode_bbox_target(dl[0], an[:h, :w], clip)
    b, s, _ = og.rpn_proposals_level_b1(dec, lg[0].reshape(-1), img_hw, K, thr)
    ab.append(b); asc.append(s)
  ab = np.concatenate(ab, 0); asc = np.concatenate(asc, 0)
  tk = tfops.top_k(asc, min(asc.size, K))
  return ab[tk]


def _oracle_proposals_multi(logits, deltas, anchors, img_hw, K, thr, clip):
  B = logits[0].shape[0]
  lb, ls = [], []
  for lg, dl, an in zip(logits, deltas, anchors):
    h, w = lg.shape[1:3]
    k = min(K, lg[0].size)
    bb = np.zeros((B, k, 1, 4), F); ss = np.zeros((B, k, 1), F)
    for b in range(B):
      dec = og.decode_bbox_target(dl[b], an[:h, :w], clip)
      sc = lg[b].reshape(-1)
      idx = tfops.top_k(sc, k)
      bb[b, :, 0] = og.clip_boxes(dec[idx], img_hw); ss[b, :, 0] = sc[idx]
    nb, ns, _, _ = tfops.combined_non_max_suppression(bb, ss, K, K, thr)
    lb.append(nb); ls.append(ns)
  lb = np.concatenate(lb, 1); ls = np.concatenate(ls, 1)
  out = []
  for b in range(B):
    tk = tfops.top_k(ls[b], min(ls.shape[1], K))
    pb = lb[b][tk]
    area = (pb[:, 3] - pb[:, 1]) * (pb[:, 2] - pb[:, 0])
    out.append(pb[area > 0])
  return out


@pytest.mark.parametrize("K", [16, 100])
def test_proposals_single(backend, K):
  name, lib = backend
  rng = np.random.default_rng(K)
  img_hw = (96, 128)
  levels = [(24, 32), (12, 16), (6, 8), (3, 4), (2, 2)]
  rpn, anchors, logits, deltas = _proposal_inputs(rng, 1, levels, img_hw)
  clip = float(np.log(256 / 16.0))
  props, nprops = ops.proposals(ODT_GRAPH_SINGLE, rpn, anchors, img_hw, K, 0.7, clip, lib=lib)
  ref = _oracle_proposals_single(logits, deltas, anchors, img_hw, K, 0.7, clip)
  assert nprops[0] == ref.shape[0]
  # expf on the device vs numpy: <= 2 ulp on coordinates up to ~128 px
  np.testing.assert_allclose(props[0, :nprops[0]], ref, rtol=0, atol=1e-4)


def test_proposals_multi_chunk_level(backend):
  """A level larger than one top-k chunk (32768 logits): the per-chunk winners are merged."""
  name, lib = backend
  rng = np.random.default_rng(21)
  img_hw = (512, 1024)
  levels = [(128, 256), (4, 8)]                       # 98 304 logits (3 chunks) + a small level
  K = 100
  strides, sizes = (4, 128), (32, 256)
  rpn, anchors, logits, deltas = [], [], [], []
  for (h, w), st, sz in zip(levels, strides, sizes):
    lg = (rng.standard_normal((1, h, w, 3)) * 1.5).astype(F)
    lg.reshape(-1)[rng.integers(0, lg.size, 64)] = F(4.25)      # ties across chunk borders
    dl = (rng.standard_normal((1, h, w, 3, 4)) * 0.4).astype(F)
    an = all_anchors(st, [sz], (0.5, 1, 2), 1024)
    rpn.append(ops.pack_rpn(lg, dl)); anchors.append(an); logits.append(lg); deltas.append(dl)
  clip = float(np.log(1024 / 16.0))
  props, nprops = ops.proposals(ODT_GRAPH_SINGLE, rpn, anchors, img_hw, K, 0.7, clip, lib=lib)
  ref = _oracle_proposals_single(logits, deltas, anchors, img_hw, K, 0.7, clip)
  assert nprops[0] == ref.shape[0]
  np.testing.assert_allclose(props[0, :nprops[0]], ref, rtol=0, atol=1e-3)


@pytest.mark.parametrize("neg_shift", [0.0, 2.5])
def test_proposals_multi_zero_padding_quirk(backend, neg_shift):
  """Multibatch graph: NMS output is zero-padded to K per level and the padding (score 0)
  outranks every negative logit in the cross-level top-k (reference models.py:2487-2520)."""
  name, lib = backend
  rng = np.random.default_rng(7)
  img_hw = (96, 128)
  levels = [(24, 32), (12, 16), (6, 8), (3, 4), (2, 2)]
  K = 40
  rpn, anchors, logits, deltas = _proposal_inputs(rng, 2, levels, img_hw, neg_shift)
  clip = float(np.log(256 / 16.0))
  props, nprops = ops.proposals(ODT_GRAPH_MULTI, rpn, anchors, img_hw, K, 0.7, clip, lib=lib)
  ref = _oracle_proposals_multi(logits, deltas, anchors, img_hw, K, 0.7, clip)
  for b in range(2):
    assert nprops[b] == ref[b].shape[0]
    np.testing.assert_allclose(props[b, :nprops[b]], ref[b], rtol=0, atol=1e-4)
  if neg_shift > 0:
    assert nprops.min() < K      # the quirk actually bit


def test_roi_align(backend):
  """TF crop_and_resize semantics incl. zeroed out-of-range samples, level boundaries."""
  name, lib = backend
  rng = np.random.default_rng(8)
  B, Cc = 2, 64 if name == "emu" else 256
  hw = [(24, 32), (12, 16), (6, 8), (3, 4)]
  feats = [rng.standard_normal((B, h, w, Cc)).astype(F) for h, w in hw]
  strides = (4, 8, 16, 32)
  boxes = np.array([
      [10.3, 12.7, 50.1, 60.9], [0, 0, 128, 96], [-5, -5, 20, 20], [100, 70, 140, 110],
      [0, 0, 111.9, 111.9], [0, 0, 112.1, 112.1], [0, 0, 223.9, 223.9], [0, 0, 224.2, 224.1],
      [5, 5, 5, 5], [30, 30, 31, 31], [0, 0, 127, 95], [64, 48, 127.9, 95.9]], F)
  box_ind = np.array([0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 0, 1], np.int32)
  out, pooled = ops.roi_align(feats, strides, boxes, box_ind, lib=lib)
  ref = og.multilevel_roi_align([f.transpose(0, 3, 1, 2) for f in feats], boxes, box_ind, strides)
  np.testing.assert_allclose(out, ref, rtol=1e-5, atol=2e-6)
  np.testing.assert_allclose(pooled, ref.mean(axis=(2, 3)), rtol=1e-5, atol=2e-6)
  # the 7x7 mean is formed INSIDE the ROIAlign kernel (round 5), in the order of the second pass it replaces: a sequential
  # f32 sum over the row-major window, then / 49 -- bit for bit what that pass made of the kernel's own output
  seq = np.zeros(out.shape[:2], F)
  flat = out.reshape(out.shape[0], out.shape[1], 49)
  for q in range(49):
    seq = (seq + flat[:, :, q]).astype(F)
  assert np.array_equal(pooled, (seq / F(49)).astype(F))


def _head_inputs(rng, B, K, Cn, img_hw):
  props = np.zeros((B, K, 4), F)
  for b in range(B):
    bx = _random_boxes(rng, K, size=min(img_hw), clustered=True)
    props[b] = og.clip_boxes(bx, img_hw)
  nprops = np.array([K] + [max(1, K - 7 * b) for b in range(1, B)], np.int32)
  cls = (rng.standard_normal((B * K, Cn)) * 2.0).astype(F)
  box = (rng.standard_normal((B * K, Cn, 4)) * 0.8).astype(F)
  return props, nprops, cls, box


def test_detections_single(backend):
  name, lib = backend
  rng = np.random.default_rng(9)
  K, Cn, img_hw = (48, 6, (96, 128)) if name == "emu" else (300, 15, (1080, 1920))
  props, nprops, cls, box = _head_inputs(rng, 1, K, Cn, img_hw)
  rw = np.array([10, 10, 5, 5], F)
  clip = float(np.log(1333 / 16.0))
  per_im = 20 if name == "emu" else 100
  gb, gp, gl, gv = ops.detections(ODT_GRAPH_SINGLE, cls, box, props, nprops, img_hw, rw, clip,
                                  1e-4, 0.5, per_im, lib=lib)
  dec, probs = og.head_decode(props[0], box[:, 1:], cls, img_hw, rw)
  pi, fp, _ = og.fastrcnn_predictions(dec, probs, 1e-4, per_im, 0.5)
  # Selection runs on device-computed probabilities/boxes (expf differs by <= 2 ulp from
  # numpy): compare as matched sets, then demand identical order where scores are distinct.
  r = gv[0]
  assert r == pi.shape[0]
  fb = dec[pi[:, 0], pi[:, 1]]
  np.testing.assert_allclose(gp[0, :r], fp, rtol=0, atol=5e-6)
  np.testing.assert_allclose(gb[0, :r], fb, rtol=0, atol=1e-3 * max(img_hw) / 128)
  assert np.array_equal(gl[0, :r], (pi[:, 1] + 1).astype(np.int32))


def test_detections_multi(backend):
  name, lib = backend
  rng = np.random.default_rng(10)
  B, K, Cn, img_hw = 2, 40, 5, (96, 128)
  props, nprops, cls, box = _head_inputs(rng, B, K, Cn, img_hw)
  rw = np.array([10, 10, 5, 5], F)
  clip = float(np.log(1333 / 16.0))
  per_im = 30
  gb, gp, gl, gv = ops.detections(ODT_GRAPH_MULTI, cls, box, props, nprops, img_hw, rw, clip,
                                  1e-4, 0.5, per_im, lib=lib)
  # oracle: scatter into zero-padded [B,M,...] slots + combined NMS (models.py:2924-2976)
  rows = [(b, j) for b in range(B) for j in range(nprops[b])]
  M = len(rows)
  rb = np.stack([props[b, j] for b, j in rows]); sel = [b * K + j for b, j in rows]
  dec, probs = og.head_decode(rb, box[sel][:, 1:], cls[sel], img_hw, rw)
  bidx = np.array([b for b, _ in rows])
  pbx = np.zeros((B, M, Cn - 1, 4), F); ppr = np.zeros((B, M, Cn - 1), F)
  pbx[bidx, np.arange(M)] = dec; ppr[bidx, np.arange(M)] = probs[:, 1:]
  nb, ns, nc, nv = tfops.combined_non_max_suppression(pbx, ppr, per_im, per_im, 0.5)
  assert np.array_equal(gv, nv)
  np.testing.assert_allclose(gp, ns, rtol=0, atol=5e-6)
  np.testing.assert_allclose(gb, nb, rtol=0, atol=1e-3)
  assert np.array_equal(gl, (nc + 1).astype(np.int32))


def test_detections_multi_padding_when_few_rois(backend):
  """Fewer real detections than result_per_im: zero-score slots of the other image are legal
  picks of combined_non_max_suppression (score_threshold = -inf)."""
  name, lib = backend
  rng = np.random.default_rng(11)
  B, K, Cn, img_hw = 2, 8, 3, (96, 128)
  props, nprops, cls, box = _head_inputs(rng, B, K, Cn, img_hw)
  nprops[:] = [3, 5]
  rw = np.array([10, 10, 5, 5], F)
  clip = float(np.log(1333 / 16.0))
  per_im = 12
  gb, gp, gl, gv = ops.detections(ODT_GRAPH_MULTI, cls, box, props, nprops, img_hw, rw, clip,
                                  1e-4, 0.5, per_im, lib=lib)
  rows = [(b, j) for b in range(B) for j in range(nprops[b])]
  M = len(rows)
  rb = np.stack([props[b, j] for b, j in rows]); sel = [b * K + j for b, j in rows]
  dec, probs = og.head_decode(rb, box[sel][:, 1:], cls[sel], img_hw, rw)
  bidx = np.array([b for b, _ in rows])
  pbx = np.zeros((B, M, Cn - 1, 4), F); ppr = np.zeros((B, M, Cn - 1), F)
  pbx[bidx, np.arange(M)] = dec; ppr[bidx, np.arange(M)] = probs[:, 1:]
  nb, ns, nc, nv = tfops.combined_non_max_suppression(pbx, ppr, per_im, per_im, 0.5)
  assert np.array_equal(gv, nv)
  np.testing.assert_allclose(gp, ns, rtol=0, atol=5e-6)
  np.testing.assert_allclose(gb, nb, rtol=0, atol=1e-3)
  assert np.array_equal(gl, (nc + 1).astype(np.int32))


def test_nn_cosine(backend):
  """vs the restatement of deep_sort/nn_matching.py:31-54,78-96,156-177."""
  name, lib = backend
  rng = np.random.default_rng(12)
  T, N, D = 7, 23, 256
  sizes = rng.integers(1, 6, T)
  seg = np.r_[0, np.cumsum(sizes)].astype(np.int32)
  gal = rng.standard_normal((seg[-1], D)).astype(F)
  det = rng.standard_normal((N, D)).astype(F)
  got = ops.nn_cosine(gal, seg, det, lib=lib)
  ref = np.zeros((T, N))
  for t in range(T):
    a = gal[seg[t]:seg[t + 1]]
    a = a / np.linalg.norm(a, axis=1, keepdims=True)
    b = det / np.linalg.norm(det, axis=1, keepdims=True)
    ref[t] = (1. - np.dot(a, b.T)).min(axis=0)
  assert got.dtype == np.float64
  np.testing.assert_allclose(got, ref, rtol=0, atol=2e-6)
  assert ops.nn_cosine(gal, seg, np.zeros((0, D), F), lib=lib).shape == (T, 0)


@pytest.mark.parametrize("D", [256, 132, 30])
def test_nn_cosine_long_galleries_and_odd_lengths(backend, D):
  """Round 6 kernel: 32-row blocks of whole tracks; a gallery of more than 32 rows (no budget: nn_matching.py:125-154 appends one
  row per matched frame) is cut into blocks whose minima meet through an atomic min; more than 128 detections (second column of
  workgroups); feature lengths that are not a multiple of 8 / of 4 (scalar loads)."""
  name, lib = backend
  rng = np.random.default_rng(14)
  sizes = np.array([5, 70, 1, 32, 33, 2, 31, 1, 64, 3])
  T, N = len(sizes), 150
  seg = np.r_[0, np.cumsum(sizes)].astype(np.int32)
  gal = rng.standard_normal((seg[-1], D)).astype(F)
  det = rng.standard_normal((N, D)).astype(F)
  got = ops.nn_cosine(gal, seg, det, lib=lib)
  a = gal / np.linalg.norm(gal, axis=1, keepdims=True); b = det / np.linalg.norm(det, axis=1, keepdims=True)
  full = 1. - a.astype(np.float64) @ b.astype(np.float64).T
  ref = np.stack([full[seg[t]:seg[t + 1]].min(axis=0) for t in range(T)])
  np.testing.assert_allclose(got, ref, rtol=0, atol=3e-6)


def test_nn_cosine_scratch_growth(backend):
  """The call's persistent scratch (pinned + device, CosineCtx) starts at 2^20 floats and is doubled when outgrown;
  outgrown buffers are retired, not freed (a hipFree would wait for a detector's forward in flight).  Small call,
  a call past the initial capacity, small call again: all three correct."""
  name, lib = backend
  rng = np.random.default_rng(13)
  def check(T, per, N, D):
    seg = (np.arange(T + 1) * per).astype(np.int32)
    gal = rng.standard_normal((T * per, D)).astype(F); det = rng.standard_normal((N, D)).astype(F)
    got = ops.nn_cosine(gal, seg, det, lib=lib)
    a = gal / np.linalg.norm(gal, axis=1, keepdims=True); b = det / np.linalg.norm(det, axis=1, keepdims=True)
    ref = (1. - a.astype(np.float64) @ b.astype(np.float64).T).reshape(T, per, N).min(axis=1)
    np.testing.assert_allclose(got, ref, rtol=0, atol=3e-6)
  check(5, 2, 9, 128)
  check(300, 3, 140, 1024)          # (900 + 140) x 1024 floats > 2^20
  check(4, 5, 17, 256)


# ---- randomized conv coverage: every kernel instantiation (tile x stages x loop style) ----------
def _fuzz_case(rng, lib, big, couts=None):
  B = int(rng.integers(1, 3)); k = int(rng.choice([1, 1, 3])); stride = int(rng.choice([1, 1, 2]))
  dil = int(rng.choice([1, 1, 2])) if k == 3 else 1
  lo, hi = (6, 20) if not big else (9, 70)
  H, W = int(rng.integers(lo, hi)), int(rng.integers(lo, hi))
  Cin = int(rng.choice([32, 64, 96] if not big else [32, 64, 96, 256]))
  Cout = int(rng.choice([4, 8, 15, 32, 64, 72, 128, 200] if big else [4, 15, 32, 72]))
  if couts is not None:
    Cout = int(rng.choice(couts))
  pad_t, pad_l = int(rng.integers(0, k)) * dil, int(rng.integers(0, k)) * dil
  ke = (k - 1) * dil + 1
  Ho = (H + pad_t + int(rng.integers(0, ke)) - ke) // stride + 1
  Wo = (W + pad_l + int(rng.integers(0, ke)) - ke) // stride + 1
  if Ho < 1 or Wo < 1:
    return
  oy, ox = (int(rng.integers(0, 2)), int(rng.integers(0, 2)))
  res_mode = int(rng.choice([0, 0, 1, 2])) if (oy, ox) == (0, 0) else 0
  relu = bool(rng.integers(0, 2))
  x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
  w = (rng.standard_normal((k, k, Cin, Cout)) * (1.0 / np.sqrt(k * k * Cin))).astype(np.float32)
  b = rng.standard_normal((Cout,)).astype(np.float32)
  res = None
  if res_mode == 1:
    res = rng.standard_normal((B, Ho, Wo, Cout)).astype(np.float32)
  elif res_mode == 2:
    res = rng.standard_normal((B, (Ho + 1) // 2, (Wo + 1) // 2, Cout)).astype(np.float32)
  want = torch_conv_nhwc(x, w, b, stride, dil, pad_t, pad_l, Ho, Wo)
  if res_mode == 1:
    want = want + res
  elif res_mode == 2:
    want = want + np.repeat(np.repeat(res, 2, 1), 2, 2)[:, :Ho, :Wo]
  if relu:
    want = np.maximum(want, 0)
  got = ops.conv2d(x, w, b, stride=stride, dil=dil, pad_t=pad_t, pad_l=pad_l, out_hw=(Ho, Wo),
                   out_off=(oy, ox), res=res, res_mode=res_mode, relu=relu, lib=lib)
  assert np.all(got[:, :oy] == 0) and np.all(got[:, :, :ox] == 0)
  np.testing.assert_allclose(got[:, oy:, ox:], want, rtol=2e-4, atol=2e-4)


def test_conv_fuzz_all_variants(backend, monkeypatch):
  name, lib = backend
  rng = np.random.default_rng(2024)
  n = 3 if name == "emu" else 12
  for tile in ("1", "2", "3"):
    for stages in ("1", "2"):
      for fine in ("0", "1"):
        monkeypatch.setenv("ODT_CONV_TILE", tile)
        monkeypatch.setenv("ODT_CONV_STAGES", stages)
        monkeypatch.setenv("ODT_CONV_FINE", fine)
        for _ in range(n):
          _fuzz_case(rng, lib, big=name == "hip")


def test_conv_two_sources(backend, monkeypatch):
  """conv3 + convshortcut as one K-concatenated GEMM (second source at stride 1 and 2)."""
  name, lib = backend
  rng = np.random.default_rng(7)
  for fine in ("0", "1"):
    monkeypatch.setenv("ODT_CONV_FINE", fine)
    for stride_b, Ca, Cb, Cout in ((1, 64, 64, 256), (2, 128, 256, 512), (2, 32, 96, 40)):
      B, Ho, Wo = 2, 9, 11
      Hb, Wb = (Ho, Wo) if stride_b == 1 else (2 * Ho, 2 * Wo - 1)
      a = rng.standard_normal((B, Ho, Wo, Ca)).astype(np.float32)
      b2 = rng.standard_normal((B, Hb, Wb, Cb)).astype(np.float32)
      wa = (rng.standard_normal((Ca, Cout)) / np.sqrt(Ca)).astype(np.float32)
      wb = (rng.standard_normal((Cb, Cout)) / np.sqrt(Cb)).astype(np.float32)
      bias = rng.standard_normal((Cout,)).astype(np.float32)
      want = a @ wa + b2[:, ::stride_b, ::stride_b][:, :Ho, :Wo] @ wb + bias
      got = ops.conv2d_cat(a, b2, wa, wb, bias, stride_b=stride_b, relu=True, lib=lib)
      np.testing.assert_allclose(got, np.maximum(want, 0), rtol=2e-4, atol=2e-4)


# ---- bf16x3 split path (csrc/conv_split{,1,3}.hip): f32 results through six exact bf16 products ----
SPLIT_CASES = [
    # B, H, W, Cin, Cout, k, stride, dil, pad_t, pad_l, Ho, Wo, relu
    (1, 9, 11, 64, 256, 3, 1, 1, 1, 1, 9, 11, True),         # 3x3 SAME, M = 99 (ragged tile)
    (2, 10, 13, 32, 512, 1, 1, 1, 0, 0, 10, 13, False),      # dense 1x1, two N tiles, M = 260
    (1, 12, 14, 32, 256, 3, 2, 1, 1, 1, 6, 7, True),         # pad T/L 1 + 3x3 s2 VALID (res4 block0 conv2)
    (1, 13, 15, 32, 256, 3, 2, 2, 1, 1, 5, 6, False),        # dilated + strided
    (1, 11, 13, 96, 256, 1, 2, 1, 0, 0, 5, 6, False),        # 1x1 s2 over a cropped input, 3 slices per tap
    (2, 12, 13, 64, 128, 3, 1, 1, 1, 1, 12, 13, True),       # 256 x 128 tile (res3 conv2), M = 312
    (1, 18, 17, 32, 384, 3, 1, 1, 1, 1, 18, 17, False),      # 256 x 128 tile, three N tiles, M = 306
    (1, 17, 19, 64, 64, 3, 1, 1, 1, 1, 17, 19, True),        # 256 x 64 tile (res2 conv2), M = 323
    (1, 9, 30, 256, 64, 1, 1, 1, 0, 0, 9, 30, True),         # 256 x 64 tile, dense 1x1 (res2 conv1), M = 270
    # stride-1 3x3 over several images: tiles cross image boundaries (two staged runs), borders on every side
    (3, 17, 19, 32, 256, 3, 1, 1, 1, 1, 17, 19, True),       # M = 969: 4 tiles, boundaries inside tiles 1, 2 and 3
    (2, 20, 21, 96, 128, 3, 1, 2, 2, 2, 20, 21, False),      # dilation 2 (res5 conv2 class), N = 128, six 16-channel slices
    (2, 19, 17, 32, 256, 3, 1, 1, 0, 1, 17, 17, True),       # VALID rows (no top / bottom pad), SAME columns
    (2, 20, 21, 128, 256, 1, 1, 1, 0, 0, 20, 21, True),      # dense 1x1 over four 256-row tiles, four 32-channel slices (fp16x2: each tile starts at its own slice)
]


# kernel families of the split path (conv_split_choose): "3/256", "3/128": conv_split3_kernel (8 waves, LDS-DMA weight
# stages, three-stage ring; the default) with 256- / 128-row tiles; "1": the one-stage BK = 32 loop (the 64-wide layers)
# "3/128/k3": the same with the reduction cut into three split-K ranges + split_reduce_kernel
# "3/256/nokwr": 256-row tiles with the kw-reuse kernel (conv_split3k_kernel, the default for stride-1 KH x 3 convs) off
# "2/256...": the fp16x2 kernels (conv_h2.hip: conv_h2k_kernel for the stride-1 KH x 3 convs, conv_h2_kernel otherwise; the
# stand-alone call records the input's |max| itself) where a case has an n-tile of 128 / 256 -- bf16x3 kernels elsewhere
# "2/128": the 128 x 128 4-wave tile of conv_h2_kernel (the library's choice for short reductions and few-tile layers)
SPLIT_PIPES = ["3/256", "3/256/nokwr", "3/128", "3/128/k3", "1", "2/256", "2/256/nokwr", "2/256/k3", "2/128", "2/128/k3"]


def _split_env(monkeypatch, pipe="3/256"):
  monkeypatch.setenv("ODT_CONV_SPLIT", "1")
  monkeypatch.setenv("ODT_CONV_SPLIT_MINTILES", "1")
  monkeypatch.setenv("ODT_CONV_SPLIT3_MINTILES", "1")
  monkeypatch.setenv("ODT_CONV_SPLIT_PIPE", pipe[0])
  f = pipe.split("/")
  if len(f) > 1:
    monkeypatch.setenv("ODT_CONV_SPLIT3_BM", f[1])
  else:
    monkeypatch.delenv("ODT_CONV_SPLIT3_BM", raising=False)
  monkeypatch.delenv("ODT_CONV_SPLIT3_FORCE_SPLITK", raising=False)
  monkeypatch.delenv("ODT_CONV_SPLIT3_KWR", raising=False)
  if len(f) > 2 and f[2] == "nokwr":
    monkeypatch.setenv("ODT_CONV_SPLIT3_KWR", "0")
  elif len(f) > 2:
    monkeypatch.setenv("ODT_CONV_SPLIT3_FORCE_SPLITK", f[2][1:])


@pytest.mark.parametrize("pipe", SPLIT_PIPES)
@pytest.mark.parametrize("case", SPLIT_CASES)
def test_conv2d_split(backend, case, pipe, monkeypatch):
  name, lib = backend
  _split_env(monkeypatch, pipe)
  _run_conv(lib, case, np.random.default_rng(11))


@pytest.mark.parametrize("pipe", ["3/256", "1", "2/256"])
def test_conv2d_split_matches_f32_kernel_at_f32_rounding(backend, pipe, monkeypatch):
  """The split result must sit as close to the f64 truth as the exact-f32 MFMA kernel does
  (error of an f32 dot product, not of a bf16 one), incl. operands spanning many binades."""
  name, lib = backend
  rng = np.random.default_rng(12)
  B, H, W, Cin, Cout = 1, 8, 9, 128, 256
  x = (rng.standard_normal((B, H, W, Cin)) * np.exp2(rng.integers(-12, 12, (B, H, W, Cin)))).astype(F)
  w = (rng.standard_normal((3, 3, Cin, Cout)) * np.exp2(rng.integers(-8, 8, (3, 3, Cin, Cout)))).astype(F)
  b = rng.standard_normal(Cout).astype(F)
  xp = np.pad(x.astype(np.float64), ((0, 0), (1, 1), (1, 1), (0, 0)))
  ref = np.zeros((B, H, W, Cout)); mag = np.zeros((B, H, W, Cout))
  for dy in range(3):
    for dx in range(3):
      patch = xp[:, dy:dy + H, dx:dx + W]
      ref += patch @ w[dy, dx].astype(np.float64)
      mag += np.abs(patch) @ np.abs(w[dy, dx].astype(np.float64))
  ref += b
  monkeypatch.setenv("ODT_CONV_SPLIT", "0")
  y32 = ops.conv2d(x, w, b, 1, 1, 1, 1, (H, W), lib=lib)
  _split_env(monkeypatch, pipe)
  ysp = ops.conv2d(x, w, b, 1, 1, 1, 1, (H, W), lib=lib)
  e32 = np.max(np.abs(y32 - ref) / mag)
  esp = np.max(np.abs(ysp - ref) / mag)
  # both carry the f32 accumulation error of a K = 1152 dot product (~1e-6 of sum|a||b| on this
  # data); the split adds at most 2^-23 for its dropped piece products.  A single bf16 product
  # would be at ~4e-3.
  assert esp <= 1.5 * e32 + 1.2e-7, (esp, e32)
  assert esp < 4e-6, esp


def test_conv2d_fp16x2_dynamic_range(backend, monkeypatch):
  """fp16x2 pieces scale the A operand by ONE power of two per tensor (from its |max|): rows down to 2^-15 of the tensor
  maximum keep the f32-rounding-level error of the other kernels; below that the lo piece runs into the f16 subnormal grid
  and the error is bounded in ABSOLUTE terms by 2^-39 max|A| per unit weight -- the bound DESIGN.md states."""
  name, lib = backend
  _split_env(monkeypatch, "2/256")
  rng = np.random.default_rng(21)
  K, N, G = 256, 256, 8
  x = np.maximum(rng.standard_normal((1, G, 40, K)), 0).astype(F) + F(0.01)
  for g in range(G):
    x[0, g] *= F(2.0 ** (-5 * g))
  w = (rng.standard_normal((1, 1, K, N)) * np.sqrt(2.0 / K)).astype(F)
  b = np.zeros(N, F)
  y = ops.conv2d(x, w, b, 1, 1, 0, 0, (G, 40), lib=lib)
  ref = x.astype(np.float64) @ w[0, 0].astype(np.float64)
  mag = np.abs(x.astype(np.float64)) @ np.abs(w[0, 0].astype(np.float64))
  wsum = np.abs(w[0, 0].astype(np.float64)).sum(axis=0)
  amax = float(np.abs(x).max())
  err = np.abs(y - ref)
  assert np.all(err <= 4e-7 * mag + 2.0 ** -38 * amax * wsum)
  for g in range(4):                    # rows within 2^-15 of the maximum: the f32-level bound alone
    assert np.max(err[0, g] / mag[0, g]) < 4e-7, g


@pytest.mark.parametrize("pipe", ["2/256", "2/256/nokwr", "2/128"])
def test_conv2d_fp16x2_power_of_two_equivariance(backend, pipe, monkeypatch):
  """The fp16x2 kernels scale their A operand by a power of two taken from the tensor's recorded |max|: multiplying the
  whole input by 2^k moves that power by -k and leaves every f16 piece unchanged, so the (bias-free, linear) result is the
  old one times 2^k EXACTLY -- for inputs from 2^-40 to 2^+40 of the reference scale.  (A size-independent property of the
  range plumbing: a stale or missing range record breaks it at once.)"""
  name, lib = backend
  _split_env(monkeypatch, pipe)
  rng = np.random.default_rng(31)
  x = rng.standard_normal((2, 13, 15, 64)).astype(F)
  w = (rng.standard_normal((3, 3, 64, 256)) * 0.05).astype(F)
  b = np.zeros(256, F)
  y0 = ops.conv2d(x, w, b, 1, 1, 1, 1, (13, 15), lib=lib)
  assert np.isfinite(y0).all() and np.abs(y0).max() > 0
  for k in (-40, -7, 5, 40):
    yk = ops.conv2d(x * F(2.0 ** k), w, b, 1, 1, 1, 1, (13, 15), lib=lib)
    assert np.array_equal(yk, y0 * F(2.0 ** k)), k


@pytest.mark.parametrize("pipe", SPLIT_PIPES)
def test_conv2d_split_output_offset_and_residual(backend, pipe, monkeypatch):
  name, lib = backend
  _split_env(monkeypatch, pipe)
  rng = np.random.default_rng(13)
  x = rng.standard_normal((1, 10, 12, 64)).astype(F)
  w = (rng.standard_normal((3, 3, 64, 256)) * 0.05).astype(F)
  b = rng.standard_normal(256).astype(F)
  got = ops.conv2d(x, w, b, 1, 1, 1, 1, (10, 12), out_off=(1, 1), relu=True, lib=lib)
  want = np.maximum(torch_conv_nhwc(x, w, b, 1, 1, 1, 1, 10, 12), 0)
  assert np.all(got[:, :1] == 0) and np.all(got[:, :, :1] == 0)
  np.testing.assert_allclose(got[:, 1:, 1:], want, rtol=1e-4, atol=1e-4)
  # same-shape residual (added in the epilogue by conv_split3_kernel, accumulator start value in the older loops)
  res = rng.standard_normal((1, 10, 12, 256)).astype(F)
  got = ops.conv2d(x, w, b, 1, 1, 1, 1, (10, 12), res=res, res_mode=1, relu=True, lib=lib)
  np.testing.assert_allclose(got, np.maximum(torch_conv_nhwc(x, w, b, 1, 1, 1, 1, 10, 12) + res, 0),
                             rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("pipe", SPLIT_PIPES)
def test_conv2d_split_residual_bottleneck_conv3(backend, pipe, monkeypatch):
  """1x1 conv + same-shape residual + ReLU over two N tiles and a ragged M (res4 conv3 shape class);
  then the nearest-2x upsampled residual of an FPN lateral (odd sizes: the coarse level is ceil(n/2))."""
  name, lib = backend
  _split_env(monkeypatch, pipe)
  rng = np.random.default_rng(14)
  x = rng.standard_normal((2, 9, 11, 256)).astype(F)
  w = (rng.standard_normal((1, 1, 256, 512)) / 16).astype(F)
  b = rng.standard_normal(512).astype(F)
  res = rng.standard_normal((2, 9, 11, 512)).astype(F)
  got = ops.conv2d(x, w, b, res=res, res_mode=1, relu=True, lib=lib)
  want = np.maximum(torch_conv_nhwc(x, w, b, 1, 1, 0, 0, 9, 11) + res, 0)
  np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-4)
  for cout in (128, 64):               # the 256 x 128 and 256 x 64 tiles
    w2 = (rng.standard_normal((1, 1, 256, cout)) / 16).astype(F)
    r2 = rng.standard_normal((2, 9, 11, cout)).astype(F)
    got = ops.conv2d(x, w2, b[:cout], res=r2, res_mode=1, relu=True, lib=lib)
    want = np.maximum(torch_conv_nhwc(x, w2, b[:cout], 1, 1, 0, 0, 9, 11) + r2, 0)
    np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-4)
  up = rng.standard_normal((2, 5, 6, 512)).astype(F)
  got = ops.conv2d(x, w, b, res=up, res_mode=2, lib=lib)
  want = torch_conv_nhwc(x, w, b, 1, 1, 0, 0, 9, 11) + np.repeat(np.repeat(up, 2, 1), 2, 2)[:, :9, :11]
  np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("pipe", SPLIT_PIPES)
def test_conv2d_split_two_sources(backend, pipe, monkeypatch):
  """Stage-entry fusion on the split kernel: conv3(t2) + convshortcut(x[::s]) as one K-concatenated
  GEMM, second source at stride 1 and 2, all three tile configurations."""
  name, lib = backend
  _split_env(monkeypatch, pipe)
  rng = np.random.default_rng(15)
  for stride_b, Ca, Cb, Cout in ((1, 64, 64, 256), (2, 128, 256, 512), (2, 32, 96, 128), (1, 64, 32, 64)):
    B, Ho, Wo = 2, 9, 11
    Hb, Wb = (Ho, Wo) if stride_b == 1 else (2 * Ho, 2 * Wo - 1)
    a = rng.standard_normal((B, Ho, Wo, Ca)).astype(F)
    b2 = rng.standard_normal((B, Hb, Wb, Cb)).astype(F)
    wa = (rng.standard_normal((Ca, Cout)) / np.sqrt(Ca)).astype(F)
    wb = (rng.standard_normal((Cb, Cout)) / np.sqrt(Cb)).astype(F)
    bias = rng.standard_normal((Cout,)).astype(F)
    want = a @ wa + b2[:, ::stride_b, ::stride_b][:, :Ho, :Wo] @ wb + bias
    got = ops.conv2d_cat(a, b2, wa, wb, bias, stride_b=stride_b, relu=True, lib=lib)
    np.testing.assert_allclose(got, np.maximum(want, 0), rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize("pipe", SPLIT_PIPES)
def test_conv_fuzz_split(backend, pipe, monkeypatch):
  """Randomized shapes / strides / dilations / pads / output offsets / residual modes through the
  tile configurations of each split kernel family."""
  name, lib = backend
  _split_env(monkeypatch, pipe)
  rng = np.random.default_rng(2025)
  for _ in range(5 if name == "emu" else 40):
    _fuzz_case(rng, lib, big=name == "hip", couts=[64, 128, 192, 256, 384])


@pytest.mark.parametrize("pipe", ["3/256", "3/128", "3/128/k3"])
def test_conv2d_split_16wide_stage_extras(backend, pipe, monkeypatch):
  """BK = 16 stage kernels: 96- and 160-channel sources (odd numbers of 16-channel slices), residual, second
  source at stride 2."""
  name, emu_lib = backend
  _split_env(monkeypatch, pipe)
  rng = np.random.default_rng(16)
  # residual (same shape, 2x upsampled) and the K-concatenated second source
  x = rng.standard_normal((2, 9, 11, 96)).astype(F)
  w = (rng.standard_normal((1, 1, 96, 256)) / 10).astype(F)
  b = rng.standard_normal(256).astype(F)
  res = rng.standard_normal((2, 9, 11, 256)).astype(F)
  got = ops.conv2d(x, w, b, res=res, res_mode=1, relu=True, lib=emu_lib)
  np.testing.assert_allclose(got, np.maximum(torch_conv_nhwc(x, w, b, 1, 1, 0, 0, 9, 11) + res, 0), rtol=1e-4, atol=1e-4)
  a = rng.standard_normal((2, 9, 11, 64)).astype(F)
  b2 = rng.standard_normal((2, 18, 21, 160)).astype(F)
  wa = (rng.standard_normal((64, 256)) / 8).astype(F)
  wb = (rng.standard_normal((160, 256)) / 12).astype(F)
  want = a @ wa + b2[:, ::2, ::2][:, :9, :11] @ wb + b
  got = ops.conv2d_cat(a, b2, wa, wb, b, stride_b=2, relu=False, lib=emu_lib)
  np.testing.assert_allclose(got, want, rtol=2e-4, atol=2e-4)


# ---- fused bottleneck tail: conv2 (3x3) -> conv3 (1x1) inside the 3x3 kernel (conv_h2k_kernel<.., FUSE>) -----------------
def _bottleneck_ref(x, w2, b2, w3, b3, res, dil, relu3):
  """float64 restatement of nn.py:503-521's tail: conv2 + bias + ReLU (rounded to f32, as the tensor the reference holds
  between the two ops), conv3 + bias (+ shortcut), ReLU; and the magnitude sum the error is measured against."""
  xt = torch.from_numpy(x).double().permute(0, 3, 1, 2)
  y = torch.nn.functional.conv2d(xt, torch.from_numpy(w2).double().permute(3, 2, 0, 1), torch.from_numpy(b2).double(),
                                 padding=dil, dilation=dil).relu().float().double()
  w3d, b3d = torch.from_numpy(w3).double(), torch.from_numpy(b3).double()
  z = torch.einsum("bchw,cn->bhwn", y, w3d) + b3d
  mag = torch.einsum("bchw,cn->bhwn", y.abs(), w3d.abs()) + b3d.abs()
  if res is not None:
    z = z + torch.from_numpy(res).double(); mag = mag + torch.from_numpy(res).double().abs()
  if relu3:
    z = z.relu()
  return z.numpy(), mag.numpy()


@pytest.mark.parametrize("case", [
    (1, 75, 93, 3),       # 35 x 44 conv pixels (odd height: the last conv row lies outside every pool window), 17 x 22 pooled: tiles partial in y AND x, three workgroups
    (2, 53, 61, 2),       # two images, 12 x 14 pooled: 2 x 2 x 2 tiles (partial in y), two workgroups: four tiles each
    (1, 37, 140, 0),      # one tile row, odd width, one workgroup per tile
])
def test_stem_kernel_vs_two_launches_and_f64(backend, case):
  """conv_stem_kernel (conv0 + ReLU + pool0 in one persistent kernel) at geometries the model never produces -- pooled sizes
  that are not multiples of the 8 x 7 tile, odd conv maps: BIT-IDENTICAL to conv_h2_kernel + maxpool3x3s2_kernel, and both
  at the fp16x2 error level against f64."""
  name, lib = backend
  B, Hp, Wp, grid = case
  rng = np.random.default_rng(Hp * 1000 + Wp)
  x = rng.uniform(-2.2, 2.7, (B, Hp, Wp, 3)).astype(F)         # (the normalised pixel range of build_preprocess)
  w = (rng.standard_normal((7, 7, 3, 64)) * np.sqrt(2.0 / 147)).astype(F)
  b = (rng.standard_normal(64) * 0.1).astype(F)
  got = {f: ops.stem(x, w, b, fuse=f, grid=grid, lib=lib) for f in (False, True)}
  np.testing.assert_array_equal(got[True], got[False])
  xt = torch.from_numpy(x.astype(np.float64)).permute(0, 3, 1, 2)
  wt = torch.from_numpy(w.astype(np.float64)).permute(3, 2, 0, 1)
  conv = torch.relu(TF.conv2d(xt, wt, torch.from_numpy(b.astype(np.float64)), stride=2))
  mag = TF.conv2d(xt.abs(), wt.abs(), torch.from_numpy(np.abs(b).astype(np.float64)), stride=2)
  pool = lambda t: TF.max_pool2d(TF.pad(t, (1, 0, 1, 0)), 3, 2)
  ref, mg = pool(conv).permute(0, 2, 3, 1).numpy(), pool(mag).permute(0, 2, 3, 1).numpy()
  assert got[True].shape == ref.shape
  assert np.all(np.abs(got[True] - ref) <= 4e-6 * mg + 1e-6), float(np.max(np.abs(got[True] - ref) / (mg + 1e-6)))


BOTTLENECK_CASES = [
    # B, H, W, C3, dil, residual, relu3 (, C)
    (2, 11, 13, 128, 1, True, True, 128), # res3's shape class: 256 x 128 tile, the K halves are 64 wide
    (1, 19, 27, 256, 1, True, True, 64),  # res2's shape class: 256 x 64 tile (K halves 32 wide: four half-steps per chunk), M = 513
    (2, 8, 9, 64, 2, False, True, 64),    # ... dilation 2, two chunks, a tile across the image boundary, no residual
    (1, 12, 24, 64, 1, True, True),       # two tiles, the second partial
    (2, 9, 17, 128, 1, True, True),       # a tile that crosses the image boundary (two runs), odd sizes
    (1, 16, 16, 192, 2, False, False),    # dilation 2, six 32-column chunks, no residual, no activation
]


@pytest.mark.parametrize("case", BOTTLENECK_CASES)
def test_bottleneck_tail_fused_vs_f64(backend, case):
  """conv3 evaluated from conv2's accumulators inside the 3x3 kernel: error relative to sum |y||w| at the level of the two
  launches it replaces (both held to 4e-7: three exact f16 products per MAC + f32 accumulation), and the two forms agree
  to the f32 rounding of one more summation order."""
  name, lib = backend
  B, H, W, C3, dil, with_res, relu3 = case[:7]
  C = case[7] if len(case) > 7 else 256
  rng = np.random.default_rng(B * 1000 + H * 10 + C3)
  # post-ReLU input with a log-normal spread over pixels (what conv1 hands to conv2)
  x = (np.maximum(rng.standard_normal((B, H, W, C)), 0) * np.exp(rng.standard_normal((B, H, W, 1)))).astype(F)
  w2 = (rng.standard_normal((3, 3, C, C)) * np.sqrt(2.0 / (9 * C))).astype(F)
  b2 = (rng.standard_normal(C) * 0.1).astype(F)
  w3 = (rng.standard_normal((C, C3)) * np.sqrt(2.0 / C)).astype(F)
  b3 = (rng.standard_normal(C3) * 0.1).astype(F)
  res = rng.standard_normal((B, H, W, C3)).astype(F) if with_res else None
  ref, mag = _bottleneck_ref(x, w2, b2, w3, b3, res, dil, relu3)
  got = {}
  for fuse in (False, True):
    got[fuse] = ops.bottleneck_tail(x, w2, b2, w3, b3, res=res, dil=dil, relu3=relu3, fuse=fuse, lib=lib)
    err = np.abs(got[fuse] - ref) / mag
    assert err.max() < 4e-7, (fuse, err.max())
  assert np.max(np.abs(got[True] - got[False]) / mag) < 3e-7


def test_bottleneck_tail_fused_row_scale(backend):
  """The fused form splits conv3's operand with a power of two PER PIXEL ROW (and K half), taken from the row's own |max|:
  every row keeps the f32-level error relative to ITS OWN magnitude sum, whatever else the tile holds.  (conv2's operand is
  still scaled per tensor, so the rows here stay within 2^-14 of the input's maximum -- the domain of
  test_conv2d_fp16x2_dynamic_range's first bound.)"""
  name, lib = backend
  C, C3 = 256, 64
  rng = np.random.default_rng(77)
  x = np.maximum(rng.standard_normal((1, 15, 17, C)), 0).astype(F) + F(0.01)
  for r in range(15):
    x[0, r] *= F(2.0 ** (-r))             # image rows 2^0 ... 2^-14 below the maximum
  w2 = (rng.standard_normal((3, 3, C, C)) * np.sqrt(2.0 / (9 * C))).astype(F)
  w2[0] = 0; w2[2] = 0                    # 1 x 3 taps: an output row only sees its own input row (and magnitude)
  b2 = np.zeros(C, F)
  w3 = (rng.standard_normal((C, C3)) * np.sqrt(2.0 / C)).astype(F)
  b3 = np.zeros(C3, F)
  ref, mag = _bottleneck_ref(x, w2, b2, w3, b3, None, 1, False)
  y = ops.bottleneck_tail(x, w2, b2, w3, b3, dil=1, relu3=False, fuse=True, lib=lib)
  err = np.abs(y - ref) / mag
  for r in range(15):
    assert err[0, r].max() < 4e-7, (r, err[0, r].max())


def test_conv2d_fp16x2_512x64_tile(backend, monkeypatch):
  """conv_h2k_kernel<2, ., ., WN = 1>: the 64-wide stride-1 3x3 layers (res2 conv2) on 512 x 64 tiles, eight waves stacked
  along M (a 64 x 64 wave tile: 24 MFMAs per stage instead of 12).  Same K order and products as the 256 x 64 tile: the
  results are bit-identical to it; tiles cross image boundaries (two staged runs), the last one is partial."""
  name, lib = backend
  _split_env(monkeypatch, "2/256")
  rng = np.random.default_rng(64)
  B, H, W, C = 3, 23, 24, 64                 # Ho Wo = 552 >= 512, M = 1656: four tiles, boundaries inside tiles 1 and 2
  x = rng.standard_normal((B, H, W, C)).astype(F)
  w = (rng.standard_normal((3, 3, C, 64)) * 0.06).astype(F)
  b = rng.standard_normal(64).astype(F)
  ref = np.maximum(torch_conv_nhwc(x, w, b, 1, 1, 1, 1, H, W), 0)
  out = {}
  for mode in ("0", "2"):
    monkeypatch.setenv("ODT_CONV_H2_N64_BM512", mode)
    out[mode] = ops.conv2d(x, w, b, 1, 1, 1, 1, (H, W), relu=True, lib=lib)
    np.testing.assert_allclose(out[mode], ref, rtol=1e-4, atol=1e-4)
  assert np.array_equal(out["0"], out["2"])
  # dilation 2 (the other halo width)
  monkeypatch.setenv("ODT_CONV_H2_N64_BM512", "2")
  y = ops.conv2d(x, w, b, 1, 2, 2, 2, (H, W), relu=False, lib=lib)
  np.testing.assert_allclose(y, torch_conv_nhwc(x, w, b, 1, 2, 2, 2, H, W), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("case", [
    (2, 20, 30, 256, 64, 1, 1, 1, 0, 0, 20, 30, True),       # dense 1x1 (res2 conv1), M = 1200: three tiles, the last partial
    (1, 43, 47, 32, 64, 3, 2, 1, 1, 1, 22, 24, True),        # strided taps (conv0's class), M = 528
    (1, 9, 11, 64, 64, 1, 1, 1, 0, 0, 9, 11, False),         # a single partial tile, K = 64 (two stages)
])
def test_conv2d_fp16x2_512x64_tile_generic_kernel(backend, case, monkeypatch):
  """conv_h2_kernel<2, 8, ., WN = 1>: the 64-wide layers outside the kw-reuse kernel (conv0, res2 conv1) on 512 x 64 tiles of
  eight waves stacked along M; same products as the 128 x 64 tile -- bit-identical to it where the K order is the same
  (the single-source 1x1 layers start their reduction at slice `tile index mod slices`: summation order differs)."""
  name, lib = backend
  _split_env(monkeypatch, "2")
  B, H, W, Cin, Cout, k, stride, dil, pt, pl, Ho, Wo, relu = case
  rng = np.random.default_rng(65)
  x = rng.standard_normal((B, H, W, Cin)).astype(F)
  w = (rng.standard_normal((k, k, Cin, Cout)) * 0.08).astype(F)
  b = rng.standard_normal(Cout).astype(F)
  ref = torch_conv_nhwc(x, w, b, stride, dil, pt, pl, Ho, Wo)
  if relu:
    ref = np.maximum(ref, 0)
  out = {}
  for mode in ("0", "2"):
    monkeypatch.setenv("ODT_CONV_H2_N64_BM512", mode)
    out[mode] = ops.conv2d(x, w, b, stride, dil, pt, pl, (Ho, Wo), relu=relu, lib=lib)
    np.testing.assert_allclose(out[mode], ref, rtol=1e-4, atol=1e-4)
  if k > 1:
    assert np.array_equal(out["0"], out["2"])
  else:
    np.testing.assert_allclose(out["0"], out["2"], rtol=0, atol=2e-6 * float(np.abs(ref).max()))


@pytest.mark.parametrize("case", [
    (1, 10, 15, 512, 256, 1, 1, 1, 0, 0, 10, 15, True),      # dense 1x1, M = 150: 3 x 2 tiles of 64 x 128 (res4 conv1 at b = 1)
    (1, 10, 15, 256, 512, 1, 1, 1, 0, 0, 10, 15, False),     # dense 1x1, short reduction, several n-tiles (res4 conv3 at b = 1)
    (1, 9, 15, 64, 128, 1, 1, 1, 0, 0, 9, 15, True),         # dense 1x1, K = 64: below conv_h2d_kernel's two double stages, conv_h2_kernel under both settings
    (1, 9, 15, 128, 128, 1, 1, 1, 0, 0, 9, 15, True),        # dense 1x1, K = 128: exactly two double stages (the peeled steps only, no steady-state step)
    (1, 10, 15, 64, 256, 3, 1, 1, 1, 1, 10, 15, True),       # 3x3 taps on the generic kernel
    (1, 20, 15, 64, 128, 1, 2, 1, 0, 0, 10, 8, False),       # strided, one n-tile, a partial last tile (M = 80)
])
def test_conv2d_fp16x2_64x128_tile_without_splitk(backend, case, monkeypatch):
  """conv_h2_kernel<2, 1>: layers whose 128 x 128 tiles cannot fill the chip but whose 64 x 128 ones can (b = 1 below res3)
  run on two-wave 64 x 128 tiles WITHOUT split-K (no partial slabs, no combine pass); the residual path too."""
  name, lib = backend
  _split_env(monkeypatch, "2")
  monkeypatch.setenv("ODT_CONV_SPLIT3_MINTILES", "3")     # (test sizes: t128 = 2 .. 2 < 3 <= t64)
  B, H, W, Cin, Cout, k, stride, dil, pt, pl, Ho, Wo, relu = case
  rng = np.random.default_rng(66)
  x = rng.standard_normal((B, H, W, Cin)).astype(F)
  w = (rng.standard_normal((k, k, Cin, Cout)) * 0.05).astype(F)
  b = rng.standard_normal(Cout).astype(F)
  res = rng.standard_normal((B, Ho, Wo, Cout)).astype(F)
  ref = torch_conv_nhwc(x, w, b, stride, dil, pt, pl, Ho, Wo) + res
  if relu:
    ref = np.maximum(ref, 0)
  out = {}
  for mode in ("0", "1", "3"):              # (3: 64 x 64 tiles, conv_h2_kernel<1, 1>)
    monkeypatch.setenv("ODT_CONV_H2_BM64", mode)
    out[mode] = ops.conv2d(x, w, b, stride, dil, pt, pl, (Ho, Wo), res=res, res_mode=1, relu=relu, lib=lib)
    np.testing.assert_allclose(out[mode], ref, rtol=1e-4, atol=2e-4)
  for mode in ("1", "3"):
    np.testing.assert_allclose(out["0"], out[mode], rtol=0, atol=3e-6 * float(np.abs(ref).max()))
  # the dense 1x1 reductions take conv_h2d_kernel (double stages: two BK = 32 sub-stages per barrier, round 5) on these tiles;
  # ODT_CONV_H2_BK64=0 keeps conv_h2_kernel: the same products in another slice order -> equal at f32 rounding level
  if k == 1 and stride == 1 and Cin % 64 == 0:
    t128 = -(-(B * Ho * Wo) // 128) * -(-Cout // 128)
    monkeypatch.setenv("ODT_CONV_SPLIT3_MINTILES", str(t128 + 1))    # (the 128-row tiles do not "fill the chip", the 64-row ones do)
    r64 = torch_conv_nhwc(x, w, b, stride, dil, pt, pl, Ho, Wo, dtype=np.float64) + res
    if relu:
      r64 = np.maximum(r64, 0)
    for mode in ("1", "3"):
      monkeypatch.setenv("ODT_CONV_H2_BM64", mode)
      both = {}
      for bk in ("1", "0"):
        monkeypatch.setenv("ODT_CONV_H2_BK64", bk)
        both[bk] = ops.conv2d(x, w, b, stride, dil, pt, pl, (Ho, Wo), res=res, res_mode=1, relu=relu, lib=lib)
        assert np.abs(both[bk] - r64).max() <= 4e-6 * float(np.abs(r64).max()), (mode, bk)
      np.testing.assert_allclose(both["1"], both["0"], rtol=0, atol=3e-6 * float(np.abs(ref).max()))
      if Cin >= 256:
        assert not np.array_equal(both["1"], both["0"]), mode   # (another summation order: the double-stage kernel really ran)


@pytest.mark.parametrize("case", [
    (1, 20, 26, 64, 256, 3, 1, 1, 1, 1, 20, 26, True),       # M = 520: 3 x 2 tiles of 256 x 128, six (slice, kh) groups cut in two
    (2, 17, 16, 96, 256, 3, 1, 2, 2, 2, 17, 16, False),      # dilation 2, a tile across the image boundary, nine groups cut in two
])
def test_conv2d_fp16x2_kw_reuse_kernel_with_splitk(backend, case, monkeypatch):
  """conv_h2k_kernel with split-K: the stride-1 KH x 3 layers of few rows (res3 / res4 conv2 at b = 1) on 256 x 128 tiles of
  the kw-reuse kernel, the (32-channel slice, kh) groups cut into ranges, raw partial tiles combined in range order by
  split_reduce_kernel (which applies the scales, bias, activation and records the output's range)."""
  name, lib = backend
  _split_env(monkeypatch, "2")
  B, H, W, Cin, Cout, k, stride, dil, pt, pl, Ho, Wo, relu = case
  monkeypatch.setenv("ODT_CONV_SPLIT3_MINTILES", "8")
  rng = np.random.default_rng(67)
  x = rng.standard_normal((B, H, W, Cin)).astype(F)
  w = (rng.standard_normal((k, k, Cin, Cout)) * 0.05).astype(F)
  b = rng.standard_normal(Cout).astype(F)
  ref = torch_conv_nhwc(x, w, b, stride, dil, pt, pl, Ho, Wo)
  if relu:
    ref = np.maximum(ref, 0)
  out = {}
  for mode in ("0", "1"):
    monkeypatch.setenv("ODT_CONV_H2K_SPLITK", mode)
    out[mode] = ops.conv2d(x, w, b, stride, dil, pt, pl, (Ho, Wo), relu=relu, lib=lib)
    np.testing.assert_allclose(out[mode], ref, rtol=1e-4, atol=2e-4)
  np.testing.assert_allclose(out["0"], out["1"], rtol=0, atol=3e-6 * float(np.abs(ref).max()))
