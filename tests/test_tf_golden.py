"""Consumers of tests/golden/tf_ref.npz -- outputs of TensorFlow itself (tests/golden/make_tf_golden.py, to be run on a
host that has TF 1.15: the build container has none and no network).

While the file is absent every `tf_ref` test SKIPS and says so: the detector's parity is then "with the CPU restatement"
(oracle/tfops.py, oracle/graph.py), DESIGN.md section 4.  Once the file is committed the same tests pin the oracle, the
simulator build of the kernels and -- under `-m gpu` -- the HIP kernels against TensorFlow.

So that the consumers themselves cannot rot while the file is absent, `test_consumers_selfcheck_*` runs them on a
stand-in fixture of the SAME schema whose "TF outputs" come from the oracle (this proves the plumbing and the kernel
<-> oracle agreement on those inputs, not TF parity -- the test names say so).
"""
import os

import numpy as np
import pytest

from oracle import tfops

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TF_REF = os.path.join(G, "tf_ref.npz")
F = np.float32


def _tf_ref():
  if not os.path.exists(TF_REF):
    pytest.skip("tests/golden/tf_ref.npz absent (generate it with tests/golden/make_tf_golden.py on a TF-1.15 host): "
                "TF parity stays UNPINNED, the oracle is a restatement")
  return dict(np.load(TF_REF, allow_pickle=False))


# ---------------------------------------------------------------------------------------------- consumers
def check_nms(g, lib=None):
  from object_detection_tracking_amd import ops
  for i in range(int(g["nms_count"])):
    b, s = g["nms%d_boxes" % i], g["nms%d_scores" % i]
    k, thr = int(g["nms%d_args" % i][0]), float(g["nms%d_args" % i][1])
    want = list(g["nms%d_idx" % i])
    if lib is None:
      got = list(tfops.non_max_suppression(b, s, k, thr))
    elif len(s) > 1024:
      continue                                   # the stand-alone kernel entry point takes <= 1024 candidates
    else:
      got = list(ops.nms(b, s, k, thr, lib=lib))
    # tf <= 1.15 has no index tie-break among EQUAL scores; where the fixture has ties compare as sets of boxes
    if len(np.unique(s)) == len(s):
      assert got == want, ("nms", i)
    else:
      assert len(got) == len(want) and np.array_equal(np.sort(s[got])[::-1], np.sort(s[want])[::-1]), ("nms ties", i)


def check_combined_nms(g, lib=None):
  from object_detection_tracking_amd import ops
  for i in range(int(g["cnms_count"])):
    bx, sc = g["cnms%d_boxes" % i], g["cnms%d_scores" % i]
    pc, tot, thr, sthr = g["cnms%d_args" % i]
    pc, tot = int(pc), int(tot)
    wb, ws, wc, wv = (g["cnms%d_out_%s" % (i, n)] for n in ("boxes", "scores", "classes", "valid"))
    if lib is None:
      nb, ns, nc, nv = tfops.combined_non_max_suppression(bx, sc, pc, tot, float(thr), score_threshold=float(sthr))
      assert list(nv) == list(wv), ("cnms valid", i)
      for b in range(bx.shape[0]):
        v = int(wv[b])
        np.testing.assert_array_equal(ns[b, :v], ws[b, :v], err_msg="cnms %d scores" % i)
        np.testing.assert_array_equal(nb[b, :v], wb[b, :v], err_msg="cnms %d boxes" % i)
        assert list(nc[b, :v].astype(int)) == list(wc[b, :v].astype(int)), ("cnms classes", i)
      continue
    if pc < tot:
      continue                                   # the kernels take ONE cap for per-class and total size
    for b in range(bx.shape[0]):                 # per image, single-image candidate rule (score > threshold)
      st = float(sthr) if np.isfinite(sthr) else -1e30
      ob, os_, oc, ov = ops.class_nms(0, bx[b:b + 1], sc[b:b + 1], tot, float(thr), score_thresh=st, lib=lib)
      v = int(wv[b])
      assert int(ov[0]) == v, ("cnms kernel valid", i, b)
      np.testing.assert_array_equal(os_[0, :v], ws[b, :v], err_msg="cnms %d kernel scores" % i)
      np.testing.assert_array_equal(ob[0, :v], wb[b, :v], err_msg="cnms %d kernel boxes" % i)
      assert list(oc[0, :v]) == list(wc[b, :v].astype(int)), ("cnms kernel classes", i, b)


def check_crop_and_resize(g, lib=None):
  from object_detection_tracking_amd import ops
  for i in range(int(g["car_count"])):
    img, bb, ind, crop = g["car%d_image" % i], g["car%d_boxes" % i], g["car%d_ind" % i], g["car%d_crop" % i]
    want = g["car%d_out" % i]
    if lib is None:
      got = tfops.crop_and_resize(img, bb, ind, int(crop[0]))
      np.testing.assert_allclose(got, want, rtol=0, atol=2e-6 * max(1.0, float(np.abs(want).max())), err_msg="car %d" % i)
      continue
    if int(crop[0]) != 14 or img.shape[3] % 4 != 0:
      continue
    # roi_align == 2x2 average of the 14x14 crop (reference nn.py:1229-1335).  Feature-pixel boxes from the normalised
    # ones by inverting transform_fpcoor_for_tf (nn.py:1258-1271); stride 1 and small boxes keep every RoI on level 0.
    B, H, W, Cc = img.shape
    sp_x = (bb[:, 3] - bb[:, 1]) * F(W - 1) / F(13.0); sp_y = (bb[:, 2] - bb[:, 0]) * F(H - 1) / F(13.0)
    x0 = bb[:, 1] * F(W - 1) - sp_x / 2 + F(0.5); y0 = bb[:, 0] * F(H - 1) - sp_y / 2 + F(0.5)
    boxes = np.stack([x0, y0, x0 + sp_x * 14, y0 + sp_y * 14], 1).astype(F)
    area = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
    keep = (area > 1e-6) & (np.sqrt(np.maximum(area, 0)) < 100.0)       # level 0 of fpn_map_rois_to_levels
    feats = [img] + [np.zeros((B, 2, 2, Cc), F)] * 3
    out, _ = ops.roi_align(feats, [1.0, 2.0, 4.0, 8.0], boxes[keep], ind[keep], lib=lib)
    ref = want[keep].reshape(-1, 7, 2, 7, 2, Cc).mean(axis=(2, 4)).transpose(0, 3, 1, 2)
    np.testing.assert_allclose(out, ref, rtol=0, atol=2e-5 * max(1.0, float(np.abs(ref).max())), err_msg="car %d kernel" % i)


def check_topk(g, lib=None):
  from object_detection_tracking_amd import ops
  for i in range(int(g["topk_count"])):
    x, k = g["topk%d_x" % i], int(g["topk%d_k" % i])
    got = tfops.top_k(x, k) if lib is None else ops.top_k(x, k, lib=lib)
    assert list(got) == list(g["topk%d_indices" % i]), ("topk", i)      # tf.nn.top_k: ties by lower index
    np.testing.assert_array_equal(x[got], g["topk%d_values" % i])


def check_conv(g, lib=None):
  from object_detection_tracking_amd import ops
  from oracle import graph as og
  import torch
  for i in list(range(int(g["conv_count"]))) + ["same"]:
    if i == "same":
      x, w, want = g["convsame_x"], g["convsame_w"], g["convsame_out"]
      stride, dil = 2, 1
      ho, wo = want.shape[1:3]
      pt = max((ho - 1) * 2 + 3 - x.shape[1], 0) // 2; pl = max((wo - 1) * 2 + 3 - x.shape[2], 0) // 2     # extra cell at the bottom / right
    else:
      x, w, want = g["conv%d_x" % i], g["conv%d_w" % i], g["conv%d_out" % i]
      stride, dil, pt, _, pl, _ = (int(v) for v in g["conv%d_args" % i])
      ho, wo = want.shape[1:3]
    tol = 2e-5 * float(np.abs(want).max())
    if lib is None:
      from common import torch_conv_nhwc
      got = torch_conv_nhwc(x, w, None, stride, dil, pt, pl, ho, wo)
    else:
      got = ops.conv2d(x, w, None, stride, dil, pt, pl, (ho, wo), lib=lib)
    np.testing.assert_allclose(got, want, rtol=0, atol=tol, err_msg="conv %s" % i)


def check_softmax(g):
  np.testing.assert_allclose(tfops.softmax(g["softmax0_x"]), g["softmax0_out"], rtol=3e-7, atol=1e-9)


def check_model(g, lib, tol_box=1e-3):
  """The reference's Mask_RCNN_FPN graph run by TensorFlow vs oracle.graph and vs the library (boxes within 1e-3 px,
  labels equal: BASELINE.json's bar)."""
  if "model_final_boxes" not in g:
    pytest.skip("tf_ref.npz was generated without --reference: no model fixture")
  from common import match_detections
  from object_detection_tracking_amd import models
  from common import make_config
  from object_detection_tracking_amd.weights import synthetic_weights
  from oracle.graph import OracleModel
  H, W, topk, seed = (int(v) for v in g["model_config"])
  cfg = make_config(rpn_test_post_nms_topk=topk, short_edge_size=H, max_size=W, resnet_num_block=[int(v) for v in g["model_blocks"]])
  w = synthetic_weights(cfg, seed)
  fr = g["model_frame"]
  wb, wl, wp = g["model_final_boxes"], g["model_final_labels"], g["model_final_probs"]
  ref = OracleModel(cfg, w).forward(fr)
  miss, extra = match_detections(ref["final_boxes"], ref["final_labels"], ref["final_probs"], wb, wl, wp, tol_box, 1e-4)
  assert miss == 0 and extra == 0, ("oracle vs TF", miss, extra)
  if lib is not None:
    m = models.get_model(cfg, 0, weights=w, lib=lib)
    try:
      boxes, labels, probs, feats = m.predict(fr)
    finally:
      m.close()
    miss, extra = match_detections(boxes, labels, probs, wb, wl, wp, tol_box, 1e-4)
    assert miss == 0 and extra == 0, ("library vs TF", miss, extra)
    if boxes.shape == wb.shape and np.array_equal(labels, wl):
      np.testing.assert_allclose(feats, g["model_fpn_box_feat"], rtol=0, atol=5e-4 * float(np.abs(g["model_fpn_box_feat"]).max()))


# ---------------------------------------------------------------------------------------------- TF-pinned tests
def test_tf_ref_oracle():
  g = _tf_ref()
  check_nms(g); check_combined_nms(g); check_crop_and_resize(g); check_topk(g); check_conv(g); check_softmax(g)


def test_tf_ref_kernels(backend):
  g = _tf_ref()
  name, lib = backend
  check_nms(g, lib); check_combined_nms(g, lib); check_crop_and_resize(g, lib); check_topk(g, lib); check_conv(g, lib)


def test_tf_ref_model(backend):
  g = _tf_ref()
  name, lib = backend
  check_model(g, lib)


# ---------------------------------------------------------------------------------------------- self-check of the consumers
def _standin_fixture():
  """The schema of tf_ref.npz with the oracle's answers in the output slots (NOT TensorFlow's)."""
  rng = np.random.default_rng(5)
  g = {}
  def clustered(n):
    ctr = rng.uniform(0, 300, (n, 2)); wh = rng.uniform(20, 100, (n, 2))
    return np.concatenate([ctr - wh / 2, ctr + wh / 2], 1).astype(F)
  b, s = clustered(200), rng.standard_normal(200).astype(F)
  g["nms0_boxes"], g["nms0_scores"], g["nms0_args"] = b, s, np.array([50, 0.7])
  g["nms0_idx"] = np.asarray(tfops.non_max_suppression(b, s, 50, 0.7), np.int32); g["nms_count"] = np.int32(1)
  bx = np.stack([np.stack([clustered(40) for _ in range(5)], 1) for _ in range(2)], 0).astype(F)
  sc = (rng.uniform(0, 1, (2, 40, 5)) ** 3).astype(F)
  nb, ns, nc, nv = tfops.combined_non_max_suppression(bx, sc, 20, 20, 0.5, score_threshold=-np.inf)
  g.update({"cnms0_boxes": bx, "cnms0_scores": sc, "cnms0_args": np.array([20, 20, 0.5, -np.inf]), "cnms0_out_boxes": nb,
            "cnms0_out_scores": ns, "cnms0_out_classes": nc, "cnms0_out_valid": nv.astype(np.int32), "cnms_count": np.int32(1)})
  img = rng.standard_normal((2, 17, 23, 8)).astype(F)
  y1 = rng.uniform(-0.1, 0.8, 20); x1 = rng.uniform(-0.1, 0.8, 20)
  bb = np.stack([y1, x1, y1 + rng.uniform(0.05, 0.5, 20), x1 + rng.uniform(0.05, 0.5, 20)], 1).astype(F)
  ind = rng.integers(0, 2, 20).astype(np.int32)
  g.update({"car0_image": img, "car0_boxes": bb, "car0_ind": ind, "car0_crop": np.array([14, 14], np.int32),
            "car0_out": tfops.crop_and_resize(img, bb, ind, 14), "car_count": np.int32(1)})
  x = rng.standard_normal(3000).astype(F)
  ix = np.asarray(tfops.top_k(x, 100), np.int32)
  g.update({"topk0_x": x, "topk0_k": np.int32(100), "topk0_indices": ix, "topk0_values": x[ix], "topk_count": np.int32(1)})
  from common import torch_conv_nhwc
  x = rng.standard_normal((1, 21, 27, 32)).astype(F); w = (rng.standard_normal((3, 3, 32, 32)) * 0.06).astype(F)
  g.update({"conv0_x": x, "conv0_w": w, "conv0_args": np.array([2, 2, 1, 0, 1, 0], np.int32),
            "conv0_out": torch_conv_nhwc(x, w, None, 2, 2, 1, 1, 9, 12), "conv_count": np.int32(1)})
  x = rng.standard_normal((1, 15, 22, 32)).astype(F)
  g.update({"convsame_x": x, "convsame_w": w, "convsame_out": torch_conv_nhwc(x, w, None, 2, 1, 1, 0, 8, 11)})
  x = (rng.standard_normal((8, 15)) * 4).astype(F)
  g.update({"softmax0_x": x, "softmax0_out": tfops.softmax(x)})
  return g


def test_consumers_selfcheck_oracle_standin():
  g = _standin_fixture()
  check_nms(g); check_combined_nms(g); check_crop_and_resize(g); check_topk(g); check_conv(g); check_softmax(g)


def test_consumers_selfcheck_kernels_standin(backend):
  """Kernel <-> oracle agreement on the stand-in inputs through the consumers the TF fixture will use."""
  name, lib = backend
  g = _standin_fixture()
  check_nms(g, lib); check_combined_nms(g, lib); check_crop_and_resize(g, lib); check_topk(g, lib); check_conv(g, lib)


def test_generator_script_is_plain_tf1_and_matches_the_consumers():
  """The recipe exists, parses, and writes every key the consumers read."""
  import ast
  src = open(os.path.join(G, "make_tf_golden.py")).read()
  ast.parse(src)
  for key in ("nms%d_idx", "cnms%d_out_valid", "car%d_out", "topk%d_indices", "conv%d_out", "convsame_out", "softmax0_out",
              "model_final_boxes", "model_fpn_box_feat", "nms_count", "cnms_count", "car_count", "topk_count", "conv_count"):
    assert '"%s"' % key in src or "'%s'" % key in src, key
  assert "tf.image.combined_non_max_suppression" in src and "tf.image.crop_and_resize" in src and "tf.nn.top_k" in src
