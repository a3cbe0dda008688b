#!/usr/bin/env python
"""Generate tests/golden/tf_ref.npz by RUNNING TensorFlow -- the one dependency of the reference's hot path that is
not in the build container (reference README.md:63: tensorflow-gpu 1.15; not installable here: no network).

Until this file exists the detector's parity is "with a CPU restatement of the TF ops" (oracle/tfops.py, oracle/graph.py;
DESIGN.md section 4: "parity unpinned").  Run this ONCE on any host with TensorFlow 1.15 (or 2.x: the tf.compat.v1
surface is used) and numpy, commit the resulting tests/golden/tf_ref.npz, and tests/test_tf_golden.py pins the oracle,
the simulator build and the HIP kernels against TensorFlow itself:

    python tests/golden/make_tf_golden.py                       # op-level fixtures only
    python tests/golden/make_tf_golden.py --reference /path/to/Object_Detection_Tracking
                                                                 # + one small Mask_RCNN_FPN forward through the
                                                                 #   reference's own models.py / nn.py

Everything is seeded; inputs are stored next to the outputs, so the consumer needs neither TF nor this script.

Fixtures (npz keys, i = case index):
  nms{i}_boxes [n,4] y1x1y2x2, nms{i}_scores [n], nms{i}_args [max_out, iou] -> nms{i}_idx
        tf.image.non_max_suppression                      (reference nn.py:1276, 1390; models.py:1211)
  cnms{i}_boxes [B,N,C,4], cnms{i}_scores [B,N,C], cnms{i}_args [per_class, total, iou, score_thr]
        -> cnms{i}_out_boxes / _out_scores / _out_classes / _out_valid
        tf.image.combined_non_max_suppression(clip_boxes=False)   (reference nn.py:1468, models.py:2959)
  car{i}_image [B,H,W,C], car{i}_boxes [R,4] normalised y1x1y2x2, car{i}_ind [R], car{i}_crop [2] -> car{i}_out
        tf.image.crop_and_resize(bilinear, extrapolation 0)       (reference nn.py:1258-1271)
  topk{i}_x [n], topk{i}_k -> topk{i}_values, topk{i}_indices     tf.nn.top_k(sorted=True)  (nn.py:1368, models.py:1295)
  conv{i}_x [B,H,W,Cin] NHWC, conv{i}_w HWIO, conv{i}_args [stride, dil, pad_t, pad_b, pad_l, pad_r] -> conv{i}_out NHWC
        tf.pad + tf.nn.conv2d(VALID) as reference nn.py:337-381 builds its strided convs (odd pads), and 'SAME'
  softmax{i}_x -> softmax{i}_out                                   tf.nn.softmax (models.py:843)
  model_* : frame, config scalars, weight seed -> final_boxes / final_labels / final_probs / fpn_box_feat of the
        reference's Mask_RCNN_FPN graph on seeded synthetic weights (object_detection_tracking_amd.weights.
        synthetic_weights: the same generator the tests use), --reference only.
"""
import argparse
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def get_tf():
  import tensorflow as tf
  if hasattr(tf, "compat") and hasattr(tf.compat, "v1") and int(tf.__version__.split(".")[0]) >= 2:
    tf = tf.compat.v1
    tf.disable_v2_behavior()
  return tf


def op_fixtures(tf, out):
  rng = np.random.default_rng(20260926)
  F = np.float32
  sess = tf.Session(config=tf.ConfigProto(device_count={"GPU": 0}))

  def boxes_clustered(n, spread=60.0, size=(20.0, 120.0)):
    c = rng.uniform(0, 600, (max(1, n // 6), 2))
    ctr = c[rng.integers(0, len(c), n)] + rng.normal(0, spread * 0.15, (n, 2))
    wh = rng.uniform(size[0], size[1], (n, 2))
    b = np.concatenate([ctr - wh / 2, ctr + wh / 2], 1).astype(F)       # y1 x1 y2 x2
    return b

  # ---- tf.image.non_max_suppression: clustered boxes, exact score ties, zero-area boxes, flipped corners
  cases = []
  for n, k, thr in ((300, 100, 0.7), (1000, 300, 0.7), (64, 64, 0.5), (500, 20, 0.3)):
    b = boxes_clustered(n)
    s = rng.standard_normal(n).astype(F)
    cases.append((b, s, k, thr))
  b = boxes_clustered(200); s = np.round(rng.uniform(0, 1, 200), 1).astype(F)        # many ties
  cases.append((b, s, 100, 0.5))
  b = boxes_clustered(100); b[::7, 2:] = b[::7, :2]; s = rng.uniform(0, 1, 100).astype(F)    # zero-area boxes
  cases.append((b, s, 100, 0.5))
  b = boxes_clustered(100); b[::5] = b[::5][:, [2, 3, 0, 1]]; s = rng.uniform(0, 1, 100).astype(F)   # flipped corners
  cases.append((b, s, 50, 0.5))
  for i, (b, s, k, thr) in enumerate(cases):
    idx = sess.run(tf.image.non_max_suppression(tf.constant(b), tf.constant(s), k, iou_threshold=thr))
    out["nms%d_boxes" % i] = b; out["nms%d_scores" % i] = s
    out["nms%d_args" % i] = np.array([k, thr], np.float64); out["nms%d_idx" % i] = idx.astype(np.int32)
  out["nms_count"] = np.int32(len(cases))

  # ---- tf.image.combined_non_max_suppression as fastrcnn_predictions_multibatch calls it (models.py:2959-2965)
  cases = []
  for B, N, C, per_class, total, thr, sthr in ((2, 64, 14, 100, 100, 0.5, -np.inf), (3, 300, 14, 100, 100, 0.5, -np.inf),
                                               (1, 40, 3, 10, 15, 0.5, 0.05), (2, 50, 4, 100, 100, 0.5, -np.inf)):
    bx = np.stack([np.stack([boxes_clustered(N) for _ in range(C)], 1) for _ in range(B)], 0).astype(F)   # [B,N,C,4]
    sc = rng.uniform(0, 1, (B, N, C)).astype(F) ** 3
    if B == 2 and N == 50:        # zero-score padding rows (the batched graph's fewer-than-K images)
      sc[1, 30:] = 0.0; bx[1, 30:] = 0.0
    cases.append((bx, sc, per_class, total, thr, sthr))
  for i, (bx, sc, pc, tot, thr, sthr) in enumerate(cases):
    r = sess.run(tf.image.combined_non_max_suppression(tf.constant(bx), tf.constant(sc), max_output_size_per_class=pc,
                                                       max_total_size=tot, iou_threshold=thr,
                                                       score_threshold=float(sthr), pad_per_class=False, clip_boxes=False))
    out["cnms%d_boxes" % i] = bx; out["cnms%d_scores" % i] = sc
    out["cnms%d_args" % i] = np.array([pc, tot, thr, sthr], np.float64)
    out["cnms%d_out_boxes" % i] = r[0]; out["cnms%d_out_scores" % i] = r[1]
    out["cnms%d_out_classes" % i] = r[2]; out["cnms%d_out_valid" % i] = r[3].astype(np.int32)
  out["cnms_count"] = np.int32(len(cases))

  # ---- tf.image.crop_and_resize: boxes inside, crossing and outside the image, samples exactly on dim - 1
  cases = []
  for B, H, W, C, R, crop in ((2, 17, 23, 8, 40, (14, 14)), (1, 68, 120, 16, 100, (14, 14)), (1, 9, 9, 4, 12, (28, 28))):
    img = rng.standard_normal((B, H, W, C)).astype(F)
    y1 = rng.uniform(-0.2, 0.9, R); x1 = rng.uniform(-0.2, 0.9, R)
    bb = np.stack([y1, x1, y1 + rng.uniform(0.02, 0.6, R), x1 + rng.uniform(0.02, 0.6, R)], 1).astype(F)
    bb[0] = [0, 0, 1, 1]; bb[1] = [0.5, 0.5, 1.0, 1.0]; bb[2] = [1.0, 1.0, 1.2, 1.2]; bb[3] = [0.25, 0.25, 0.25, 0.25]
    ind = rng.integers(0, B, R).astype(np.int32)
    cases.append((img, bb, ind, crop))
  for i, (img, bb, ind, crop) in enumerate(cases):
    r = sess.run(tf.image.crop_and_resize(tf.constant(img), tf.constant(bb), tf.constant(ind), list(crop)))
    out["car%d_image" % i] = img; out["car%d_boxes" % i] = bb; out["car%d_ind" % i] = ind
    out["car%d_crop" % i] = np.array(crop, np.int32); out["car%d_out" % i] = r
  out["car_count"] = np.int32(len(cases))

  # ---- tf.nn.top_k (sorted=True gives the order the test can compare; the reference passes sorted=False)
  cases = [(rng.standard_normal(5000).astype(F), 300), (np.round(rng.uniform(0, 1, 2000), 2).astype(F), 1000),
           (rng.standard_normal(50).astype(F), 50)]
  for i, (x, k) in enumerate(cases):
    v, ix = sess.run(tf.nn.top_k(tf.constant(x), k=k, sorted=True))
    out["topk%d_x" % i] = x; out["topk%d_k" % i] = np.int32(k); out["topk%d_values" % i] = v
    out["topk%d_indices" % i] = ix.astype(np.int32)
  out["topk_count"] = np.int32(len(cases))

  # ---- conv2d: tf.pad + VALID with the reference's odd pads (nn.py:337-381, 871-896), dilation, and plain 'SAME'.
  # NHWC on the CPU (stock TF CPU kernels reject NCHW); the reference's NCHW graph computes the same numbers.
  cases = []
  for B, H, W, Cin, Cout, k, stride, dil, pads in ((1, 33, 41, 32, 64, 3, 2, 1, (1, 0, 1, 0)),     # maybe_reverse_pad(0, 1)
                                                   (2, 20, 28, 32, 32, 3, 1, 2, (2, 2, 2, 2)),     # dilated, symmetric
                                                   (1, 37, 45, 32, 64, 7, 2, 1, (3, 2, 3, 2)),     # conv0-style
                                                   (1, 16, 16, 64, 128, 1, 1, 1, (0, 0, 0, 0)),
                                                   (1, 21, 27, 32, 32, 3, 2, 2, (1, 0, 1, 0))):    # res5 block0: stride 2 + dilation 2
    x = rng.standard_normal((B, H, W, Cin)).astype(F)
    w = (rng.standard_normal((k, k, Cin, Cout)) * np.sqrt(2.0 / (k * k * Cin))).astype(F)
    cases.append((x, w, stride, dil, pads))
  for i, (x, w, stride, dil, pads) in enumerate(cases):
    xp = tf.pad(tf.constant(x), [[0, 0], [pads[0], pads[1]], [pads[2], pads[3]], [0, 0]])
    y = tf.nn.conv2d(xp, tf.constant(w), strides=[1, stride, stride, 1], padding="VALID", dilations=[1, dil, dil, 1])
    out["conv%d_x" % i] = x; out["conv%d_w" % i] = w
    out["conv%d_args" % i] = np.array([stride, dil] + list(pads), np.int32); out["conv%d_out" % i] = sess.run(y)
  x = rng.standard_normal((1, 15, 22, 32)).astype(F); w = (rng.standard_normal((3, 3, 32, 32)) * 0.06).astype(F)
  out["convsame_x"] = x; out["convsame_w"] = w
  out["convsame_out"] = sess.run(tf.nn.conv2d(tf.constant(x), tf.constant(w), strides=[1, 2, 2, 1], padding="SAME"))
  out["conv_count"] = np.int32(len(cases))

  x = (rng.standard_normal((64, 15)) * 4).astype(F)
  out["softmax0_x"] = x; out["softmax0_out"] = sess.run(tf.nn.softmax(tf.constant(x)))
  sess.close()


def model_fixture(tf, reference, out, size=(160, 224), topk=50):
  """One Mask_RCNN_FPN forward of the reference's own graph code on seeded synthetic weights."""
  sys.path.insert(0, ROOT)
  sys.path.insert(0, reference)
  from object_detection_tracking_amd.config import make_config
  from object_detection_tracking_amd.weights import synthetic_frames, synthetic_weights
  H, W = size
  cfg = make_config(rpn_test_post_nms_topk=topk, short_edge_size=H, max_size=W, resnet_num_block=[1, 1, 2, 3])
  # what get_args() sets beyond the namespace make_config mirrors (obj_detect_tracking.py:296-330)
  for k, v in dict(is_pack_model=False, diva_class3=True, diva_class=False, diva_class2=False, use_so_score_thres=False,
                   use_so_association=False, so_person_topk=10, freeze_rpn=True, freeze_fastrcnn=True, freeze=2,
                   small_objects=[], frcnn_batch_size=512, fastrcnn_batch_per_im=512, fastrcnn_fg_thres=0.5,
                   anchor_stride=16, use_mixup=False, use_focal_loss=False, use_frcnn_focal_loss=False,
                   wd=None, weight_decay=None, controller="/cpu:0", load_from=None, use_all_mem=False,
                   resnet152=False, resnet50=False, resnet34=False, resnet18=False, use_cascade_rcnn=False,
                   actasobj=False, bupt_exp=False, tf_pad_reverse=True, multi_scale_testing=False,
                   test_frame_extraction=False).items():
    if not hasattr(cfg, k):
      setattr(cfg, k, v)
  import models as ref_models                                    # the reference's models.py
  model = ref_models.get_model(cfg, 0, controller="/cpu:0")
  weights = synthetic_weights(cfg, 0)
  frame = synthetic_frames(1, H, W, seed=77)[0].astype(np.float32)
  with tf.Session(config=tf.ConfigProto(allow_soft_placement=True, device_count={"GPU": 0})) as sess:
    sess.run(tf.global_variables_initializer())
    missing = []
    for v in tf.global_variables():
      key = v.name.split(":")[0]
      if key in weights:
        v.load(weights[key].astype(np.float32), sess)
      elif "global_step" not in key:
        missing.append(key)
    if missing:
      raise RuntimeError("graph variables without a synthetic weight: %s" % missing[:8])
    fetch = [model.final_boxes, model.final_labels, model.final_probs, model.fpn_box_feat]
    boxes, labels, probs, feats = sess.run(fetch, feed_dict=model.get_feed_dict_forward(frame))
  out["model_frame"] = frame.astype(np.uint8)
  out["model_config"] = np.array([H, W, topk, 0], np.int32)      # H, W, rpn_test_post_nms_topk, weight seed
  out["model_blocks"] = np.array([1, 1, 2, 3], np.int32)
  out["model_final_boxes"] = boxes; out["model_final_labels"] = labels.astype(np.int64)
  out["model_final_probs"] = probs; out["model_fpn_box_feat"] = feats


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--reference", default=None, help="checkout of JunweiLiang/Object_Detection_Tracking (adds the model fixture)")
  ap.add_argument("--out", default=os.path.join(HERE, "tf_ref.npz"))
  a = ap.parse_args()
  tf = get_tf()
  import tensorflow
  out = {"tf_version": np.array(tensorflow.__version__)}
  op_fixtures(tf, out)
  if a.reference:
    tf.reset_default_graph()
    model_fixture(tf, a.reference, out)
  np.savez_compressed(a.out, **out)
  print("wrote %s (%d arrays, TensorFlow %s)" % (a.out, len(out), tensorflow.__version__))


if __name__ == "__main__":
  main()
