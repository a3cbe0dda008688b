"""ORACLE (test infrastructure only -- see oracle/__init__.py).

numpy float32 restatement of the TensorFlow-1.15 CPU op semantics the
reference's hot path depends on.  TensorFlow is an un-vendored third-party
dependency (reference README.md:63, "tensorflow-gpu==1.15"); **parity of these
ops is unpinned** by any reference test.  Each function names the reference
call site and the TF-1.15 kernel whose published algorithm it follows.

Canonical choices where TF leaves the result unspecified (the HIP path makes
the same choices, tests pin them):
  * ``tf.nn.top_k(sorted=False)``: we return the k largest in (score desc,
    index asc) order; ties at the k-th value keep the lower indices.
  * NMS candidates with equal scores: lower index first (TF<=1.15 uses a
    max-heap without an index tie-break).
  * ``combined_non_max_suppression`` cross-class merge with equal scores:
    lower class first, then per-class selection order.
All arithmetic is float32 with one rounding per operation (TF-1.15 wheels are
built without FMA), evaluated in the operand order of the TF kernels.
"""
from __future__ import annotations

import numpy as np

F = np.float32


def top_k(scores, k):
  """tf.nn.top_k(scores, k, sorted=False) -> indices (reference nn.py:1368,
  models.py:432,1295).  Canonical order: score desc, index asc."""
  scores = np.asarray(scores, F)
  order = np.argsort(-scores, kind="stable")
  return order[:k].astype(np.int64)


def _minmax(b):
  """TF normalises each box with min/max per axis (non_max_suppression_op.cc
  IOU())."""
  a0 = np.minimum(b[..., 0], b[..., 2]); a2 = np.maximum(b[..., 0], b[..., 2])
  a1 = np.minimum(b[..., 1], b[..., 3]); a3 = np.maximum(b[..., 1], b[..., 3])
  return a0, a1, a2, a3


def iou_one_to_many(box, others):
  """TF-1.15 non_max_suppression_op.cc IOU(): float32; IoU = 0 if either area
  <= 0; inter / (area_i + area_j - inter), no +1."""
  box = np.asarray(box, F); others = np.asarray(others, F).reshape(-1, 4)
  i0, i1, i2, i3 = _minmax(box)
  j0, j1, j2, j3 = _minmax(others)
  area_i = F(i2 - i0) * F(i3 - i1)
  area_j = (j2 - j0) * (j3 - j1)
  y0 = np.maximum(i0, j0); x0 = np.maximum(i1, j1)
  y1 = np.minimum(i2, j2); x1 = np.minimum(i3, j3)
  inter = np.maximum(y1 - y0, F(0)) * np.maximum(x1 - x0, F(0))
  with np.errstate(divide="ignore", invalid="ignore"):
    iou = inter / ((area_i + area_j) - inter)
  bad = (area_j <= 0) | (area_i <= 0)
  return np.where(bad, F(0), iou).astype(F)


def non_max_suppression(boxes, scores, max_output_size, iou_threshold,
                        score_threshold=-np.inf):
  """tf.image.non_max_suppression (V3) -> selected indices in selection order
  (reference nn.py:1390, models.py:1211).  Greedy: candidates by descending
  score (ties: lower index first); a candidate is dropped iff IoU with an
  already selected box is > iou_threshold (strict); stop at max_output_size."""
  boxes = np.asarray(boxes, F).reshape(-1, 4)
  scores = np.asarray(scores, F).reshape(-1)
  thr = F(iou_threshold)
  order = np.argsort(-scores, kind="stable")
  sel = []
  sel_boxes = np.zeros((0, 4), F)
  for idx in order:
    if len(sel) >= max_output_size:
      break
    if not scores[idx] > score_threshold:
      continue
    if len(sel):
      if np.any(iou_one_to_many(boxes[idx], sel_boxes) > thr):
        continue
    sel.append(int(idx))
    sel_boxes = np.concatenate([sel_boxes, boxes[idx][None]], 0)
  return np.asarray(sel, np.int64)


def combined_non_max_suppression(boxes, scores, max_output_size_per_class,
                                 max_total_size, iou_threshold,
                                 score_threshold=-np.inf):
  """tf.image.combined_non_max_suppression(clip_boxes=False,
  pad_per_class=False) (reference nn.py:1468-1474, models.py:2959-2965).

  boxes [B,N,Q,4] (Q = 1 or C), scores [B,N,C].  Returns zero-padded
  (nmsed_boxes [B,T,4], nmsed_scores [B,T], nmsed_classes [B,T] float32,
  valid_detections [B] int32), T = max_total_size.
  """
  boxes = np.asarray(boxes, F); scores = np.asarray(scores, F)
  B, N, Q, _ = boxes.shape
  C = scores.shape[2]
  T = int(max_total_size)
  out_b = np.zeros((B, T, 4), F); out_s = np.zeros((B, T), F)
  out_c = np.zeros((B, T), F); valid = np.zeros((B,), np.int32)
  for b in range(B):
    cand = []  # (score, class, order, box_index)
    for c in range(C):
      q = c if Q > 1 else 0
      sel = non_max_suppression(boxes[b, :, q], scores[b, :, c],
                                max_output_size_per_class, iou_threshold,
                                score_threshold)
      for o, i in enumerate(sel):
        cand.append((scores[b, i, c], c, o, i, q))
    cand.sort(key=lambda t: (-t[0], t[1], t[2]))
    cand = cand[:T]
    valid[b] = len(cand)
    for t, (s, c, _, i, q) in enumerate(cand):
      out_b[b, t] = boxes[b, i, q]; out_s[b, t] = s; out_c[b, t] = c
  return out_b, out_s, out_c, valid


def crop_and_resize(image, boxes, box_ind, crop_size):
  """tf.image.crop_and_resize(bilinear, extrapolation_value=0) following
  TF-1.15 crop_and_resize_op.cc (CPU functor).

  image [N,H,W,C] float32; boxes [K,4] normalised y1,x1,y2,x2; box_ind [K];
  returns [K,crop,crop,C].  A sample with in_y < 0 or in_y > H-1 (resp. x) is
  0; otherwise top=floor, bottom=ceil, lerp in the TF operand order.
  """
  image = np.asarray(image, F); boxes = np.asarray(boxes, F).reshape(-1, 4)
  _, H, W, C = image.shape
  ch = cw = int(crop_size)
  K = boxes.shape[0]
  out = np.zeros((K, ch, cw, C), F)
  ii = np.arange(ch).astype(F)
  for k in range(K):
    y1, x1, y2, x2 = boxes[k]
    img = image[int(box_ind[k])]
    if ch > 1:
      hs = (y2 - y1) * F(H - 1) / F(ch - 1)
      in_y = y1 * F(H - 1) + ii * hs
    else:                                   # crop_and_resize_op.cc: single sample at the box centre
      in_y = np.array([F(0.5) * (y1 + y2) * F(H - 1)], F)
    if cw > 1:
      ws = (x2 - x1) * F(W - 1) / F(cw - 1)
      in_x = x1 * F(W - 1) + ii * ws
    else:
      in_x = np.array([F(0.5) * (x1 + x2) * F(W - 1)], F)
    vy = ~((in_y < 0) | (in_y > F(H - 1)))
    vx = ~((in_x < 0) | (in_x > F(W - 1)))
    if not vy.any() or not vx.any():
      continue
    iy = np.where(vy, in_y, F(0)); ix = np.where(vx, in_x, F(0))
    top = np.floor(iy).astype(np.int64); bot = np.ceil(iy).astype(np.int64)
    lef = np.floor(ix).astype(np.int64); rig = np.ceil(ix).astype(np.int64)
    yl = (iy - top.astype(F))[:, None, None]
    xl = (ix - lef.astype(F))[None, :, None]
    tl = img[top][:, lef]; tr = img[top][:, rig]
    bl = img[bot][:, lef]; br = img[bot][:, rig]
    t = tl + (tr - tl) * xl
    bm = bl + (br - bl) * xl
    val = t + (bm - t) * yl
    val = np.where((vy[:, None] & vx[None, :])[:, :, None], val, F(0))
    out[k] = val
  return out


def softmax(logits):
  """tf.nn.softmax (softmax_op_functor.h): exp(x - max) * (1 / sum)."""
  x = np.asarray(logits, F)
  sh = x - x.max(axis=-1, keepdims=True)
  e = np.exp(sh).astype(F)
  inv = (F(1) / e.sum(axis=-1, keepdims=True, dtype=F)).astype(F)
  return (e * inv).astype(F)
