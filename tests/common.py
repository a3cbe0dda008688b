"""Shared helpers for the parity tests (oracle = checker, never the product)."""
import numpy as np
import torch
import torch.nn.functional as TF

from object_detection_tracking_amd.config import make_config as _make_config
from object_detection_tracking_amd.weights import synthetic_frames, synthetic_weights

_W = {}


def make_config(**kw):
  """The package's make_config with conv_split_family pinned to 0 (the fp16x2 kernels where eligible, UNGUARDED) unless
  the test says otherwise: the product default is "auto" (a bf16x3 twin handle checks the first forward, models._Engine),
  which would build two handles and run two extra forwards in every test.  The guard and the default itself have their own
  tests (test_e2e.py: test_auto_family_*, test_default_engine_is_guarded, test_trained_like_bn_statistics_*)."""
  kw.setdefault("conv_split_family", 0)
  return _make_config(**kw)


def small_config(**kw):
  base = dict(rpn_test_post_nms_topk=64, resnet_num_block=[1, 1, 2, 3], max_size=256,
              short_edge_size=96)
  base.update(kw)
  return make_config(**base)


def weights_for(cfg, seed=0):
  key = (tuple(cfg.resnet_num_block), cfg.num_class, seed, bool(getattr(cfg, "use_frcnn_class_agnostic", False)),
         bool(getattr(cfg, "add_mask", False)), getattr(cfg, "mrcnn_head_dim", 256))
  if key not in _W:
    _W[key] = synthetic_weights(cfg, seed)
  return _W[key]


def torch_conv_nhwc(x, w_hwio, b, stride, dil, pad_t, pad_l, Ho, Wo, dtype=None):
  """Plain torch reference of the conv op (NHWC in/out): fp32, or float64 with dtype=np.float64."""
  if dtype is not None:
    x = np.asarray(x, dtype); w_hwio = np.asarray(w_hwio, dtype); b = None if b is None else np.asarray(b, dtype)
  xt = torch.from_numpy(np.ascontiguousarray(x)).permute(0, 3, 1, 2)
  wt = torch.from_numpy(np.ascontiguousarray(w_hwio)).permute(3, 2, 0, 1)
  kh, kw = w_hwio.shape[:2]
  need_h = (Ho - 1) * stride + (kh - 1) * dil + 1
  need_w = (Wo - 1) * stride + (kw - 1) * dil + 1
  pb = max(0, need_h - (x.shape[1] + pad_t)); pr = max(0, need_w - (x.shape[2] + pad_l))
  xt = TF.pad(xt, (pad_l, pr, pad_t, pb))
  y = TF.conv2d(xt, wt, torch.from_numpy(b) if b is not None else None, stride=stride,
                dilation=dil)[:, :, :Ho, :Wo]
  return y.permute(0, 2, 3, 1).contiguous().numpy()


def canon_dets(boxes, labels, probs):
  """Canonical order for comparing detections: prob desc, label, box."""
  order = np.lexsort((boxes[:, 3], boxes[:, 2], boxes[:, 1], boxes[:, 0], labels, -probs))
  return boxes[order], labels[order], probs[order]


def match_detections(b1, l1, p1, b2, l2, p2, tol_box=1e-3, tol_prob=1e-4):
  """Greedy one-to-one match of two detection sets (same label, boxes within tol);
  returns number unmatched on each side."""
  used = np.zeros(len(b2), bool)
  miss = 0
  for i in range(len(b1)):
    d = np.abs(b2 - b1[i]).max(1)
    ok = (~used) & (l2 == l1[i]) & (d <= tol_box) & (np.abs(p2 - p1[i]) <= tol_prob)
    j = np.where(ok)[0]
    if j.size:
      used[j[0]] = True
    else:
      miss += 1
  return miss, int((~used).sum())


def match_pairs(b1, l1, p1, b2, l2, p2, tol_box=1e-3, tol_prob=1e-4):
  """The one-to-one assignment behind match_detections: [(i, j)] with equal labels, boxes within tol_box and scores
  within tol_prob (each i takes the nearest free j), plus the numbers left unmatched on each side."""
  used = np.zeros(len(b2), bool)
  pairs, miss = [], 0
  for i in range(len(b1)):
    d = np.abs(b2 - b1[i]).max(1) if len(b2) else np.zeros(0)
    ok = np.where((~used) & (l2 == l1[i]) & (d <= tol_box) & (np.abs(p2 - p1[i]) <= tol_prob))[0]
    if ok.size:
      j = int(ok[np.argmin(d[ok])])
      used[j] = True
      pairs.append((i, j))
    else:
      miss += 1
  return pairs, miss, int((~used).sum())


def assert_same_detections(boxes, labels, probs, feats, rboxes, rlabels, rprobs, rfeats, box_tol, prob_tol, feat_tol):
  """Unconditional comparison of two detection lists that agree as SETS: every detection pairs with exactly one of the
  other side (same label, box within box_tol px, score within prob_tol), the pair's appearance features agree to
  feat_tol of the tensor's scale, both lists are in score-descending order, and a pair's positions differ only where
  the scores in between are tied to within prob_tol (top_k(sorted=False) leaves that order to the implementation)."""
  pairs, miss, extra = match_pairs(boxes, labels, probs, rboxes, rlabels, rprobs, box_tol, prob_tol)
  assert miss == 0 and extra == 0 and len(pairs) == len(boxes) == len(rboxes), (miss, extra, len(boxes), len(rboxes))
  assert np.all(np.diff(probs) <= 0), "scores not in descending order"
  scale = max(1e-6, float(np.abs(rfeats).max())) if rfeats is not None and len(rboxes) else 1.0
  for i, j in pairs:
    assert np.abs(boxes[i] - rboxes[j]).max() <= box_tol and abs(float(probs[i]) - float(rprobs[j])) <= prob_tol
    if i != j:
      lo, hi = min(i, j), max(i, j)
      assert float(rprobs[lo]) - float(rprobs[hi]) <= 2 * prob_tol, "order differs beyond a score tie: %d vs %d" % (i, j)
    if feats is not None and rfeats is not None:
      assert float(np.abs(feats[i] - rfeats[j]).max()) / scale < feat_tol, ("feature", i, j)
  return len(pairs)


def tie_swaps(b1, l1, p1, b2, l2, p2, tol_box, tol_prob, iou_thr=0.5):
  """Of the detections match_detections leaves unmatched on both sides, the number of PAIRS that are one NMS decision
  between near-tied competitors: same label, scores within tol_prob, boxes overlapping by more than the NMS threshold
  (so exactly one of the two survives, and which one is decided by score differences below the f32 noise of any
  implementation -- random-init heads produce such plateaus).  Greedy one-to-one."""
  def unmatched(ba, la, pa, bb, lb, pb):
    used = np.zeros(len(bb), bool); left = []
    for i in range(len(ba)):
      d = np.abs(bb - ba[i]).max(1)
      j = np.where((~used) & (lb == la[i]) & (d <= tol_box) & (np.abs(pb - pa[i]) <= tol_prob))[0]
      if j.size:
        used[j[0]] = True
      else:
        left.append(i)
    return left, list(np.where(~used)[0])
  m1, m2 = unmatched(b1, l1, p1, b2, l2, p2)
  def iou(a, b):
    iw = max(0.0, min(a[2], b[2]) - max(a[0], b[0])); ih = max(0.0, min(a[3], b[3]) - max(a[1], b[1]))
    inter = iw * ih
    return inter / ((a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1]) - inter + 1e-12)
  taken = set(); ties = 0
  for i in m1:
    for j in m2:
      if j in taken or l1[i] != l2[j] or abs(float(p1[i]) - float(p2[j])) > tol_prob:
        continue
      if iou(b1[i], b2[j]) > iou_thr:
        taken.add(j); ties += 1
        break
  return ties
