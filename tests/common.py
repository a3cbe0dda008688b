"""Shared helpers for the parity tests (oracle = checker, never the product)."""
import numpy as np
import torch
import torch.nn.functional as TF

from object_detection_tracking_amd.config import make_config
from object_detection_tracking_amd.weights import synthetic_frames, synthetic_weights

_W = {}


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


def torch_conv_nhwc(x, w_hwio, b, stride, dil, pad_t, pad_l, Ho, Wo):
  """Plain torch fp32 reference of the conv op (NHWC in/out)."""
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
