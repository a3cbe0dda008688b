"""ORACLE (test infrastructure only -- see oracle/__init__.py).

CPU restatement of the device graph that one ``sess.run`` of the reference
executes: dense parts with torch-CPU fp32
(conv2d / matmul, in the reference's NCHW layout), every selection / gather /
box op with the numpy TF-op restatements in oracle/tfops.py.

  b=1  : Mask_RCNN_FPN.build_forward        reference models.py:488-973
  b=B  : Mask_RCNN_FPN_multi.build_forward  reference models.py:2058-2408
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as TF

from . import tfops
from .anchors import all_anchors_fpn

F = np.float32


# ----------------------------------------------------------------- dense ops
def _w(weights, name):
  return torch.from_numpy(np.ascontiguousarray(weights[name], dtype=F))


def conv2d(x, weights, scope, stride=1, padding="SAME", dilation=1):
  """reference nn.py:337-381 (tf.nn.conv2d NCHW, HWIO weights, optional bias).
  SAME is only used with stride 1 on this path => symmetric (k-1)*d/2."""
  W = _w(weights, scope + "/W").permute(3, 2, 0, 1).contiguous()
  b = _w(weights, scope + "/b") if (scope + "/b") in weights else None
  k = W.shape[2]
  if padding == "SAME":
    assert stride == 1
    p = (k - 1) * dilation // 2
    x = TF.pad(x, (p, p, p, p))
  return TF.conv2d(x, W, b, stride=stride, dilation=dilation)


def batch_norm(x, weights, scope, eps=1e-5):
  """reference nn.py:1771-1774 -> tf.nn.batch_normalization:
  inv = rsqrt(var+eps)*gamma ; y = x*inv + (beta - mean*inv)."""
  g = _w(weights, scope + "/gamma"); b = _w(weights, scope + "/beta")
  m = _w(weights, scope + "/mean/EMA"); v = _w(weights, scope + "/variance/EMA")
  inv = torch.rsqrt(v + eps) * g
  return x * inv.view(1, -1, 1, 1) + (b - m * inv).view(1, -1, 1, 1)


def pad_tl(x, n=1):
  """maybe_reverse_pad(0, 1, reverse=True) -> [1, 0]: pad top/left only
  (reference nn.py:777-781)."""
  return TF.pad(x, (n, 0, n, 0))


def bottleneck(x, weights, pre, ch_out, stride, dilation):
  """reference nn.py:459-521 + shortcut nn.py:551-566."""
  sc = x
  l = torch.relu(batch_norm(conv2d(x, weights, pre + "/conv1"), weights,
                            pre + "/conv1/bn"))
  if stride == 2:
    l = pad_tl(l)
    l = conv2d(l, weights, pre + "/conv2", stride=2, padding="VALID",
               dilation=dilation)
    l = torch.relu(batch_norm(l, weights, pre + "/conv2/bn"))
    if dilation != 1:
      l = pad_tl(l)            # nn.py:493-497 ("weird" zero row/col AFTER relu)
  else:
    l = conv2d(l, weights, pre + "/conv2", dilation=dilation)
    l = torch.relu(batch_norm(l, weights, pre + "/conv2/bn"))
  l = batch_norm(conv2d(l, weights, pre + "/conv3"), weights, pre + "/conv3/bn")
  if sc.shape[1] != ch_out * 4:
    if stride == 2:
      sc = sc[:, :, :-1, :-1]
      sc = conv2d(sc, weights, pre + "/convshortcut", stride=2, padding="VALID")
    else:
      sc = conv2d(sc, weights, pre + "/convshortcut")
    sc = batch_norm(sc, weights, pre + "/convshortcut/bn")
  return torch.relu(l + sc)    # nn.py:519 + relu at nn.py:587


def preprocess(images):
  """reference models.py:340-355: x*(1/255), minus BGR mean, / BGR std,
  NHWC->NCHW.  images: [B,H,W,3] uint8 or float32 BGR 0..255."""
  x = torch.from_numpy(np.ascontiguousarray(images)).to(torch.float32)
  mean = torch.tensor([0.485, 0.456, 0.406][::-1], dtype=torch.float32)
  std = torch.tensor([0.229, 0.224, 0.225][::-1], dtype=torch.float32)
  x = x * np.float32(1.0 / 255)
  x = (x - mean) / std
  return x.permute(0, 3, 1, 2).contiguous()


def backbone(img, weights, config, taps=None):
  """reference nn.py:843-944 (tf_pad_reverse=True)."""
  H, W = img.shape[2:]
  mult = config.fpn_resolution_requirement
  ph = int(np.ceil(H / mult) * mult) - H
  pw = int(np.ceil(W / mult) * mult) - W
  # pad_base = maybe_reverse_pad(2,3,True) = [3,2]
  l = TF.pad(img, (3, 2 + pw, 3, 2 + ph))
  l = conv2d(l, weights, "conv0", stride=2, padding="VALID")
  l = torch.relu(batch_norm(l, weights, "conv0/bn"))
  if taps is not None: taps["conv0"] = l.numpy()
  l = pad_tl(l)
  l = TF.max_pool2d(l, 3, 2)
  if taps is not None: taps["pool0"] = l.numpy()
  feats = []
  for g, (ch, cnt) in enumerate(zip((64, 128, 256, 512),
                                    config.resnet_num_block)):
    for i in range(cnt):
      stride = (1 if g == 0 else 2) if i == 0 else 1
      dil = 1
      if g == 3 and config.use_dilations and i >= cnt - 3:
        dil = 2                                     # nn.py:577-579,932-936
      l = bottleneck(l, weights, "group%d/block%d" % (g, i), ch, stride, dil)
      if taps is not None and i == 0:
        taps["group%d/block0" % g] = l.numpy()
    feats.append(l)
  return feats


def fpn(c2345, weights):
  """reference nn.py:947-1014."""
  lat = [conv2d(c, weights, "fpn/lateral_1x1_c%d" % (i + 2))
         for i, c in enumerate(c2345)]
  sums = []
  for idx, l in enumerate(lat[::-1]):
    if idx > 0:
      up = sums[-1].repeat_interleave(2, 2).repeat_interleave(2, 3)
      l = l + up
    sums.append(l)
  p = [conv2d(c, weights, "fpn/posthoc_3x3_p%d" % (i + 2))
       for i, c in enumerate(sums[::-1])]
  p6 = p[-1][:, :, ::2, ::2]           # 1x1 max-pool stride 2 VALID
  return p + [p6]


def rpn_head(feat, weights):
  """reference models.py:979-1009 -> logits [B,H,W,A], deltas [B,H,W,A,4]."""
  h = torch.relu(conv2d(feat, weights, "rpn/conv0"))
  lab = conv2d(h, weights, "rpn/class").permute(0, 2, 3, 1)
  box = conv2d(h, weights, "rpn/box").permute(0, 2, 3, 1)
  B, H, W, A = lab.shape
  return lab.contiguous().numpy(), box.reshape(B, H, W, A, 4).numpy()


# ------------------------------------------------------------------- box ops
def decode_bbox_target(deltas, anchors, clip):
  """reference nn.py:1518-1538 (float32, same operand order)."""
  d = np.asarray(deltas, F).reshape(-1, 4); a = np.asarray(anchors, F).reshape(-1, 4)
  waha = a[:, 2:] - a[:, :2]
  xaya = (a[:, 2:] + a[:, :2]) * F(0.5)
  wbhb = np.exp(np.minimum(d[:, 2:], F(clip))).astype(F) * waha
  xbyb = d[:, :2] * waha + xaya
  x1y1 = xbyb - wbhb * F(0.5)
  x2y2 = xbyb + wbhb * F(0.5)
  return np.concatenate([x1y1, x2y2], 1).astype(F)


def clip_boxes(boxes, hw):
  """reference nn.py:1339-1346."""
  h, w = hw
  b = np.maximum(np.asarray(boxes, F), F(0))
  return np.minimum(b, np.array([w, h, w, h], F))


def rpn_proposals_level_b1(boxes, scores, hw, K, nms_thr):
  """reference nn.py:1353-1400 (rpn_min_size = 0, strict >)."""
  k = min(K, scores.size)
  idx = tfops.top_k(scores, k)
  tb = clip_boxes(boxes[idx], hw); ts = scores[idx]
  wh = tb[:, 2:] - tb[:, :2]
  valid = np.all(wh > F(0), axis=1)
  vb, vs = tb[valid], ts[valid]
  keep = tfops.non_max_suppression(vb, vs, K, nms_thr)
  return vb[keep], vs[keep], dict(topk_idx=idx, valid=valid, keep=keep,
                                  nms_in_boxes=vb, nms_in_scores=vs)


def level_of_boxes(boxes):
  """reference models.py:439-461 (float32)."""
  b = np.asarray(boxes, F)
  area = (b[:, 3] - b[:, 1]) * (b[:, 2] - b[:, 0])
  sq = np.sqrt(area).astype(F)
  with np.errstate(divide="ignore", invalid="ignore"):
    lvl = np.floor(F(4) + np.log(sq * F(1. / 224) + F(1e-6)).astype(F) *
                   F(1.0 / np.log(2)))
  lvl = np.nan_to_num(lvl, nan=-100.0, neginf=-100.0, posinf=100.0)
  return np.clip(lvl.astype(np.int32), 2, 5)


def _transform_fpcoor_for_tf(boxes, shape_hw, crop):
  """reference nn.py:1238-1271."""
  b = np.asarray(boxes, F)
  x0, y0, x1, y1 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
  H, W = shape_hw
  sw = (x1 - x0) / F(crop); sh = (y1 - y0) / F(crop)
  nx0 = (x0 + sw / F(2) - F(0.5)) / F(W - 1)
  ny0 = (y0 + sh / F(2) - F(0.5)) / F(H - 1)
  nw = sw * F(crop - 1) / F(W - 1)
  nh = sh * F(crop - 1) / F(H - 1)
  return np.stack([ny0, nx0, ny0 + nh, nx0 + nw], 1).astype(F)


def roi_align(featuremap, boxes, box_ind, out):
  """reference nn.py:1315-1335: crop_and_resize(2*out) then 2x2 avg pool.
  featuremap [B,C,h,w] (numpy NCHW) -> [K,C,out,out]."""
  img = np.ascontiguousarray(np.transpose(featuremap, (0, 2, 3, 1)))
  nb = _transform_fpcoor_for_tf(boxes, img.shape[1:3], out * 2)
  c = tfops.crop_and_resize(img, nb, box_ind, out * 2)   # [K,2o,2o,C]
  K = c.shape[0]
  c = c.reshape(K, out, 2, out, 2, -1)
  s = ((c[:, :, 0, :, 0] + c[:, :, 0, :, 1]) + c[:, :, 1, :, 0]) + c[:, :, 1, :, 1]
  s = (s * F(0.25)).astype(F)
  return np.ascontiguousarray(np.transpose(s, (0, 3, 1, 2)))


def multilevel_roi_align(features, boxes, box_ind, strides, out=7):
  """reference models.py:465-485 (b=1) / models.py:2571-2587 (multi).
  features: list of 4 numpy [B,C,h,w] (already sliced); boxes [K,4]."""
  boxes = np.asarray(boxes, F).reshape(-1, 4)
  K = boxes.shape[0]
  C = features[0].shape[1]
  res = np.zeros((K, C, out, out), F)
  lvl = level_of_boxes(boxes)
  for i in range(4):
    ids = np.where(lvl == i + 2)[0]
    if ids.size == 0:
      continue
    bf = boxes[ids] * F(1.0 / strides[i])
    res[ids] = roi_align(features[i], bf, np.asarray(box_ind)[ids], out)
  return res


def box_head(roi_feat, weights, num_class, partial_ids=None, class_agnostic=False):
  """reference models.py:1030-1108 (fc6/fc7 ReLU, class, box[:,1:])."""
  x = torch.from_numpy(roi_feat.reshape(roi_feat.shape[0], -1))
  h = torch.relu(x @ _w(weights, "fastrcnn/fc6/W") + _w(weights, "fastrcnn/fc6/b"))
  h = torch.relu(h @ _w(weights, "fastrcnn/fc7/W") + _w(weights, "fastrcnn/fc7/b"))
  cls = h @ _w(weights, "fastrcnn/outputs/class/W") + \
      _w(weights, "fastrcnn/outputs/class/b")
  box = h @ _w(weights, "fastrcnn/outputs/box/W") + \
      _w(weights, "fastrcnn/outputs/box/b")
  if class_agnostic:
    # reference models.py:1164-1168 (one box per RoI) + :798-802 (tile over the foreground classes)
    box = box.reshape(-1, 1, 4).repeat(1, num_class - 1, 1)
  else:
    box = box.reshape(-1, num_class, 4)[:, 1:, :]
  cls, box = cls.numpy(), np.ascontiguousarray(box.numpy())
  if partial_ids is not None:
    # reference models.py:807-829 (multi :2267-2287): gather label logits [0]+ids, box logits ids-1
    ids = [int(i) for i in partial_ids]
    cls = np.ascontiguousarray(cls[:, [0] + ids])
    box = np.ascontiguousarray(box[:, [i - 1 for i in ids], :])
  return cls, box


def mask_head(roi_feat14, weights, final_labels):
  """reference models.py:1173-1199 (maskrcnn_up4conv_head: 4 x conv3x3+ReLU, Conv2DTranspose
  2x2 stride 2 + ReLU, conv1x1 to num_class-1) and :951-962 (gather each detection's own class,
  sigmoid) -> final_masks [R,28,28].  Conv2DTranspose kernel is [kh,kw,out,in] (nn.py:383-413),
  torch wants [in,out,kh,kw]."""
  import torch.nn.functional as TFn
  x = torch.from_numpy(np.ascontiguousarray(roi_feat14))
  with torch.no_grad():
    for k in range(4):
      W = _w(weights, "maskrcnn/fcn%d/W" % k).permute(3, 2, 0, 1).contiguous()
      x = torch.relu(TFn.conv2d(x, W, _w(weights, "maskrcnn/fcn%d/b" % k), padding=1))
    Wd = _w(weights, "maskrcnn/deconv/W").permute(3, 2, 0, 1).contiguous()
    x = torch.relu(TFn.conv_transpose2d(x, Wd, _w(weights, "maskrcnn/deconv/b"), stride=2))
    Wc = _w(weights, "maskrcnn/conv/W").permute(3, 2, 0, 1).contiguous()
    logits = TFn.conv2d(x, Wc, _w(weights, "maskrcnn/conv/b")).numpy()        # [R,C-1,28,28]
  idx = np.asarray(final_labels, np.int64) - 1
  sel = logits[np.arange(logits.shape[0]), idx]
  return (F(1.0) / (F(1.0) + np.exp(-sel))).astype(F), logits


def head_decode(rcnn_boxes, box_logits, cls_logits, hw, reg_weights):
  """reference models.py:828-843: decode with /[10,10,5,5] and the DEFAULT clip
  log(1333/16) (nn.py:1518), clip to image, softmax."""
  K, Cm1, _ = box_logits.shape
  anchors = np.repeat(rcnn_boxes[:, None, :], Cm1, 1)
  dec = decode_bbox_target(box_logits / np.asarray(reg_weights, F), anchors,
                           np.log(1333 / 16.0))
  dec = clip_boxes(dec, hw).reshape(K, Cm1, 4)
  return dec, tfops.softmax(cls_logits)


def fastrcnn_predictions(boxes, probs, score_thres, per_im, iou_thres):
  """reference models.py:1202-1223,1258-1304 -> (pred_indices [R,2] (box,
  class), final_probs [R]).  Canonical final order: prob desc, then class,
  then box (tf.where order)."""
  K, Cm1, _ = boxes.shape
  mask = np.zeros((Cm1, K), bool)
  for c in range(Cm1):
    p = probs[:, c + 1]
    ids = np.where(p > F(score_thres))[0]
    sel = tfops.non_max_suppression(boxes[ids, c], p[ids], per_im, iou_thres)
    mask[c, ids[sel]] = True
  sel_idx = np.argwhere(mask)                 # [n,2] (class, box) class-major
  pm = probs[:, 1:].T[mask]
  k = min(per_im, pm.size)
  tk = tfops.top_k(pm, k)
  filt = sel_idx[tk][:, ::-1]                 # -> (box, class)
  return filt.astype(np.int64), pm[tk].astype(F), mask


class OracleModel(object):
  """CPU restatement of Mask_RCNN_FPN / Mask_RCNN_FPN_multi (inference)."""

  def __init__(self, config, weights):
    self.config = config
    self.weights = weights
    self.anchors = all_anchors_fpn(config)
    self.partial_ids = None
    if getattr(config, "use_partial_classes", False):
      self.partial_ids = [config.classname2id[n] for n in config.partial_classes]

  # -- shared trunk ---------------------------------------------------------
  def trunk(self, images, taps):
    cfg = self.config
    with torch.no_grad():
      img = preprocess(images)
      H, W = img.shape[2:]
      c2345 = backbone(img, self.weights, cfg, taps)
      p = fpn(c2345, self.weights)
      # slice_feature_and_anchors (models.py:372-400)
      anchors = []
      for i, s in enumerate(cfg.anchor_strides):
        if i < 3:
          th = int(np.ceil(np.float32(H) * np.float32(1.0 / s)))
          tw = int(np.ceil(np.float32(W) * np.float32(1.0 / s)))
          p[i] = p[i][:, :, :th, :tw]
        h, w = p[i].shape[2:]
        anchors.append(self.anchors[i][:h, :w])
      rpn = [rpn_head(pi, self.weights) for pi in p]
    for i, c in enumerate(c2345):
      taps["c%d" % (i + 2)] = c.numpy()
    pn = [np.ascontiguousarray(x.numpy()) for x in p]
    for i, x in enumerate(pn):
      taps["p%d" % (i + 2)] = x
    for i, (l, b) in enumerate(rpn):
      taps["rpn_logits%d" % (i + 2)] = l
      taps["rpn_deltas%d" % (i + 2)] = b
    return (H, W), pn, anchors, rpn

  # -- b = 1 ----------------------------------------------------------------
  def forward(self, image):
    """image [H,W,3] (uint8/float32 BGR).  Returns dict with final_boxes,
    final_labels (int64), final_probs, fpn_box_feat + stage taps."""
    cfg = self.config
    taps = {}
    hw, p, anchors, rpn = self.trunk(np.asarray(image)[None], taps)
    K = cfg.rpn_test_post_nms_topk
    ab, asc = [], []
    for lvl in range(5):
      logits, deltas = rpn[lvl][0][0], rpn[lvl][1][0]
      dec = decode_bbox_target(deltas, anchors[lvl], cfg.bbox_decode_clip)
      b, s, dbg = rpn_proposals_level_b1(dec, logits.reshape(-1), hw, K,
                                         cfg.rpn_proposal_nms_thres)
      taps["rpn_lvl%d" % (lvl + 2)] = dbg
      ab.append(b); asc.append(s)
    ab = np.concatenate(ab, 0); asc = np.concatenate(asc, 0)
    tk = tfops.top_k(asc, min(asc.size, K))
    props, pscores = ab[tk], asc[tk]
    taps["proposals"] = props; taps["proposal_scores"] = pscores
    zeros = np.zeros((props.shape[0],), np.int32)
    rf = multilevel_roi_align(p[:4], props, zeros, cfg.anchor_strides)
    taps["roi_feat"] = rf
    cls, box = box_head(rf, self.weights, cfg.num_class, self.partial_ids,
                        getattr(cfg, "use_frcnn_class_agnostic", False))
    taps["cls_logits"] = cls; taps["box_logits"] = box
    dec, probs = head_decode(props, box, cls, hw, cfg.fastrcnn_bbox_reg_weights)
    taps["decoded_boxes"] = dec; taps["label_probs"] = probs
    pi, fp, mask = fastrcnn_predictions(dec, probs, cfg.result_score_thres,
                                        cfg.result_per_im,
                                        cfg.fastrcnn_nms_iou_thres)
    taps["nms_mask"] = mask
    fb = dec[pi[:, 0], pi[:, 1]]
    fl = (pi[:, 1] + 1).astype(np.int64)
    feat = multilevel_roi_align(p[:4], fb, np.zeros((fb.shape[0],), np.int32),
                                cfg.anchor_strides)
    taps.update(final_boxes=fb, final_labels=fl, final_probs=fp,
                fpn_box_feat=feat, pred_indices=pi)
    if getattr(cfg, "add_mask", False):          # models.py:932-962
      rf14 = multilevel_roi_align(p[:4], fb, np.zeros((fb.shape[0],), np.int32),
                                  cfg.anchor_strides, out=14)
      taps["final_masks"], taps["mask_logits"] = mask_head(rf14, self.weights, fl)
    return taps

  # -- b = B ----------------------------------------------------------------
  def forward_multi(self, images):
    """images [B,H,W,3].  Returns final_boxes [B,100,4], final_labels [B,100]
    float32, final_probs [B,100], final_valid_indices [B] int32, fpn_box_feat
    [M,256,7,7] (reference models.py:2246-2378, nn.py:1406-1482)."""
    cfg = self.config
    taps = {}
    hw, p, anchors, rpn = self.trunk(np.asarray(images), taps)
    B = p[0].shape[0]
    K = cfg.rpn_test_post_nms_topk
    lb, ls = [], []
    for lvl in range(5):
      logits, deltas = rpn[lvl]
      n = logits[0].size
      k = min(K, n)
      bb = np.zeros((B, k, 1, 4), F); ss = np.zeros((B, k, 1), F)
      for b in range(B):
        dec = decode_bbox_target(deltas[b], anchors[lvl], cfg.bbox_decode_clip)
        sc = logits[b].reshape(-1)
        idx = tfops.top_k(sc, k)
        bb[b, :, 0] = clip_boxes(dec[idx], hw); ss[b, :, 0] = sc[idx]
      nb, ns, _, nv = tfops.combined_non_max_suppression(
          bb, ss, K, K, cfg.rpn_proposal_nms_thres)
      taps["rpn_lvl%d" % (lvl + 2)] = dict(nms_in_boxes=bb[:, :, 0],
                                           nms_in_scores=ss[:, :, 0],
                                           out_boxes=nb, out_scores=ns, valid=nv)
      lb.append(nb); ls.append(ns)
    lb = np.concatenate(lb, 1); ls = np.concatenate(ls, 1)   # [B,5K,..]
    kk = min(ls.shape[1], K)
    props = []
    for b in range(B):
      tk = tfops.top_k(ls[b], kk)
      pb = lb[b][tk]
      props.append(np.concatenate([np.full((kk, 1), b, F), pb], 1))
    props = np.concatenate(props, 0)                           # [B*kk,5]
    area = (props[:, 4] - props[:, 2]) * (props[:, 3] - props[:, 1])
    props = props[area > 0]
    taps["proposals"] = props
    bidx = props[:, 0].astype(np.int32); rb = props[:, 1:]
    rf = multilevel_roi_align(p[:4], rb, bidx, cfg.anchor_strides)
    taps["roi_feat"] = rf
    cls, box = box_head(rf, self.weights, cfg.num_class, self.partial_ids,
                        getattr(cfg, "use_frcnn_class_agnostic", False))
    dec, probs = head_decode(rb, box, cls, hw, cfg.fastrcnn_bbox_reg_weights)
    taps["decoded_boxes"] = dec; taps["label_probs"] = probs
    # fastrcnn_predictions_multibatch (models.py:2924-2976): scatter into
    # [B,M,C-1,*] zero-padded slots, combined NMS with score_threshold=-inf.
    M = dec.shape[0]; Cm1 = dec.shape[1]
    pbx = np.zeros((B, M, Cm1, 4), F); ppr = np.zeros((B, M, Cm1), F)
    pbx[bidx, np.arange(M)] = dec; ppr[bidx, np.arange(M)] = probs[:, 1:]
    nb, ns, ncls, nv = tfops.combined_non_max_suppression(
        pbx, ppr, cfg.result_per_im, cfg.result_per_im,
        cfg.fastrcnn_nms_iou_thres)
    ncls = ncls + F(1)
    sel = []
    for b in range(B):
      for t in range(int(nv[b])):
        sel.append(np.concatenate([[F(b)], nb[b, t]]))
    sel = np.asarray(sel, F).reshape(-1, 5)
    feat = multilevel_roi_align(p[:4], sel[:, 1:], sel[:, 0].astype(np.int32),
                                cfg.anchor_strides)
    taps.update(final_boxes=nb, final_labels=ncls, final_probs=ns,
                final_valid_indices=nv, fpn_box_feat=feat)
    return taps
