"""numpy-in / numpy-out wrappers of the stand-alone op entry points of the C ABI
(include/odt.h ``odt_op_*``).  Each runs exactly the HIP kernels ``odt_forward``
uses; the staged parity tests call the kernels through these.  ``lib`` defaults
to the product library (libodt_hip.so, GPU required).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import RPN_CH, c_float_p, f32, fptr, i32, iptr


def _L(lib):
  return lib if lib is not None else _lib.get_lib()


def conv2d(x_nhwc, w_hwio, bias=None, stride=1, dil=1, pad_t=0, pad_l=0, out_hw=None,
           out_off=(0, 0), res=None, res_mode=0, relu=False, lib=None, device=0):
  """reference nn.py:337-381 (+ folded-BN bias, residual, ReLU epilogue)."""
  lib = _L(lib)
  x = f32(x_nhwc); w = f32(w_hwio)
  B, H, W, Cin = x.shape
  kh, kw, _, Cout = w.shape
  if out_hw is None:
    ke_h, ke_w = (kh - 1) * dil + 1, (kw - 1) * dil + 1
    out_hw = ((H + 2 * pad_t - ke_h) // stride + 1, (W + 2 * pad_l - ke_w) // stride + 1)
  Ho, Wo = out_hw
  oy, ox = out_off
  out = np.zeros((B, Ho + oy, Wo + ox, Cout), np.float32)
  b = f32(bias) if bias is not None else None
  r = f32(res) if res is not None else None
  lib.check(lib.dll.odt_op_conv2d(device, fptr(x), B, H, W, Cin, fptr(w),
                                  fptr(b) if b is not None else None, kh, kw, Cout, stride, dil,
                                  pad_t, pad_l, Ho, Wo, oy, ox,
                                  fptr(r) if r is not None else None, res_mode, int(relu),
                                  fptr(out)))
  return out


def conv2d_cat(a, b2, wa, wb, bias=None, stride_b=1, relu=False, lib=None, device=0):
  """1x1 conv over the K-concatenation [a | b2[:, ::stride_b, ::stride_b]] (fused conv3 +
  convshortcut, reference nn.py:503-521)."""
  lib = _L(lib)
  a = f32(a); b2 = f32(b2); wa = f32(wa); wb = f32(wb)
  B, Ho, Wo, Ca = a.shape
  _, Hb, Wb, Cb = b2.shape
  Cout = wa.shape[1]
  out = np.zeros((B, Ho, Wo, Cout), np.float32)
  bb = f32(bias) if bias is not None else None
  lib.check(lib.dll.odt_op_conv2d_cat(device, fptr(a), B, Ho, Wo, Ca, fptr(b2), Hb, Wb, Cb, stride_b,
                                      fptr(wa), fptr(wb), fptr(bb) if bb is not None else None, Cout,
                                      int(relu), fptr(out)))
  return out


def bottleneck_tail(x, w2, b2, w3, b3, res=None, dil=1, relu3=True, fuse=True, lib=None, device=0):
  """conv2 (3x3, 'SAME', ReLU) -> conv3 (1x1 (+ res), ReLU) of a bottleneck block (reference nn.py:503-521) on the fp16x2
  kernels; fuse=True: conv3 inside the 3x3 kernel (one launch), False: the two launches it replaces."""
  lib = _L(lib)
  x = f32(x); w2 = f32(w2); b2 = f32(b2); w3 = f32(w3); b3 = f32(b3)
  B, H, W, Cc = x.shape
  C3 = w3.shape[1]
  r = f32(res) if res is not None else None
  out = np.zeros((B, H, W, C3), np.float32)
  lib.check(lib.dll.odt_op_bottleneck_tail(device, fptr(x), B, H, W, Cc, fptr(w2), fptr(b2), dil, fptr(w3), fptr(b3), C3,
                                           fptr(r) if r is not None else None, int(relu3), int(fuse), fptr(out)))
  return out


def stem(frame_pad, w_hwio, bias, fuse=True, grid=0, lib=None, device=0):
  """conv0 (7x7 stride 2 VALID + bias + ReLU) -> pool0 (3x3 stride 2 max over the top/left zero-padded map) on a padded frame
  tensor [B, Hp, Wp, 3] (reference nn.py:860-896, 784-792), fp16x2 arithmetic; fuse=True: conv_stem_kernel (one launch),
  False: the two launches it replaces.  Returns [B, Hq, Wq, 64]."""
  lib = _L(lib)
  x = f32(frame_pad); w = f32(w_hwio); b = f32(bias)
  B, Hp, Wp, _ = x.shape
  Ho0, Wo0 = (Hp - 7) // 2 + 1, (Wp - 7) // 2 + 1
  Hq, Wq = (Ho0 + 1 - 3) // 2 + 1, (Wo0 + 1 - 3) // 2 + 1
  out = np.zeros((B, Hq, Wq, 64), np.float32)
  lib.check(lib.dll.odt_op_stem(device, fptr(x), B, Hp, Wp, fptr(w), fptr(b), int(fuse), int(grid), fptr(out)))
  return out


def preprocess(frames, pad_t, pad_l, Hp, Wp, lib=None, device=0):
  """reference models.py:340-355 + zero pad; returns [B,Hp,Wp,4]."""
  lib = _L(lib)
  fr = np.ascontiguousarray(frames)
  assert fr.dtype in (np.uint8, np.float32)
  B, H, W, _ = fr.shape
  out = np.zeros((B, Hp, Wp, 4), np.float32)
  lib.check(lib.dll.odt_op_preprocess(device, fr.ctypes.data_as(C.c_void_p),
                                      0 if fr.dtype == np.uint8 else 1, B, H, W, pad_t, pad_l,
                                      Hp, Wp, fptr(out)))
  return out


def maxpool3x3s2(x_nhwc, lib=None, device=0):
  """reference nn.py:890-896."""
  lib = _L(lib)
  x = f32(x_nhwc)
  B, H, W, Cc = x.shape
  out = np.zeros((B, (H + 1 - 3) // 2 + 1, (W + 1 - 3) // 2 + 1, Cc), np.float32)
  lib.check(lib.dll.odt_op_maxpool(device, fptr(x), B, H, W, Cc, fptr(out)))
  return out


def top_k(scores, k, lib=None, device=0):
  lib = _L(lib)
  s = f32(scores).reshape(-1)
  idx = np.zeros((k,), np.int32)
  lib.check(lib.dll.odt_op_topk(device, fptr(s), s.size, k, iptr(idx)))
  return idx


def nms(boxes, scores, max_out, iou_thresh, lib=None, device=0):
  lib = _L(lib)
  b = f32(boxes).reshape(-1, 4); s = f32(scores).reshape(-1)
  idx = np.zeros((max(1, s.size),), np.int32)
  n = C.c_int(0)
  lib.check(lib.dll.odt_op_nms(device, fptr(b), fptr(s), s.size, max_out, iou_thresh, iptr(idx),
                               C.byref(n)))
  return idx[:n.value].copy()


def pack_rpn(logits, deltas):
  """[B,h,w,3] + [B,h,w,3,4] -> the device layout [B,h,w,16]."""
  B, h, w, A = logits.shape
  out = np.zeros((B, h, w, RPN_CH), np.float32)
  out[..., :A] = logits
  out[..., A:A + 4 * A] = np.asarray(deltas, np.float32).reshape(B, h, w, 4 * A)
  return out


def proposals(graph, rpn_levels, anchors, img_hw, K, nms_thresh, decode_clip, lib=None, device=0):
  """generate_fpn_proposals (reference models.py:402-436 / :2458-2522).
  rpn_levels: list of [B,h,w,16]; anchors: list of [S,S,3,4].  -> (props [B,K,4], nprops [B])."""
  lib = _L(lib)
  L = len(rpn_levels)
  rp = [f32(r) for r in rpn_levels]; an = [f32(a) for a in anchors]
  B = rp[0].shape[0]
  hs = i32([r.shape[1] for r in rp]); ws = i32([r.shape[2] for r in rp])
  fs = i32([a.shape[0] for a in an])
  rpp = (c_float_p * L)(*[fptr(r) for r in rp]); anp = (c_float_p * L)(*[fptr(a) for a in an])
  props = np.zeros((B, K, 4), np.float32); nprops = np.zeros((B,), np.int32)
  lib.check(lib.dll.odt_op_proposals(device, graph, B, L, iptr(hs), iptr(ws), iptr(fs), rpp, anp,
                                     int(img_hw[0]), int(img_hw[1]), K, nms_thresh, decode_clip,
                                     fptr(props), iptr(nprops)))
  return props, nprops


def roi_align(feats_nhwc, strides, boxes, box_ind, lib=None, device=0):
  """multilevel_roi_align (reference models.py:465-485) -> ([R,C,7,7], [R,C])."""
  lib = _L(lib)
  ft = [f32(x) for x in feats_nhwc]
  B, _, _, Cc = ft[0].shape
  hs = i32([x.shape[1] for x in ft]); ws = i32([x.shape[2] for x in ft])
  fp = (c_float_p * 4)(*[fptr(x) for x in ft])
  st = f32(strides); bx = f32(boxes).reshape(-1, 4); bi = i32(box_ind)
  R = bx.shape[0]
  out = np.zeros((R, Cc, 7, 7), np.float32); pooled = np.zeros((R, Cc), np.float32)
  lib.check(lib.dll.odt_op_roi_align(device, B, Cc, iptr(hs), iptr(ws), fp, fptr(st), fptr(bx),
                                     iptr(bi), R, fptr(out), fptr(pooled)))
  return out, pooled


def detections(graph, cls_logits, box_logits, props, nprops, img_hw, reg_weights, decode_clip,
               score_thresh, nms_thresh, per_im, lib=None, device=0):
  """inference tail (reference models.py:828-843, :1258-1304 / :2924-2976).
  cls_logits [B*K,C], box_logits [B*K,C,4] (class 0 ignored), props [B,K,4]."""
  lib = _L(lib)
  pr = f32(props); B, K, _ = pr.shape
  cl = f32(cls_logits); Cn = cl.shape[1]
  bl = f32(box_logits).reshape(B * K, Cn * 4)
  npz = i32(nprops); rw = f32(reg_weights)
  boxes = np.zeros((B, per_im, 4), np.float32); probs = np.zeros((B, per_im), np.float32)
  labels = np.zeros((B, per_im), np.int32); valid = np.zeros((B,), np.int32)
  lib.check(lib.dll.odt_op_detections(device, graph, B, K, Cn, fptr(cl), fptr(bl), fptr(pr),
                                      iptr(npz), int(img_hw[0]), int(img_hw[1]), fptr(rw),
                                      decode_clip, score_thresh, nms_thresh, per_im, fptr(boxes),
                                      fptr(probs), iptr(labels), iptr(valid)))
  return boxes, probs, labels, valid


def class_nms(graph, boxes, scores, per_im, iou_thresh, score_thresh=0.0, ncand=None, lib=None, device=0):
  """The selection half of the tail on caller-supplied data: per-class NMS + merged top ``per_im``
  (tf.image.combined_non_max_suppression for graph ODT_GRAPH_MULTI, nms_return_masks + fastrcnn_predictions for
  ODT_GRAPH_SINGLE).  boxes [B,N,C,4] (or [B,N,1,4], shared by the classes), scores [B,N,C].
  Returns (boxes [B,per_im,4], scores [B,per_im], classes [B,per_im] 0-based, valid [B])."""
  lib = _L(lib)
  sc = f32(scores); B, N, Cn = sc.shape
  bx = f32(boxes)
  if bx.shape[2] == 1 and Cn > 1:
    bx = f32(np.repeat(bx, Cn, axis=2))
  nc = i32(ncand if ncand is not None else np.full((B,), N))
  ob = np.zeros((B, per_im, 4), np.float32); os_ = np.zeros((B, per_im), np.float32)
  ol = np.zeros((B, per_im), np.int32); ov = np.zeros((B,), np.int32)
  lib.check(lib.dll.odt_op_class_nms(device, graph, B, N, Cn, fptr(bx), fptr(sc), iptr(nc), score_thresh, iou_thresh,
                                     per_im, fptr(ob), fptr(os_), iptr(ol), iptr(ov)))
  return ob, os_, ol - 1, ov


def nn_cosine(gallery, seg_offsets, dets, lib=None, device=0):
  """NearestNeighborDistanceMetric.distance, cosine (reference
  deep_sort/nn_matching.py:156-177) -> float64 [T,N]."""
  lib = _L(lib)
  g = f32(gallery); d = f32(dets); s = i32(seg_offsets)
  T = s.size - 1
  N = d.shape[0] if d.ndim == 2 else 0
  cost = np.zeros((T, N), np.float64)
  if T == 0 or N == 0:
    return cost
  lib.check(lib.dll.odt_nn_cosine(device, fptr(g), iptr(s), T, fptr(d), N, d.shape[1],
                                  cost.ctypes.data_as(_lib.c_double_p)))
  return cost
