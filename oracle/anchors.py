"""ORACLE (test infrastructure only -- see oracle/__init__.py).

Restatement of the reference's FPN anchor grid:
  generate_anchors.py:42-109 (cell anchors: ratio enumeration with np.round,
  then scale enumeration), utils.py:606-658 (get_all_anchors: shift grid of
  ceil(max_size/stride) cells, float32 cast, then x2,y2 += 1) and
  models.py:359-369 (one size per level).

Pinned by the known-answer table in generate_anchors.py:20-38 (python output =
table - 1) and by fixtures generated from the reference module itself
(tests/golden/anchors_ref.npz).
"""
import numpy as np


def _whctrs(a):
  w = a[2] - a[0] + 1.0
  h = a[3] - a[1] + 1.0
  return w, h, a[0] + 0.5 * (w - 1), a[1] + 0.5 * (h - 1)


def _mk(ws, hs, xc, yc):
  ws = np.asarray(ws, np.float64).reshape(-1, 1)
  hs = np.asarray(hs, np.float64).reshape(-1, 1)
  return np.hstack((xc - 0.5 * (ws - 1), yc - 0.5 * (hs - 1),
                    xc + 0.5 * (ws - 1), yc + 0.5 * (hs - 1)))


def cell_anchors(base_size, ratios, scales):
  """generate_anchors.py:42-56: base window (0,0,base-1,base-1) -> ratio
  anchors (rounded w,h) -> scaled anchors; row order ratio-major."""
  ratios = np.asarray(ratios, np.float64); scales = np.asarray(scales, np.float64)
  base = np.array([1, 1, base_size, base_size], np.float32) - 1
  w, h, xc, yc = _whctrs(base.astype(np.float64))
  size_ratios = (w * h) / ratios
  ws = np.round(np.sqrt(size_ratios))
  hs = np.round(ws * ratios)
  ratio_anchors = _mk(ws, hs, xc, yc)
  out = []
  for r in ratio_anchors:
    w, h, xc, yc = _whctrs(r)
    out.append(_mk(w * scales, h * scales, xc, yc))
  return np.vstack(out)


def all_anchors(stride, sizes, ratios, max_size):
  """utils.py:606-658 -> [S,S,A,4] float32, S = ceil(max_size/stride)."""
  cell = cell_anchors(stride, ratios, np.asarray(sizes, np.float64) / stride)
  fs = int(np.ceil(max_size / stride))
  shifts = np.arange(0, fs) * stride
  sx, sy = np.meshgrid(shifts, shifts)
  sh = np.vstack((sx.ravel(), sy.ravel(), sx.ravel(), sy.ravel())).T
  A = cell.shape[0]
  field = cell.reshape(1, A, 4) + sh.reshape(1, -1, 4).transpose(1, 0, 2)
  field = field.reshape(fs, fs, A, 4).astype(np.float32)
  field[:, :, :, [2, 3]] += 1
  return field


def all_anchors_fpn(config):
  """models.py:359-369: one anchor size per FPN level."""
  return [all_anchors(s, [z], config.anchor_ratios, config.max_size)
          for s, z in zip(config.anchor_strides, config.anchor_sizes)]
