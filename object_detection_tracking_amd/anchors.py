"""Host-side FPN anchor field (computed once, uploaded as "anchors/lvl<i>").

Same result as the reference's utils.get_all_anchors / generate_anchors.py
(reference utils.py:606-658, generate_anchors.py:42-109, models.py:359-369):
per level one anchor size, three ratios with integer-rounded widths/heights,
grid shift of `stride` over ceil(max_size/stride) cells, x2,y2 += 1.
"""
from __future__ import annotations

import numpy as np


def _centered(ws, hs, xc, yc):
  ws = np.asarray(ws, np.float64)[:, None]; hs = np.asarray(hs, np.float64)[:, None]
  return np.hstack([xc - (ws - 1) / 2, yc - (hs - 1) / 2, xc + (ws - 1) / 2, yc + (hs - 1) / 2])


def level_cell_anchors(stride, size, ratios):
  """[A,4] anchors of the cell at the origin (ratio-major rows)."""
  ctr = (stride - 1) / 2.0
  ratios = np.asarray(ratios, np.float64)
  ws = np.round(np.sqrt(stride * stride / ratios))
  hs = np.round(ws * ratios)
  scale = float(size) / stride
  return _centered(ws * scale, hs * scale, ctr, ctr)


def fpn_anchor_fields(config):
  """list of float32 [S_l,S_l,A,4], S_l = ceil(max_size / stride_l)."""
  out = []
  for stride, size in zip(config.anchor_strides, config.anchor_sizes):
    cell = level_cell_anchors(stride, size, config.anchor_ratios)
    fs = int(np.ceil(config.max_size / stride))
    sh = np.arange(fs) * stride
    field = np.zeros((fs, fs, cell.shape[0], 4), np.float64)
    field[..., 0] = sh[None, :, None] + cell[None, None, :, 0]
    field[..., 1] = sh[:, None, None] + cell[None, None, :, 1]
    field[..., 2] = sh[None, :, None] + cell[None, None, :, 2]
    field[..., 3] = sh[:, None, None] + cell[None, None, :, 3]
    field = field.astype(np.float32)
    field[..., 2:] += 1
    out.append(field)
  return out
