"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of the reference's frame resize.

reference nn.py:1540-1560: ``resizeImage(im, short_size, max_size)`` = ``get_new_hw`` (scale the short edge to
``short_size`` unless the long edge would pass ``max_size``; sizes rounded with ``int(x + 0.5)``) followed by
``cv2.resize(im, (neww, newh), interpolation=cv2.INTER_LINEAR)``; obj_detect_tracking.py:597-608 feeds it the
frame as float32.  cv2 is not installable in the build container, so the INTER_LINEAR rule is restated from its
definition -- destination pixel centre (d + 0.5) * (src / dst) - 0.5, the two neighbouring source taps with weights
(1 - f, f), taps clamped to the image -- in float64, one output pixel at a time (nothing shared with the package's
vectorised float32 restatement in object_detection_tracking_amd/nn.py, which the device kernel follows operation by
operation).  Pinned by tests/test_oracle_golden.py: the rule's published 2x2 -> 4x4 table, and
torch.nn.functional.interpolate(mode="bilinear", align_corners=False) -- an independent implementation of the same
half-pixel rule -- on random images, up- and down-scaling.
"""
import numpy as np


def get_new_hw(h, w, size, max_size):
  """reference nn.py:1548-1560 -> (neww, newh)."""
  scale = size * 1.0 / min(h, w)
  if h < w:
    newh, neww = size, scale * w
  else:
    newh, neww = scale * h, size
  if max(newh, neww) > max_size:
    scale = max_size * 1.0 / max(newh, neww)
    newh, neww = newh * scale, neww * scale
  return int(neww + 0.5), int(newh + 0.5)


def _taps(n_src, n_dst, d):
  f = (d + 0.5) * (float(n_src) / float(n_dst)) - 0.5
  i0 = int(np.floor(f))
  frac = f - i0
  if i0 < 0:
    i0, frac = 0, 0.0
  if i0 >= n_src - 1:
    i0, frac = n_src - 1, 0.0
  return i0, min(i0 + 1, n_src - 1), frac


def inter_linear(im, newh, neww):
  """cv2.resize(im, (neww, newh), interpolation=cv2.INTER_LINEAR) for a float image [H,W(,C)] -> float32."""
  src = np.asarray(im, dtype=np.float64)
  if src.ndim == 2:
    src = src[:, :, None]
  h, w, c = src.shape
  out = np.zeros((newh, neww, c), np.float64)
  xt = [_taps(w, neww, x) for x in range(neww)]
  for y in range(newh):
    y0, y1, fy = _taps(h, newh, y)
    for x in range(neww):
      x0, x1, fx = xt[x]
      top = src[y0, x0] * (1.0 - fx) + src[y0, x1] * fx
      bot = src[y1, x0] * (1.0 - fx) + src[y1, x1] * fx
      out[y, x] = top * (1.0 - fy) + bot * fy
  out = out.astype(np.float32)
  return out[:, :, 0] if np.asarray(im).ndim == 2 else out


def resize_image(im, short_size, max_size):
  """reference nn.py:1540-1546 on a float32 frame."""
  h, w = im.shape[:2]
  neww, newh = get_new_hw(h, w, short_size, max_size)
  if (newh, neww) == (h, w):
    return np.asarray(im, np.float32)
  return inter_linear(im, newh, neww)
