"""Host-side frame helpers of the reference's nn.py that sit in front of the forward.

``get_new_hw`` / ``resizeImage`` mirror reference nn.py:1540-1560: scale the short edge to
``short_size`` unless the long edge would exceed ``max_size``; sizes are rounded with
``int(x + 0.5)``; when the frame already has the target size it is returned unchanged (the native
1920x1080 case of the benchmark), otherwise it is resized with bilinear interpolation.  The
reference calls ``cv2.resize(..., interpolation=cv2.INTER_LINEAR)``; OpenCV is not a dependency
here, so the same sampling rule is restated in numpy (pixel centres at (i + 0.5) * scale - 0.5,
source index clamped to the image, weights (1 - f, f)); cv2 is not installed in the build container:
the rule is pinned by its published 2x2 -> 4x4 table and, through oracle/imgproc.py, by torch's
bilinear / align_corners=False (tests/test_oracle_golden.py).
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
    newh = newh * scale
    neww = neww * scale
  return int(neww + 0.5), int(newh + 0.5)


def _axis_taps(n_src, n_dst):
  f = (np.arange(n_dst, dtype=np.float64) + 0.5) * (float(n_src) / n_dst) - 0.5
  i0 = np.floor(f).astype(np.int64)
  w1 = f - i0
  w1[i0 < 0] = 0.0
  i0 = np.maximum(i0, 0)
  over = i0 >= n_src - 1
  i0[over] = n_src - 1
  w1[over] = 0.0
  i1 = np.minimum(i0 + 1, n_src - 1)
  return i0, i1, w1.astype(np.float32)


def resizeImage(im, short_size, max_size):
  """reference nn.py:1540-1546."""
  h, w = im.shape[:2]
  neww, newh = get_new_hw(h, w, short_size, max_size)
  if h == newh and w == neww:
    return im
  src = np.asarray(im, dtype=np.float32)
  y0, y1, wy = _axis_taps(h, newh)
  x0, x1, wx = _axis_taps(w, neww)
  wy = wy[:, None, None]; wx = wx[None, :, None]
  if src.ndim == 2:
    src = src[:, :, None]
  top = src[y0][:, x0] * (1 - wx) + src[y0][:, x1] * wx
  bot = src[y1][:, x0] * (1 - wx) + src[y1][:, x1] * wx
  out = top * (1 - wy) + bot * wy
  if im.ndim == 2:
    out = out[:, :, 0]
  return out.astype(im.dtype) if np.issubdtype(im.dtype, np.floating) else \
      np.clip(np.rint(out), 0, 255).astype(im.dtype)
