"""Weight ingest for the detector: Tensorpack-style ``.npz`` (the only format
the reference can load without TensorFlow, reference
obj_detect_tracking.py:417-435) and a seeded synthetic generator that emits
exactly the same variable names / layouts:

  conv ``W``  : HWIO  [kh, kw, Cin, Cout]        (reference nn.py:350,367)
  dense ``W`` : [in, out], ``in`` flattened NCHW (reference nn.py:736-757)
  BN          : gamma, beta, mean/EMA, variance/EMA (reference nn.py:1821-1839)

No network and no checkpoints ship with the reference, so benchmarks and
parity tests use :func:`synthetic_weights` (random-init weights of the real
architecture).
"""
from __future__ import annotations

import numpy as np


def _strip(name):
  return name[:-2] if name.endswith(":0") else name


def load_npz(path):
  """Load a Tensorpack/zoo ``.npz`` into {name: float32 array} (names without
  the ``:0`` suffix the reference adds back, obj_detect_tracking.py:419-420)."""
  with np.load(path) as z:
    return {_strip(k): np.asarray(z[k]) for k in z.files}


def backbone_conv_specs(config):
  """Yield (scope, kh, cin, cout, has_bn, has_bias) for every conv on the path,
  in execution order (reference nn.py:843-1014, models.py:979-1009)."""
  blocks = config.resnet_num_block
  yield ("conv0", 7, 3, 64, True, False)
  cin = 64
  for g, (feat, cnt) in enumerate(zip((64, 128, 256, 512), blocks)):
    for i in range(cnt):
      pre = "group%d/block%d" % (g, i)
      yield (pre + "/conv1", 1, cin, feat, True, False)
      yield (pre + "/conv2", 3, feat, feat, True, False)
      yield (pre + "/conv3", 1, feat, feat * 4, True, False)
      if i == 0:
        yield (pre + "/convshortcut", 1, cin, feat * 4, True, False)
      cin = feat * 4
  ch = config.fpn_num_channel
  for i, c in enumerate((256, 512, 1024, 2048)):
    yield ("fpn/lateral_1x1_c%d" % (i + 2), 1, c, ch, False, True)
  for i in range(4):
    yield ("fpn/posthoc_3x3_p%d" % (i + 2), 3, ch, ch, False, True)
  na = len(config.anchor_ratios)
  yield ("rpn/conv0", 3, ch, ch, False, True)
  yield ("rpn/class", 1, ch, na, False, True)
  yield ("rpn/box", 1, ch, 4 * na, False, True)


def synthetic_weights(config, seed=0):
  """Seeded random-init weights with the reference's names and layouts.

  Distributions follow SURVEY.md section 8(d): He-init convs, BN close to
  identity (so activations stay O(1) through 101 layers), the last BN gamma of
  each bottleneck scaled by 0.12, RPN class bias +1 (so the top-K logits are
  positive and the multibatch zero-padding quirk does not starve the head), box-head ``class`` /
  ``box`` weights wide enough that several classes pass the 1e-4 score filter
  and boxes actually move.
  """
  rng = np.random.default_rng(seed)
  w = {}

  def normal(shape, std):
    return (rng.standard_normal(shape, dtype=np.float32) * np.float32(std))

  for scope, k, cin, cout, has_bn, has_bias in backbone_conv_specs(config):
    fan_in = k * k * cin
    std = np.sqrt(2.0 / fan_in)
    if scope.startswith("fpn/lateral"):
      std = 0.35 * np.sqrt(1.0 / fan_in)
    if scope.startswith("fpn/posthoc"):
      std = 0.8 * np.sqrt(1.0 / fan_in)
    if scope in ("rpn/class", "rpn/box"):
      std = 0.05 if scope == "rpn/class" else 0.02
    w[scope + "/W"] = normal((k, k, cin, cout), std)
    if has_bias:
      b = normal((cout,), 0.02)
      if scope == "rpn/class":
        b = b + np.float32(1.0)
      w[scope + "/b"] = b
    if has_bn:
      gamma = rng.uniform(0.9, 1.1, cout).astype(np.float32)
      if scope.endswith("/conv3"):
        gamma *= np.float32(0.12)
      w[scope + "/bn/gamma"] = gamma
      w[scope + "/bn/beta"] = normal((cout,), 0.02)
      w[scope + "/bn/mean/EMA"] = normal((cout,), 0.05)
      w[scope + "/bn/variance/EMA"] = rng.uniform(0.8, 1.2, cout).astype(
          np.float32)
  dim = config.fpn_frcnn_fc_head_dim
  ch = config.fpn_num_channel
  nc = config.num_class
  w["fastrcnn/fc6/W"] = normal((ch * 49, dim), np.sqrt(2.0 / (ch * 49)))
  w["fastrcnn/fc6/b"] = normal((dim,), 0.02)
  w["fastrcnn/fc7/W"] = normal((dim, dim), np.sqrt(2.0 / dim))
  w["fastrcnn/fc7/b"] = normal((dim,), 0.02)
  w["fastrcnn/outputs/class/W"] = normal((dim, nc), 0.05)
  w["fastrcnn/outputs/class/b"] = normal((nc,), 0.02)
  if getattr(config, "add_mask", False):           # maskrcnn_up4conv_head (models.py:1173-1199)
    md = getattr(config, "mrcnn_head_dim", 256)
    cin = ch
    for k in range(4):
      w["maskrcnn/fcn%d/W" % k] = normal((3, 3, cin, md), np.sqrt(2.0 / (9 * cin)))
      w["maskrcnn/fcn%d/b" % k] = normal((md,), 0.02)
      cin = md
    w["maskrcnn/deconv/W"] = normal((2, 2, md, md), np.sqrt(2.0 / md))     # [kh, kw, out, in]
    w["maskrcnn/deconv/b"] = normal((md,), 0.02)
    w["maskrcnn/conv/W"] = normal((1, 1, md, nc - 1), np.sqrt(2.0 / md))
    w["maskrcnn/conv/b"] = normal((nc - 1,), 0.02)
  nb = 1 if getattr(config, "use_frcnn_class_agnostic", False) else nc   # models.py:1164
  w["fastrcnn/outputs/box/W"] = normal((dim, nb * 4), 0.01)
  w["fastrcnn/outputs/box/b"] = normal((nb * 4,), 0.002)
  return w


def synthetic_frames(batch, height, width, seed=1234):
  """Seeded synthetic BGR frames, uint8 [B,H,W,3]:
  smooth low-frequency background around 110 plus 40 solid noisy rectangles so
  that RPN / NMS see clustered, overlapping candidates."""
  rng = np.random.default_rng(seed)
  out = np.empty((batch, height, width, 3), np.uint8)
  for b in range(batch):
    gh, gw = height // 64 + 2, width // 64 + 2
    coarse = rng.normal(110.0, 20.0, (gh, gw, 3))
    ys = np.linspace(0, gh - 1.001, height)
    xs = np.linspace(0, gw - 1.001, width)
    y0 = np.floor(ys).astype(int); x0 = np.floor(xs).astype(int)
    fy = (ys - y0)[:, None, None]; fx = (xs - x0)[None, :, None]
    img = (coarse[y0][:, x0] * (1 - fy) * (1 - fx) +
           coarse[y0][:, x0 + 1] * (1 - fy) * fx +
           coarse[y0 + 1][:, x0] * fy * (1 - fx) +
           coarse[y0 + 1][:, x0 + 1] * fy * fx)
    img += rng.normal(0.0, 4.0, img.shape)
    for _ in range(40):
      rw = int(rng.uniform(16, min(400, width // 2)))
      rh = int(rng.uniform(16, min(400, height // 2)))
      x = int(rng.uniform(0, width - rw)); y = int(rng.uniform(0, height - rh))
      col = rng.uniform(0, 255, 3)
      img[y:y + rh, x:x + rw] = col + rng.normal(0, 6.0, (rh, rw, 3))
    out[b] = np.clip(np.rint(img), 0, 255).astype(np.uint8)
  return out


def select_partial_classes(weights, class_ids, num_class):
  """``--use_partial_classes`` (reference models.py:807-829, multi :2267-2287): the graph gathers
  label logits ``[0] + ids`` and box logits ``ids - 1`` (after the BG box was dropped).  Both are
  columns of a linear layer, so gathering the columns of ``fastrcnn/outputs/{class,box}`` once at
  load time yields the identical dot products; the head then runs with ``len(ids) + 1`` classes.
  Returns a new dict (the input is not modified)."""
  cols = np.asarray([0] + [int(i) for i in class_ids], np.int64)
  if cols.min() < 0 or cols.max() >= num_class:
    raise ValueError("partial class id outside [0, %d)" % num_class)
  out = dict(weights)
  W = np.asarray(weights["fastrcnn/outputs/class/W"]); b = np.asarray(weights["fastrcnn/outputs/class/b"])
  out["fastrcnn/outputs/class/W"] = np.ascontiguousarray(W[:, cols])
  out["fastrcnn/outputs/class/b"] = np.ascontiguousarray(b[cols])
  W = np.asarray(weights["fastrcnn/outputs/box/W"]); b = np.asarray(weights["fastrcnn/outputs/box/b"])
  W = W.reshape(W.shape[0], num_class, 4)[:, cols, :]
  out["fastrcnn/outputs/box/W"] = np.ascontiguousarray(W.reshape(W.shape[0], -1))
  out["fastrcnn/outputs/box/b"] = np.ascontiguousarray(b.reshape(num_class, 4)[cols].reshape(-1))
  return out


def expand_class_agnostic_box(weights, num_class):
  """``use_frcnn_class_agnostic`` (model versions 4-6; reference models.py:1126-1170 and
  :798-802): the head regresses ONE box per RoI (``fastrcnn/outputs/box`` is [dim, 4]) and the
  graph tiles it over the ``num_class - 1`` foreground classes.  Tiling the four weight columns
  over the classes once at load time yields the identical dot products for every class slot, so
  the per-class plan runs unchanged.  Returns a new dict."""
  W = np.asarray(weights["fastrcnn/outputs/box/W"]); b = np.asarray(weights["fastrcnn/outputs/box/b"])
  if W.shape[1] != 4 or b.shape[0] != 4:
    raise ValueError("class-agnostic box head expects fastrcnn/outputs/box/W of [dim, 4], got %s"
                     % (W.shape,))
  out = dict(weights)
  out["fastrcnn/outputs/box/W"] = np.ascontiguousarray(np.tile(W, (1, num_class)))
  out["fastrcnn/outputs/box/b"] = np.ascontiguousarray(np.tile(b, num_class))
  return out
