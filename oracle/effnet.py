"""CPU ORACLE (test infrastructure only, see oracle/__init__.py) -- EfficientNet backbone of the
reference's EfficientDet path: efficientdet/backbone/efficientnet_model.py:162-330 (MBConvBlock:
expand 1x1 + BN + swish, depthwise kxk + BN + swish, squeeze-excite, project 1x1 + BN, identity
skip), :600-650 (stem 3x3 s2 + BN + swish, block loop, reduction_1..5 endpoints); inference BN with
epsilon 1e-3 (efficientnet_builder.py:177); TF 'SAME' padding (pad_total = max((ceil(n/s)-1)*s + k -
n, 0), the extra pixel at the bottom / right); swish = x * sigmoid(x) (tf.nn.swish).
Pinning: TensorFlow is not installable and no checkpoint ships, so values are unpinned; the
ARCHITECTURE arithmetic (filter rounding, repeats, SE widths, variable shapes) is pinned by the
published EfficientNet parameter counts (tests/test_efficientnet.py).
"""
import numpy as np
import torch
import torch.nn.functional as TF

from object_detection_tracking_amd.efficientdet.arch import backbone_spec

F = np.float32
BN_EPS = 1e-3


def _t(a):
  return torch.from_numpy(np.ascontiguousarray(a, dtype=F))


def same_pad(x, k, s):
  """TF SAME padding for an NCHW tensor."""
  h, w = x.shape[2], x.shape[3]
  ph = max((-(-h // s) - 1) * s + k - h, 0); pw = max((-(-w // s) - 1) * s + k - w, 0)
  return TF.pad(x, (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2))


def bn(x, w, scope):
  g, b, m, v = (_t(w[scope + "/" + s]) for s in ("gamma", "beta", "moving_mean", "moving_variance"))
  inv = g / torch.sqrt(v + BN_EPS)
  return x * inv.view(1, -1, 1, 1) + (b - m * inv).view(1, -1, 1, 1)


def swish(x):
  return x * torch.sigmoid(x)


def conv(x, w, name, k=1, s=1):
  W = _t(w[name + "/kernel"]).permute(3, 2, 0, 1).contiguous()
  b = _t(w[name + "/bias"]) if name + "/bias" in w else None
  return TF.conv2d(same_pad(x, k, s) if k > 1 else x, W, b, stride=s)


def depthwise(x, w, name, k, s):
  W = _t(w[name + "/depthwise_kernel"]).permute(2, 3, 0, 1).contiguous()      # [C,1,k,k]
  return TF.conv2d(same_pad(x, k, s), W, None, stride=s, groups=x.shape[1])


def preprocess(frames_bgr_u8):
  """efficientdet_wrapper.py:45-60 + dataloader normalize_image: BGR -> RGB, [0,1], (x - mean) / std
  with the ImageNet RGB constants; NCHW float32."""
  x = np.asarray(frames_bgr_u8)[..., ::-1].astype(F) * F(1.0 / 255)      # tf.image.convert_image_dtype
  x = (x - np.array([0.485, 0.456, 0.406], F)) / np.array([0.229, 0.224, 0.225], F)
  return torch.from_numpy(np.ascontiguousarray(x.transpose(0, 3, 1, 2)))


def backbone_forward(name, weights, x, taps=None):
  """x: NCHW float32 (already normalised).  Returns {level: NCHW numpy} for reduction_1..5."""
  sp = backbone_spec(name)
  pre = name + "/"
  out = {}
  with torch.no_grad():
    x = swish(bn(conv(x, weights, pre + "stem/conv2d", 3, 2), weights, pre + "stem/tpu_batch_normalization"))
    if taps is not None:
      taps["stem"] = x.numpy()
    for b in sp["blocks"]:
      p = pre + "blocks_%d/" % b["idx"]
      inp = x
      nconv = nbn = 0
      def cname():
        nonlocal nconv
        n = "conv2d" if nconv == 0 else "conv2d_%d" % nconv
        nconv += 1
        return p + n
      def bname():
        nonlocal nbn
        n = "tpu_batch_normalization" if nbn == 0 else "tpu_batch_normalization_%d" % nbn
        nbn += 1
        return p + n
      if b["expand"] != 1:
        x = swish(bn(conv(x, weights, cname()), weights, bname()))
      x = swish(bn(depthwise(x, weights, p + "depthwise_conv2d", b["kernel"], b["stride"]), weights, bname()))
      se = x.mean(dim=(2, 3), keepdim=True)
      se = conv(swish(conv(se, weights, p + "se/conv2d")), weights, p + "se/conv2d_1")
      x = torch.sigmoid(se) * x
      x = bn(conv(x, weights, cname()), weights, bname())
      if b["stride"] == 1 and b["cin"] == b["cout"]:
        x = x + inp
      if taps is not None:
        taps["block_%d" % b["idx"]] = x.numpy()
      if b["reduction"]:
        out[b["reduction"]] = x.numpy()
  return out
