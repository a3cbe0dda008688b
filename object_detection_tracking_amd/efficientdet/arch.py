"""Architecture arithmetic of the EfficientNet backbones the reference's EfficientDet uses
(reference efficientdet/backbone/efficientnet_builder.py:37-53 model table, :162-168 block strings,
efficientnet_model.py:137-159 round_filters / round_repeats, :520-577 block expansion,
efficientdet_wrapper.py:511-587 D0..D7 -> backbone), variable names as the TF graph creates them
(Keras layers named 'conv2d' / 'depthwise_conv2d' / 'tpu_batch_normalization', uniquified in call
order inside each ``blocks_N`` scope), and a seeded synthetic weight generator with those names."""
from __future__ import annotations

import math

import numpy as np

# (width_coefficient, depth_coefficient)        efficientnet_builder.py:40-51
_PARAMS = {"efficientnet-b0": (1.0, 1.0), "efficientnet-b1": (1.0, 1.1), "efficientnet-b2": (1.1, 1.2),
           "efficientnet-b3": (1.2, 1.4), "efficientnet-b4": (1.4, 1.8), "efficientnet-b5": (1.6, 2.2),
           "efficientnet-b6": (1.8, 2.6), "efficientnet-b7": (2.0, 3.1)}
# (repeat, kernel, stride, expand, in, out)     efficientnet_builder.py:162-168 (se 0.25 everywhere)
_BLOCKS = [(1, 3, 1, 1, 32, 16), (2, 3, 2, 6, 16, 24), (2, 5, 2, 6, 24, 40), (3, 3, 2, 6, 40, 80),
           (3, 5, 1, 6, 80, 112), (4, 5, 2, 6, 112, 192), (1, 3, 1, 6, 192, 320)]
# efficientdet_wrapper.py:511-587
EFFICIENTDET = {
    "efficientdet-d0": dict(backbone="efficientnet-b0", image_size=512, fpn_num_filters=64, fpn_cell_repeats=3, box_class_repeats=3),
    "efficientdet-d1": dict(backbone="efficientnet-b1", image_size=640, fpn_num_filters=88, fpn_cell_repeats=4, box_class_repeats=3),
    "efficientdet-d2": dict(backbone="efficientnet-b2", image_size=768, fpn_num_filters=112, fpn_cell_repeats=5, box_class_repeats=3),
    "efficientdet-d3": dict(backbone="efficientnet-b3", image_size=896, fpn_num_filters=160, fpn_cell_repeats=6, box_class_repeats=4),
    "efficientdet-d4": dict(backbone="efficientnet-b4", image_size=1024, fpn_num_filters=224, fpn_cell_repeats=7, box_class_repeats=4),
    "efficientdet-d5": dict(backbone="efficientnet-b5", image_size=1280, fpn_num_filters=288, fpn_cell_repeats=7, box_class_repeats=4),
    "efficientdet-d6": dict(backbone="efficientnet-b6", image_size=1280, fpn_num_filters=384, fpn_cell_repeats=8, box_class_repeats=5),
    "efficientdet-d7": dict(backbone="efficientnet-b6", image_size=1536, fpn_num_filters=384, fpn_cell_repeats=8, box_class_repeats=5),
}


def efficientnet_params(name):
  return _PARAMS[name]


def round_filters(filters, width, divisor=8):
  """efficientnet_model.py:137-151."""
  filters *= width
  new = max(divisor, int(filters + divisor / 2) // divisor * divisor)
  if new < 0.9 * filters:
    new += divisor
  return int(new)


def round_repeats(repeats, depth):
  """efficientnet_model.py:154-159."""
  return int(math.ceil(depth * repeats))


def backbone_spec(name):
  """-> dict(stem=filters, blocks=[dict(idx, kernel, stride, expand, cin, cout, se, reduction)]).
  ``reduction`` is the level (1..5) whose feature map this block's output is
  (efficientnet_model.py:617-646: the last block before a stride-2 block, and the last block)."""
  width, depth = _PARAMS[name]
  blocks = []
  for (r, k, s, e, i, o) in _BLOCKS:
    cin, cout = round_filters(i, width), round_filters(o, width)
    for rep in range(round_repeats(r, depth)):
      blocks.append(dict(kernel=k, stride=s if rep == 0 else 1, expand=e, cin=cin if rep == 0 else cout,
                         cout=cout, se=max(1, int((cin if rep == 0 else cout) * 0.25)), reduction=0))
  red = 0
  for idx, b in enumerate(blocks):
    b["idx"] = idx
    if idx == len(blocks) - 1 or blocks[idx + 1]["stride"] > 1:
      red += 1
      b["reduction"] = red
  return dict(name=name, stem=round_filters(32, width), blocks=blocks)


def backbone_variable_shapes(name):
  """{variable name: shape} in the TF layouts (conv HWIO, depthwise [k,k,C,1])."""
  sp = backbone_spec(name)
  pre = name + "/"
  v = {}
  def bn(scope, c):
    for s in ("gamma", "beta", "moving_mean", "moving_variance"):
      v[scope + "/" + s] = (c,)
  v[pre + "stem/conv2d/kernel"] = (3, 3, 3, sp["stem"])
  bn(pre + "stem/tpu_batch_normalization", sp["stem"])
  for b in sp["blocks"]:
    p = pre + "blocks_%d/" % b["idx"]
    mid = b["cin"] * b["expand"]
    nconv = nbn = 0
    def conv_name():
      nonlocal nconv
      n = "conv2d" if nconv == 0 else "conv2d_%d" % nconv
      nconv += 1
      return n
    def bn_name():
      nonlocal nbn
      n = "tpu_batch_normalization" if nbn == 0 else "tpu_batch_normalization_%d" % nbn
      nbn += 1
      return n
    if b["expand"] != 1:
      v[p + conv_name() + "/kernel"] = (1, 1, b["cin"], mid)
      bn(p + bn_name(), mid)
    v[p + "depthwise_conv2d/depthwise_kernel"] = (b["kernel"], b["kernel"], mid, 1)
    bn(p + bn_name(), mid)
    v[p + "se/conv2d/kernel"] = (1, 1, mid, b["se"]); v[p + "se/conv2d/bias"] = (b["se"],)
    v[p + "se/conv2d_1/kernel"] = (1, 1, b["se"], mid); v[p + "se/conv2d_1/bias"] = (mid,)
    v[p + conv_name() + "/kernel"] = (1, 1, mid, b["cout"])
    bn(p + bn_name(), b["cout"])
  return v


def synthetic_backbone_weights(name, seed=0):
  """Seeded random-init weights with the TF names / layouts (He-style convs, BN close to identity,
  the last BN gamma of every block damped so that the residual stream stays O(1))."""
  rng = np.random.default_rng(seed)
  w = {}
  for k, shp in backbone_variable_shapes(name).items():
    base = k.rsplit("/", 1)[1]
    if base in ("kernel", "depthwise_kernel"):
      fan_in = shp[0] * shp[1] * (shp[2] if base == "kernel" else 1)
      w[k] = (rng.standard_normal(shp) * np.sqrt(2.0 / fan_in)).astype(np.float32)
    elif base == "bias":
      w[k] = (rng.standard_normal(shp) * 0.1).astype(np.float32)
    elif base == "gamma":
      w[k] = rng.uniform(0.8, 1.2, shp).astype(np.float32)
    elif base == "beta":
      w[k] = (rng.standard_normal(shp) * 0.05).astype(np.float32)
    elif base == "moving_mean":
      w[k] = (rng.standard_normal(shp) * 0.05).astype(np.float32)
    elif base == "moving_variance":
      w[k] = rng.uniform(0.8, 1.2, shp).astype(np.float32)
  sp = backbone_spec(name)
  for b in sp["blocks"]:                       # damp the projection BN of residual blocks
    nbn = 2 if b["expand"] != 1 else 1
    g = "%s/blocks_%d/tpu_batch_normalization%s/gamma" % (name, b["idx"], "_%d" % nbn if nbn else "")
    w[g] = (w[g] * 0.4).astype(np.float32)
  return w


# ---- EfficientDet feature network + heads -------------------------------------------------------
# bifpn node table (efficientdet_arch.py:508-522): (feat_level, inputs_offsets)
BIFPN_NODES = [(6, [3, 4]), (5, [2, 5]), (4, [1, 6]), (3, [0, 7]), (4, [1, 7, 8]), (5, [2, 6, 9]),
               (6, [3, 5, 10]), (7, [4, 11])]
NUM_ANCHORS = 9              # num_scales 3 x aspect_ratios 3 (efficientdet_wrapper.py:179-181)
NUM_CLASSES = 90


def feat_sizes(height, width, max_level=7):
  """utils.get_feat_sizes (efficientdet/utils.py:467-484)."""
  out = [(height, width)]
  for _ in range(max_level):
    h, w = out[-1]
    out.append(((h - 1) // 2 + 1, (w - 1) // 2 + 1))
  return out


def det_config(model_name):
  c = dict(EFFICIENTDET[model_name])
  c["name"] = model_name
  c["weight_method"] = "sum" if model_name == "efficientdet-d7" else "fastattn"   # wrapper :585, arch :582-591
  c["anchor_scale"] = 5.0 if model_name == "efficientdet-d7" else 4.0
  return c


def det_variable_shapes(model_name, num_classes=NUM_CLASSES):
  """Variables of the feature network and the class / box nets (TF names and layouts), on top of
  backbone_variable_shapes(backbone)."""
  c = det_config(model_name)
  F_ = c["fpn_num_filters"]
  sp = backbone_spec(c["backbone"])
  red = {b["reduction"]: b["cout"] for b in sp["blocks"] if b["reduction"]}
  v = {}
  def bn(scope, ch):
    for s in ("gamma", "beta", "moving_mean", "moving_variance"):
      v[scope + "/" + s] = (ch,)
  def resample(scope, cin):
    if cin != F_:
      v[scope + "/conv2d/kernel"] = (1, 1, cin, F_); v[scope + "/conv2d/bias"] = (F_,)
      bn(scope + "/bn", F_)
  resample("resample_p6", red[5])                       # P6 from P5 (P7 from P6: channels equal)
  chans = [red[3], red[4], red[5], F_, F_]              # channels of feats[0..4] entering cell 0
  for rep in range(c["fpn_cell_repeats"]):
    ch = list(chans) if rep == 0 else [F_] * 5
    for i, (lvl, offs) in enumerate(BIFPN_NODES):
      p = "fpn_cells/cell_%d/fnode%d/" % (rep, i)
      for idx, off in enumerate(offs):
        resample(p + "resample_%d_%d_%d" % (idx, off, len(ch)), ch[off])
      if c["weight_method"] == "fastattn":
        for idx in range(len(offs)):
          v[p + ("WSM" if idx == 0 else "WSM_%d" % idx)] = ()
      q = p + "op_after_combine%d/" % len(ch)
      v[q + "conv/depthwise_kernel"] = (3, 3, F_, 1); v[q + "conv/pointwise_kernel"] = (1, 1, F_, F_)
      v[q + "conv/bias"] = (F_,)
      bn(q + "bn", F_)
      ch.append(F_)
  for net, nout in (("class", num_classes * NUM_ANCHORS), ("box", 4 * NUM_ANCHORS)):
    for i in range(c["box_class_repeats"]):
      p = "%s_net/%s-%d/" % (net, net, i)
      v[p + "depthwise_kernel"] = (3, 3, F_, 1); v[p + "pointwise_kernel"] = (1, 1, F_, F_); v[p + "bias"] = (F_,)
      for lvl in range(3, 8):
        bn("%s_net/%s-%d-bn-%d" % (net, net, i, lvl), F_)
    p = "%s_net/%s-predict/" % (net, net)
    v[p + "depthwise_kernel"] = (3, 3, F_, 1); v[p + "pointwise_kernel"] = (1, 1, F_, nout); v[p + "bias"] = (nout,)
  return v


def bench_gain(model_name):
  """Kernel gain that keeps random-init activations O(1) through the model's depth (synthetic benchmarks only)."""
  return 0.7 if det_config(model_name)["fpn_cell_repeats"] >= 8 else 1.0


def synthetic_det_weights(model_name, seed=0, num_classes=NUM_CLASSES, gain=1.0):
  """Backbone + feature network + heads, seeded, TF names.  ``gain`` scales the standard deviation of the feature
  network's / heads' conv kernels: the deep models that fuse by plain sums (D7: eight cells of 'sum' nodes) need < 1 for
  the random-init activations to stay finite (``bench_gain``)."""
  c = det_config(model_name)
  w = synthetic_backbone_weights(c["backbone"], seed)
  rng = np.random.default_rng(seed + 1)
  for k, shp in det_variable_shapes(model_name, num_classes).items():
    base = k.rsplit("/", 1)[1]
    if base.startswith("WSM"):
      w[k] = np.asarray(rng.uniform(0.5, 1.5), np.float32)
    elif base in ("kernel", "pointwise_kernel"):
      w[k] = (rng.standard_normal(shp) * (gain * np.sqrt(1.5 / shp[2]))).astype(np.float32)
    elif base == "depthwise_kernel":
      w[k] = (rng.standard_normal(shp) * np.sqrt(1.5 / 9)).astype(np.float32)
    elif base == "bias":
      w[k] = (rng.standard_normal(shp) * 0.05).astype(np.float32)
    elif base == "gamma":
      w[k] = rng.uniform(0.8, 1.2, shp).astype(np.float32)
    elif base in ("beta", "moving_mean"):
      w[k] = (rng.standard_normal(shp) * 0.05).astype(np.float32)
    elif base == "moving_variance":
      w[k] = rng.uniform(0.8, 1.2, shp).astype(np.float32)
  w["class_net/class-predict/bias"] = (w["class_net/class-predict/bias"] - 3.0).astype(np.float32)  # sparse positives
  return w


def algorithmic_traffic_and_flops(model_name, height, width, num_classes=NUM_CLASSES, fused=False):
  """(bytes, flops) of one forward with every operator reading its inputs and writing its output
  exactly once in fp32 (unpadded channel counts, weights read once) -- the unfused HBM floor the
  kernels are measured against -- and 2*MAC flops.  Walks the same graph as the plan builder.
  fused=True: the byte count of the best fusion the graph allows instead -- an MBConv block as two passes
  (expand -> depthwise -> squeeze sums | gate * project + skip: the squeeze-excite mean is a global reduction, so
  the depthwise output has to reach memory once), a BiFPN node (resample + fusion + separable conv) and a class /
  box net layer (depthwise + pointwise) as one pass each; same flops."""
  if fused:
    return _fused_traffic(model_name, height, width, num_classes), algorithmic_traffic_and_flops(model_name, height, width, num_classes)[1]
  c = det_config(model_name)
  sp = backbone_spec(c["backbone"])
  F_ = c["fpn_num_filters"]
  by = fl = 0
  def t(h, w, ch):
    return 4 * h * w * ch
  h, w = -(-height // 2), -(-width // 2)
  by += height * width * 3 + t(h, w, sp["stem"]); fl += 2 * h * w * 27 * sp["stem"]
  red = {}
  for b in sp["blocks"]:
    mid = b["cin"] * b["expand"]
    if b["expand"] != 1:
      by += t(h, w, b["cin"]) + t(h, w, mid) + 4 * b["cin"] * mid; fl += 2 * h * w * b["cin"] * mid
    ho, wo = (-(-h // 2), -(-w // 2)) if b["stride"] == 2 else (h, w)
    k2 = b["kernel"] ** 2
    by += t(h, w, mid) + t(ho, wo, mid); fl += 2 * ho * wo * mid * k2            # depthwise
    by += t(ho, wo, mid)                                                          # SE mean
    by += 2 * t(ho, wo, mid) + 8 * mid * b["se"]; fl += 4 * mid * b["se"] + ho * wo * mid   # gate FCs + scale
    by += t(ho, wo, mid) + t(ho, wo, b["cout"]) + 4 * mid * b["cout"]; fl += 2 * ho * wo * mid * b["cout"]
    if b["stride"] == 1 and b["cin"] == b["cout"]:
      by += t(ho, wo, b["cout"])
    h, w = ho, wo
    if b["reduction"]:
      red[b["reduction"]] = (h, w, b["cout"])
  sizes = feat_sizes(height, width)
  def sep(hh, ww, cin, cout):
    return (2 * t(hh, ww, cin) + t(hh, ww, cin) + t(hh, ww, cout) + 4 * cin * (9 + cout),
            2 * hh * ww * cin * (9 + cout))
  chans = [red[3][2], red[4][2], red[5][2], F_, F_]
  by += t(*red[5]) + 2 * t(sizes[6][0], sizes[6][1], F_) + t(sizes[7][0], sizes[7][1], F_)     # P6, P7
  fl += 2 * red[5][0] * red[5][1] * red[5][2] * F_
  for rep in range(c["fpn_cell_repeats"]):
    ch = list(chans) if rep == 0 else [F_] * 5
    lv = [3, 4, 5, 6, 7]
    for lvl, offs in BIFPN_NODES:
      hh, ww = sizes[lvl]
      for off in offs:
        hs, ws = sizes[lv[off]]
        if ch[off] != F_:
          by += t(hs, ws, ch[off]) + t(hs, ws, F_); fl += 2 * hs * ws * ch[off] * F_
        by += t(hs, ws, F_)
      by += t(hh, ww, F_)
      b2, f2 = sep(hh, ww, F_, F_); by += b2; fl += f2
      ch.append(F_); lv.append(lvl)
  for lvl in range(3, 8):
    hh, ww = sizes[lvl]
    for nout in (num_classes * NUM_ANCHORS, 4 * NUM_ANCHORS):
      for _ in range(c["box_class_repeats"]):
        b2, f2 = sep(hh, ww, F_, F_); by += b2; fl += f2
      b2, f2 = sep(hh, ww, F_, nout); by += b2; fl += f2
  nlog = sum(sizes[l][0] * sizes[l][1] for l in range(3, 8)) * NUM_ANCHORS * num_classes
  by += 4 * nlog * 10                                   # pack + 8 radix passes + compaction over the logits
  return by, fl


def _fused_traffic(model_name, height, width, num_classes=NUM_CLASSES):
  c = det_config(model_name)
  sp = backbone_spec(c["backbone"])
  F_ = c["fpn_num_filters"]
  by = 0
  def t(h, w, ch):
    return 4 * h * w * ch
  h, w = -(-height // 2), -(-width // 2)
  by += height * width * 3 + t(h, w, sp["stem"])
  red = {}
  for b in sp["blocks"]:
    mid = b["cin"] * b["expand"]
    ho, wo = (-(-h // 2), -(-w // 2)) if b["stride"] == 2 else (h, w)
    # pass 1: block input -> (expand, depthwise) -> depthwise output + squeeze sums; pass 2: depthwise output (+ skip) -> out
    by += t(h, w, b["cin"]) + t(ho, wo, mid) + 4 * (b["cin"] * mid if b["expand"] != 1 else 0) + 4 * mid * b["kernel"] ** 2
    by += t(ho, wo, mid) + t(ho, wo, b["cout"]) + 4 * mid * b["cout"] + 8 * mid * b["se"]
    if b["stride"] == 1 and b["cin"] == b["cout"]:
      by += t(ho, wo, b["cout"])
    h, w = ho, wo
    if b["reduction"]:
      red[b["reduction"]] = (h, w, b["cout"])
  sizes = feat_sizes(height, width)
  chans = [red[3][2], red[4][2], red[5][2], F_, F_]
  by += t(*red[5]) + t(sizes[6][0], sizes[6][1], F_) + t(sizes[7][0], sizes[7][1], F_)      # P6 (1x1 + pool), P7 (pool)
  for rep in range(c["fpn_cell_repeats"]):
    ch = list(chans) if rep == 0 else [F_] * 5
    lv = [3, 4, 5, 6, 7]
    for lvl, offs in BIFPN_NODES:
      hh, ww = sizes[lvl]
      for off in offs:
        hs, ws = sizes[lv[off]]
        by += t(hs, ws, ch[off]) + (4 * ch[off] * F_ if ch[off] != F_ else 0)
      by += t(hh, ww, F_) + 4 * F_ * (9 + F_)
      ch.append(F_); lv.append(lvl)
  for lvl in range(3, 8):
    hh, ww = sizes[lvl]
    for nout in (num_classes * NUM_ANCHORS, 4 * NUM_ANCHORS):
      for _ in range(c["box_class_repeats"]):
        by += 2 * t(hh, ww, F_) + 4 * F_ * (9 + F_)
      by += t(hh, ww, F_) + t(hh, ww, nout) + 4 * F_ * (9 + nout)
  nlog = sum(sizes[l][0] * sizes[l][1] for l in range(3, 8)) * NUM_ANCHORS * num_classes
  by += 4 * nlog * 2                                    # one read of the logits for the select, keys of the survivors
  return by
