#!/usr/bin/env python
"""Per-layer HIP-event timing of the conv launches of one forward (GPU only).
  python tools/profile_layers.py [--batch 8] [--steps 3] [--tile N]
Groups identical GEMM shapes and prints TFLOP/s and its fraction of the running kernel's own matrix-pipe ceiling
(2500 / 3 TF of f32 work on the fp16x2 kernels, 2500 / 6 on the bf16x3 ones, 157.3 on the exact-f32 kernel)."""
import argparse, os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--batch", type=int, default=8)
  ap.add_argument("--steps", type=int, default=3)
  ap.add_argument("--height", type=int, default=1080)
  ap.add_argument("--width", type=int, default=1920)
  ap.add_argument("--effdet", default="", help="profile an EfficientDet model (e.g. efficientdet-d7, batch 1, native size) instead")
  ap.add_argument("--top", type=int, default=0)
  a = ap.parse_args()
  if a.effdet:
    return effdet(a)
  from object_detection_tracking_amd import models
  from object_detection_tracking_amd.config import make_config
  from object_detection_tracking_amd.weights import synthetic_frames, synthetic_weights
  cfg = make_config(rpn_test_post_nms_topk=300, im_batch_size=a.batch)
  m = models.get_model(cfg, 0, weights=synthetic_weights(cfg, 0), is_multi=True)
  e = m.engine(a.batch, a.height, a.width)
  fr = synthetic_frames(a.batch, a.height, a.width)
  e.forward(fr)
  e.profile(True)
  for _ in range(a.steps):
    e.forward(fr)
  rows = e.profile_layers()
  tot = e.profile_read()
  groups = collections.OrderedDict()
  for name, fl, ms, mnk in rows:
    kind = name.split("/")[-1].split("@")[0] if not name.startswith("fpn") and not name.startswith("rpn") and not name.startswith("fastrcnn") else name
    key = (mnk, kind if mnk[2] in (147, 224) else "")
    if "[fused into" in name:
      continue                       # no launch of its own: counted with the kernel that evaluates it
    g = groups.setdefault((mnk, "+conv3" in name, "+head" in name), [0, 0.0, 0.0, name])
    g[0] += 1; g[1] += fl; g[2] += ms / a.steps
  print("%-44s %5s %9s %7s %7s %9s %8s %7s" % ("first layer of shape", "n", "M", "N", "K", "ms/step", "TFLOP/s", "frac"))
  for (mnk, _, _), (n, fl, ms, name) in sorted(groups.items(), key=lambda kv: -kv[1][2]):
    tf = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0
    # frac: against the ceiling of the kernel that runs the layer -- dense 16-bit MFMA peak / 3 products (fp16x2), / 6 (bf16x3),
    # the f32 MFMA peak (exact-f32 kernel)
    ceil = 2500.0 / 3 if name.endswith("[fp16x2]") else (2500.0 / 6 if name.endswith("[bf16x3]") else 157.3)
    print("%-44s %5d %9d %7d %7d %9.3f %8.1f %7.3f" % (name[:44], n, mnk[0], mnk[1], mnk[2], ms, tf, tf / ceil))
  cms = tot["conv_ms"] / a.steps
  print("conv total %.2f ms/step, %.1f TFLOP/s (%.3f of peak); step %.2f ms" %
        (cms, tot["conv_flops"] / a.steps / (cms * 1e-3) / 1e12, tot["conv_flops"] / a.steps / (cms * 1e-3) / 1e12 / 157.3, tot["total_ms"] / a.steps))
  m.close()

def effdet(a):
  from object_detection_tracking_amd import models
  from object_detection_tracking_amd.config import make_config
  from object_detection_tracking_amd.efficientdet import arch
  from object_detection_tracking_amd.weights import synthetic_frames
  S = arch.det_config(a.effdet)["image_size"]
  cfg = make_config(is_efficientdet=True, efficientdet_modelname=a.effdet, efficientdet_max_detection_topk=5000, short_edge_size=S, max_size=S)
  cfg.max_size = S
  m = models.get_model(cfg, 0, weights=arch.synthetic_det_weights(a.effdet, 0, gain=arch.bench_gain(a.effdet)))
  fr = synthetic_frames(1, S, S)[0]
  e = m.engine((S, S))
  m.predict(fr)
  E = models._Engine                     # (the profiling entry points only need .lib and .h)
  E.profile(e, True)
  for _ in range(a.steps):
    m.predict(fr)
  rows = E.profile_layers(e)
  tot = E.profile_read(e)
  groups = collections.OrderedDict()
  for name, fl, ms, mnk in rows:
    if "[fused into" in name:
      continue                       # no launch of its own: counted with the kernel that evaluates it
    g = groups.setdefault((mnk, "+conv3" in name, "+head" in name), [0, 0.0, 0.0, name])
    g[0] += 1; g[1] += fl; g[2] += ms / a.steps
  print("%-60s %5s %9s %7s %7s %9s %8s %9s" % ("first layer of shape", "n", "M", "N", "K", "ms/step", "TFLOP/s", "us/launch"))
  srt = sorted(groups.items(), key=lambda kv: -kv[1][2])
  for (mnk, _c3, _hd), (n, fl, ms, name) in (srt[:a.top] if a.top else srt):
    tf = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0
    print("%-60s %5d %9d %7d %7d %9.3f %8.1f %9.1f" % (name[:60], n, mnk[0], mnk[1], mnk[2], ms, tf, ms * 1e3 / n))
  cms = tot["conv_ms"] / a.steps
  print("conv total %.2f ms/step, %.1f TFLOP/s; step %.2f ms" % (cms, tot["conv_flops"] / a.steps / (cms * 1e-3) / 1e12, tot["total_ms"] / a.steps))
  m.close()


if __name__ == "__main__":
  main()
