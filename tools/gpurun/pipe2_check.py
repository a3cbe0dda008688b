"""Torch-free GPU check of the experimental two-stage split loop: the same conv through the default
split kernel and through conv_split2_kernel (ODT_CONV_SPLIT_PIPE=2) must agree at f32 rounding."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from object_detection_tracking_amd import _lib, ops
os.environ["ODT_CONV_SPLIT"] = "1"; os.environ["ODT_CONV_SPLIT_MINTILES"] = "1"
lib = _lib.get_lib()
rng = np.random.default_rng(0)
for (B, H, W, Cin, Cout, k, res) in ((2, 68, 120, 256, 256, 3, False), (1, 67, 119, 96, 512, 1, True)):
  x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
  w = (rng.standard_normal((k, k, Cin, Cout)) / np.sqrt(k * k * Cin)).astype(np.float32)
  b = rng.standard_normal(Cout).astype(np.float32)
  r = rng.standard_normal((B, H, W, Cout)).astype(np.float32) if res else None
  out = {}
  for pipe in ("0", "2"):
    os.environ["ODT_CONV_SPLIT_PIPE"] = pipe
    t0 = time.perf_counter()
    out[pipe] = ops.conv2d(x, w, b, 1, 1, k // 2, k // 2, (H, W), res=r, res_mode=1 if res else 0, relu=True, lib=lib)
    dt = time.perf_counter() - t0
  d = float(np.abs(out["0"] - out["2"]).max())
  print("conv %dx%d %d->%d k%d res=%s: max |pipe0 - pipe2| = %.3e (max |y| %.2f)  %s" %
        (H, W, Cin, Cout, k, res, d, float(np.abs(out["0"]).max()), "OK" if d < 1e-4 else "MISMATCH"))
