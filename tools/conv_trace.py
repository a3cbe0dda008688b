#!/usr/bin/env python
"""Tuning aid: run one conv shape through odt_op_conv2d with ODT_CONV_TRACE=1 (per-phase
wall-clock stamps inside the kernel).  python tools/conv_trace.py M_pixels_w N K ksize [res]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
os.environ["ODT_CONV_TRACE"] = "1"
from object_detection_tracking_amd import ops
shapes = {"conv3": (8, 68, 120, 256, 1024, 1, True), "conv2": (8, 68, 120, 256, 256, 3, False),
          "conv1": (8, 68, 120, 1024, 256, 1, False), "short0": (2, 272, 480, 64, 256, 1, False),
          "conv3b1": (1, 68, 120, 256, 1024, 1, True), "conv3b2": (2, 68, 120, 256, 1024, 1, True),
          "conv3nores": (8, 68, 120, 256, 1024, 1, False),
          "r5conv3": (8, 34, 60, 512, 2048, 1, True), "r5conv1": (8, 34, 60, 2048, 512, 1, False), "r5conv2": (8, 34, 60, 512, 512, 3, False),
          "lat2": (8, 272, 480, 256, 256, 1, True), "r3conv1": (8, 136, 240, 512, 128, 1, False),
          "e1": (1, 32, 128, 2304, 1024, 1, False), "e2": (1, 64, 128, 2304, 1024, 1, False),
          "e3": (1, 96, 128, 2304, 1024, 1, False), "e4": (1, 128, 128, 2304, 1024, 1, False)}
for name in sys.argv[1:]:
  B, H, W, Cin, Cout, k, res = shapes[name]
  rng = np.random.default_rng(0)
  x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
  w = (rng.standard_normal((k, k, Cin, Cout)) * 0.05).astype(np.float32)
  r = rng.standard_normal((B, H, W, Cout)).astype(np.float32) if res else None
  print("==", name, (B * H * W, Cout, k * k * Cin), flush=True)
  ops.conv2d(x, w, np.zeros(Cout, np.float32), pad_t=k // 2, pad_l=k // 2, out_hw=(H, W), res=r,
             res_mode=1 if res else 0, relu=True)
