#!/usr/bin/env python
"""How long does a tiny operation on another stream take while the detector's forward occupies the GPU?  (kernel,
H2D from pinned memory, D2H to pinned memory; each followed by a stream synchronize.)  GPU only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch


def main():
  from object_detection_tracking_amd import models
  from object_detection_tracking_amd._lib import ODT_DTYPE_U8
  from object_detection_tracking_amd.config import make_config
  from object_detection_tracking_amd.weights import synthetic_frames, synthetic_weights
  B, H, W = 8, 1080, 1920
  cfg = make_config(rpn_test_post_nms_topk=300, im_batch_size=B, max_size=max(H, W), short_edge_size=min(H, W))
  m = models.get_model(cfg, 0, weights=synthetic_weights(cfg, seed=0), is_multi=True)
  eng = m.engine(B, H, W)
  dev = torch.from_numpy(synthetic_frames(B, H, W, seed=1)).cuda(0)
  eng.forward_device_async(dev.data_ptr(), ODT_DTYPE_U8); eng.synchronize()
  x = torch.zeros(4096, device="cuda"); hp = torch.zeros(100000, dtype=torch.float32).pin_memory(); hd = torch.zeros(100000, device="cuda")
  for prio in (0, -1):
    s = torch.cuda.Stream(priority=prio)
    for busy in (False, True):
      res = {"kernel": [], "h2d": [], "d2h": []}
      eng.synchronize(); torch.cuda.synchronize()
      if busy:
        for _ in range(6):
          eng.forward_device_async(dev.data_ptr(), ODT_DTYPE_U8)
      with torch.cuda.stream(s):
        for i in range(12):
          t = time.perf_counter(); x.add_(1.0); s.synchronize(); res["kernel"].append(time.perf_counter() - t)
          t = time.perf_counter(); hd.copy_(hp, non_blocking=True); s.synchronize(); res["h2d"].append(time.perf_counter() - t)
          t = time.perf_counter(); hp.copy_(hd, non_blocking=True); s.synchronize(); res["d2h"].append(time.perf_counter() - t)
      t = time.perf_counter(); eng.synchronize(); rest = time.perf_counter() - t
      print("priority %d, detector %s: " % (prio, "busy" if busy else "idle") +
            "  ".join("%s med %.3f max %.3f ms" % (k, 1e3 * float(np.median(v)), 1e3 * max(v)) for k, v in res.items()) +
            "   (forward still running for %.1f ms afterwards)" % (1e3 * rest))
  m.close()


if __name__ == "__main__":
  main()
