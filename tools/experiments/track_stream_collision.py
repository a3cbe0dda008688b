#!/usr/bin/env python
"""Does the tracker's cosine stream end up behind the detector's forward on a shared hardware queue?  Runs bench.py's
detect_track leg (array entry points) after creating n extra streams in the process (n = 0..5: what earlier handles /
torch do to the runtime's stream -> hardware-queue assignment).  ODT_COSINE_STREAM_PRIORITY=0 is the plain stream."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch


def main():
  import bench
  from object_detection_tracking_amd import models
  from object_detection_tracking_amd.config import make_config
  from object_detection_tracking_amd.weights import synthetic_frames, synthetic_weights
  B, H, W = 8, 1080, 1920
  cfg = make_config(rpn_test_post_nms_topk=300, im_batch_size=B, max_size=max(H, W), short_edge_size=min(H, W))
  torch.cuda.set_device(0)
  m = models.get_model(cfg, 0, weights=synthetic_weights(cfg, seed=0), is_multi=True)
  eng = m.engine(B, H, W)
  frames = synthetic_frames(B, H, W, seed=1234)
  for _ in eng.forward_stream([frames] * 3):
    pass
  keep = []
  for n in range(6):
    r = bench.detect_track_leg(eng, frames, B, 0, nbatches=5, arrays=True)
    print("extra streams %d: %.1f FPS, host tracking %.2f ms/frame" % (n, r["detect_track_fps"], r["detect_track"]["host_tracking_ms_per_frame"]), flush=True)
    s = torch.cuda.Stream(); x = torch.zeros(16, device="cuda")
    with torch.cuda.stream(s):
      x.add_(1)
    s.synchronize()
    keep.append(s)
  m.close()


if __name__ == "__main__":
  main()
