"""Tuning aid: the detect + track leg of bench.py alone (both tracker front ends), for library A/Bs."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from object_detection_tracking_amd import models
from object_detection_tracking_amd.config import make_config
from object_detection_tracking_amd.weights import synthetic_frames, synthetic_weights

B, H, W = 8, 1080, 1920
cfg = make_config(rpn_test_post_nms_topk=300, im_batch_size=B, max_size=W, short_edge_size=H, conv_split_family=0)
m = models.get_model(cfg, 0, weights=synthetic_weights(cfg, 0), is_multi=True)
e = m.engine(B, H, W)
fr = synthetic_frames(B, H, W, seed=1234)
e.forward(fr)
out = {}
for arrays in (False, True, False, True):
  r = bench.detect_track_leg(e, fr, B, 0, nbatches=8, arrays=arrays)
  out.setdefault("arrays" if arrays else "objects", []).append((round(r["detect_track_fps"], 1), round(r["detect_track"]["host_tracking_ms_per_frame"], 3)))
print(json.dumps(out))
