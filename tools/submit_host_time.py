"""Tuning aid: host time of one odt_submit_ex (pageable 8 x 1080p uint8 frames: staging copy + enqueue) while the GPU is busy."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from object_detection_tracking_amd import models
from object_detection_tracking_amd.config import make_config
from object_detection_tracking_amd.weights import synthetic_frames, synthetic_weights
B, H, W = 8, 1080, 1920
cfg = make_config(rpn_test_post_nms_topk=300, im_batch_size=B, max_size=W, short_edge_size=H, conv_split_family=0)
m = models.get_model(cfg, 0, weights=synthetic_weights(cfg, 0), is_multi=True)
e = m.engine(B, H, W)
fr = synthetic_frames(B, H, W, seed=1)
e.forward(fr)
ts = []
pend = [e.submit(fr, want_feats=False, want_pooled=True)]
for k in range(12):
  t0 = time.perf_counter(); pend.append(e.submit(fr, want_feats=False, want_pooled=True)); ts.append(time.perf_counter() - t0)
  e.collect(pend.pop(0))
e.collect(pend.pop(0))
print(json.dumps({"submit_host_ms_median": round(1e3 * sorted(ts)[len(ts) // 2], 3), "min": round(1e3 * min(ts), 3), "max": round(1e3 * max(ts), 3)}))
