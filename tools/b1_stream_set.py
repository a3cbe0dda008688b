"""Tuning aid: b = 1 frames in flight on predict_stream's handles (replicas without side streams), K = 1 .. 4, device-resident."""
import os, sys, time, json
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from object_detection_tracking_amd import models
from object_detection_tracking_amd._lib import ODT_DTYPE_U8
from object_detection_tracking_amd.config import make_config
from object_detection_tracking_amd.weights import synthetic_frames, synthetic_weights
H, W = 1080, 1920
cfg = make_config(rpn_test_post_nms_topk=300, im_batch_size=1, max_size=W, short_edge_size=H)
m = models.get_model(cfg, 0, weights=synthetic_weights(cfg, 0), is_multi=False)
frs = [torch.from_numpy(synthetic_frames(1, H, W, seed=1234 + 77 * r)).cuda(0) for r in range(4)]
out = {}
extra_first = os.environ.get("WITH_MAIN_ENGINE") == "1"
if extra_first:
  e0 = m.engine(1, H, W); e0.forward_device_async(frs[0].data_ptr(), ODT_DTYPE_U8); e0.synchronize()
es = []
for K in (1, 2, 3, 4):
  while len(es) < K:
    e = m.engine(1, H, W, replica=len(es), stream_set=True); e.forward_device_async(frs[0].data_ptr(), ODT_DTYPE_U8); e.synchronize(); es.append(e)
  n = 120
  for k in range(8 + n):
    if k == 8:
      for e in es: e.synchronize()
      t0 = time.perf_counter()
    es[k % K].forward_device_async(frs[k % 4].data_ptr(), ODT_DTYPE_U8)
  for e in es: e.synchronize()
  out["frames_in_flight_%d" % K] = round(n / (time.perf_counter() - t0), 2)
print(json.dumps(out))
