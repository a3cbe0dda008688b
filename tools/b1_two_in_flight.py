"""BASELINE config #2 (b = 1, Mask_RCNN_FPN) with K consecutive frames in flight on K handles (one stream each): frame t on handle
t mod K, device-resident frames.  Prints frames/s for K = 1, 2, 3, 4."""
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
w = synthetic_weights(cfg, 0)
frs = [torch.from_numpy(synthetic_frames(1, H, W, seed=1234 + 77 * r)).cuda(0) for r in range(4)]
out = {}
ms, es = [], []
for K in (1, 2, 3, 4):
  while len(es) < K:
    m = models.get_model(cfg, 0, weights=w, is_multi=False); ms.append(m)
    e = m.engine(1, H, W); e.forward_device_async(frs[0].data_ptr(), ODT_DTYPE_U8); e.synchronize(); es.append(e)
  n = 120
  for k in range(8 + n):
    if k == 8:
      for e in es: e.synchronize()
      t0 = time.perf_counter()
    es[k % K].forward_device_async(frs[k % 4].data_ptr(), ODT_DTYPE_U8)
  for e in es: e.synchronize()
  out["frames_in_flight_%d" % K] = round(n / (time.perf_counter() - t0), 2)
print(json.dumps(out))
