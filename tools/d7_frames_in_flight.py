"""EfficientDet-D7 @1536 (config #5) with K consecutive frames in flight on K handles (one stream each), device-resident frames."""
import os, sys, time, json
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from object_detection_tracking_amd import models
from object_detection_tracking_amd._lib import ODT_DTYPE_U8
from object_detection_tracking_amd.config import make_config
from object_detection_tracking_amd.efficientdet import arch
from object_detection_tracking_amd.weights import synthetic_frames
model = "efficientdet-d7"
S = arch.det_config(model)["image_size"]
cfg = make_config(is_efficientdet=True, efficientdet_modelname=model, efficientdet_max_detection_topk=5000, short_edge_size=S, max_size=S)
cfg.max_size = S
w = arch.synthetic_det_weights(model, 0, gain=arch.bench_gain(model))
fr = synthetic_frames(1, S, S)[0]
dev = torch.from_numpy(fr[None].copy()).cuda(0)
ms, es, out = [], [], {}
for K in (1, 3, 4, 5, 6, 8):
  while len(es) < K:
    m = models.get_model(cfg, 0, weights=w); ms.append(m)
    e = m.engine((S, S)); es.append(e)
    e.lib.check(e.lib.dll.odt_forward_async(e.h, dev.data_ptr(), ODT_DTYPE_U8, 1, None)); e.synchronize()
  n = 60
  for k in range(6 + n):
    if k == 6:
      for e in es: e.synchronize()
      t0 = time.perf_counter()
    e = es[k % K]
    e.lib.check(e.lib.dll.odt_forward_async(e.h, dev.data_ptr(), ODT_DTYPE_U8, 1, None))
  for e in es: e.synchronize()
  out["frames_in_flight_%d" % K] = round(n / (time.perf_counter() - t0), 2)
print(json.dumps(out))
