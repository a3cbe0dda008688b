#!/bin/bash
mkdir -p gpurun_out
timeout 170 python - <<'PY' 2>&1 | tail -40
import faulthandler, sys, time, os
faulthandler.dump_traceback_later(100, exit=True)
sys.path.insert(0, os.getcwd())
import numpy as np
import bench
from object_detection_tracking_amd import models
from object_detection_tracking_amd.config import make_config
from object_detection_tracking_amd.weights import synthetic_frames, synthetic_weights
B, H, W = 8, 1080, 1920
cfg = make_config(rpn_test_post_nms_topk=300, im_batch_size=B, max_size=1920, short_edge_size=1080)
w = synthetic_weights(cfg, seed=0)
m = models.get_model(cfg, 0, weights=w, is_multi=True)
eng = m.engine(B, H, W)
frames = synthetic_frames(B, H, W, seed=1234)
t0 = time.perf_counter()
out = list(eng.forward_stream([frames] * 3))
print("forward_stream x3 ok %.2fs" % (time.perf_counter() - t0), out[0][3], flush=True)
t0 = time.perf_counter()
r = bench.detect_track_leg(eng, frames, B, 0, nbatches=4)
print("detect_track ok %.2fs" % (time.perf_counter() - t0), r, flush=True)
t0 = time.perf_counter()
import torch
print("nproc", os.cpu_count(), "torch threads", torch.get_num_threads(), flush=True)
from oracle.graph import OracleModel
om = OracleModel(cfg, w)
om.forward_multi(frames[:1]); print("oracle pass default threads %.2fs" % (time.perf_counter() - t0), flush=True)
torch.set_num_threads(os.cpu_count()); t0 = time.perf_counter()
om.forward_multi(frames[:1]); print("oracle pass nproc threads %.2fs" % (time.perf_counter() - t0), flush=True)
PY
