# EfficientNet backbone: parity tests, timings, kernel breakdown
timeout 600 python -m pytest tests/test_efficientnet.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3
cat > /tmp/effbench.py <<'PY'
import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from object_detection_tracking_amd import models
from object_detection_tracking_amd.config import make_config
from object_detection_tracking_amd.efficientdet import arch
from object_detection_tracking_amd.weights import synthetic_frames
cases = [("efficientdet-d7", 1536, (1536, 1536)), ("efficientdet-d7", 1536, (1080, 1920)), ("efficientdet-d0", 512, (512, 512))]
if len(sys.argv) > 1: cases = cases[:1]
for name, S, src in cases:
  cfg = make_config(is_efficientdet=True, efficientdet_modelname=name, efficientdet_max_detection_topk=5000,
                    short_edge_size=S, max_size=S)
  cfg.max_size = S
  m = models.get_model(cfg, 0, weights=arch.synthetic_det_weights(name, 0))
  fr = synthetic_frames(1, src[0], src[1])[0]
  for _ in range(2): out = m.predict(fr)
  t = time.perf_counter(); n = 5
  for _ in range(n): out = m.predict(fr)
  dt = (time.perf_counter() - t) / n
  print("%s input %dx%d, frame %dx%d: %.2f ms/frame host-to-host (%.1f frames/s), %d detections" % (name, S, S, src[1], src[0], dt * 1e3, 1 / dt, len(out[0])), flush=True)
  m.close()
PY
python /tmp/effbench.py
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp; rm -rf $R/gpurun_out/prof_eff
timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_eff -o eff -- python /tmp/effbench.py one > /dev/null 2>&1
cd $R; python tools/kernel_stats.py gpurun_out/prof_eff | cut -c1-150 | head -16
find gpurun_out/prof_eff -name "*.db" -size +20M -delete
