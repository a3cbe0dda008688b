# EfficientNet backbone: parity tests, timings, kernel breakdown
timeout 600 python -m pytest tests/test_efficientnet.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3
cat > /tmp/effbench.py <<'PY'
import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from object_detection_tracking_amd.efficientdet import EfficientNetBackbone, arch
from object_detection_tracking_amd.weights import synthetic_frames
cases = [("efficientdet-d7", 1536, 1536, 1), ("efficientdet-d7", 1536, 1536, 2), ("efficientdet-d0", 512, 512, 8)]
if len(sys.argv) > 1: cases = cases[:1]
for name, H, W, B in cases:
  w = arch.synthetic_det_weights(name, 0)
  net = EfficientNetBackbone(arch.det_config(name)["backbone"], w, B, H, W, det=name)
  fr = synthetic_frames(B, H, W)
  for _ in range(2): net.forward_async(fr)
  net.synchronize()
  t = time.perf_counter(); n = 5
  for _ in range(n): net.forward_async(fr)
  net.synchronize()
  dt = (time.perf_counter() - t) / n
  print("%s %dx%d b=%d: %.2f ms/step (%.1f frames/s) incl. H2D of uint8 frames" % (name, W, H, B, dt * 1e3, B / dt), flush=True)
  net.close()
PY
python /tmp/effbench.py
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp; rm -rf $R/gpurun_out/prof_eff
timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_eff -o eff -- python /tmp/effbench.py one > /dev/null 2>&1
cd $R; python tools/kernel_stats.py gpurun_out/prof_eff | cut -c1-150 | head -16
find gpurun_out/prof_eff -name "*.db" -size +20M -delete
