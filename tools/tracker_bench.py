#!/usr/bin/env python
"""Tracker half of BASELINE config #3 on a seeded synthetic detection sequence (200 frames,
T ~ 64 tracks, N ~ 100 detections per frame, nn_budget 5).

  python tools/tracker_bench.py reference   # the UNMODIFIED reference deep_sort (numpy / scipy) on this host's CPU --
                                            # needs /root/reference, i.e. the build container; writes
                                            # profiles/r02_tracker_cpu_baseline.json
  python tools/tracker_bench.py native      # object_detection_tracking_amd.deep_sort.Tracker (C++ core + HIP cosine
                                            # kernel) on the MI355X box; writes gpurun_out/r02_tracker_native.json

Both runs see byte-identical detections (same seed) and report ms per frame (predict + update), the number of
confirmed tracks at the end and a checksum of the final track ids, so the two result files can be compared.
"""
import json, os, sys, time, types
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def sequence(frames=200, objects=64, clutter=36, D=256, seed=7):
  rng = np.random.default_rng(seed)
  pos = rng.uniform(100, 1700, (objects, 2)); vel = rng.uniform(-4, 4, (objects, 2))
  wh = rng.uniform(40, 160, (objects, 2))
  feat = rng.standard_normal((objects, D)).astype(np.float32)
  out = []
  for t in range(frames):
    p = pos + vel * t
    boxes = np.concatenate([p - wh / 2 + rng.normal(0, 1.0, p.shape), wh + rng.normal(0, 1.0, wh.shape)], 1)
    f = feat + 0.15 * rng.standard_normal(feat.shape).astype(np.float32)
    cb = np.concatenate([rng.uniform(0, 1800, (clutter, 2)), rng.uniform(30, 120, (clutter, 2))], 1)
    cf = rng.standard_normal((clutter, D)).astype(np.float32)
    conf = rng.uniform(0.86, 1.0, objects + clutter)
    out.append((np.concatenate([boxes, cb]).astype(np.float64), conf, np.concatenate([f, cf]).astype(np.float32)))
  return out


def run(make_tracker, Detection, seq):
  trk = make_tracker()
  times = []
  for boxes, conf, feats in seq:
    dets = [Detection(boxes[i], conf[i], feats[i]) for i in range(len(boxes))]
    t0 = time.perf_counter()
    trk.predict()
    trk.update(dets)
    times.append(time.perf_counter() - t0)
  tr = trk.tracks
  conf_ids = sorted(int(t.track_id) for t in tr if t.is_confirmed())
  return {"frames": len(seq), "detections_per_frame": len(seq[0][0]), "ms_per_frame_median": 1e3 * float(np.median(times)),
          "ms_per_frame_mean": 1e3 * float(np.mean(times)), "ms_total": 1e3 * float(np.sum(times)),
          "confirmed_tracks_at_end": len(conf_ids), "tracks_at_end": len(tr),
          "confirmed_id_checksum": int(sum((i + 1) * v for i, v in enumerate(conf_ids)) % 1000003)}


def main():
  mode = sys.argv[1] if len(sys.argv) > 1 else "native"
  seq = sequence()
  if mode == "reference":
    REF = "/root/reference"
    np.float = float; np.int = int                                 # numpy-2 removed the aliases the reference uses
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    sys.path.append(REF)
    from deep_sort import nn_matching
    from deep_sort.detection import Detection
    from deep_sort.tracker import Tracker
    mk = lambda: Tracker(nn_matching.NearestNeighborDistanceMetric("cosine", 0.5, 5), max_iou_distance=0.5, max_age=60, n_init=1)
    res = run(mk, Detection, seq)
    res.update(kind="reference", what="unmodified /root/reference deep_sort.Tracker (numpy + scipy.optimize.linear_sum_assignment)",
               cpu_model=open("/proc/cpuinfo").read().split("model name")[1].split(":")[1].split("\n")[0].strip(),
               nproc=os.cpu_count())
    path = os.path.join(ROOT, "profiles", "r02_tracker_cpu_baseline.json")
  else:
    from object_detection_tracking_amd.deep_sort import Detection, NearestNeighborDistanceMetric, Tracker
    mk = lambda: Tracker(NearestNeighborDistanceMetric("cosine", 0.5, 5), max_iou_distance=0.5, max_age=60, n_init=1)
    res = run(mk, Detection, seq)
    res.update(kind="native", what="object_detection_tracking_amd.deep_sort.Tracker: C++ core in libodt_hip.so, one HIP cosine-NN "
                                   "kernel call per update on the tracker's own stream (times include the Python / ctypes glue)")
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    path = os.path.join(ROOT, "gpurun_out", "r02_tracker_native.json")
  json.dump(res, open(path, "w"), indent=1)
  print(json.dumps(res))


if __name__ == "__main__":
  main()
