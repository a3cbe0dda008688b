"""Generate golden fixtures by RUNNING the reference's own Python modules.

Runs only in the build container (needs /root/reference); the GPU box and the test-suite only
read the committed .npz files.  Importable reference modules on this path:
generate_anchors.py (as is) and deep_sort/* + application_util/preprocessing.py after restoring
the numpy aliases the reference uses (np.float / np.int were removed in numpy 2) and stubbing
the unused cv2 import.

  python tests/golden/make_golden_from_reference.py
"""
import os
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))

np.float = float   # noqa: deep_sort/detection.py:30 et al.
np.int = int
sys.modules.setdefault("cv2", types.ModuleType("cv2"))
sys.path.insert(0, REF)

import generate_anchors as ga                                    # noqa: E402
from application_util import preprocessing                       # noqa: E402
from deep_sort import nn_matching                                # noqa: E402
from deep_sort.detection import Detection                        # noqa: E402
from deep_sort.tracker import Tracker                            # noqa: E402
from deep_sort.utils import create_obj_infos                     # noqa: E402


def anchors():
  out = {}
  # known-answer table of generate_anchors.py:20-38 (python output == table - 1)
  out["kat_default"] = ga.generate_anchors()
  strides, sizes, ratios = (4, 8, 16, 32, 64), (32, 64, 128, 256, 512), (0.5, 1, 2)
  for s, z in zip(strides, sizes):
    out["cell_s%d" % s] = ga.generate_anchors(
        s, scales=np.array([z], dtype=float) / s, ratios=np.array(ratios, dtype=float))
  np.savez(os.path.join(OUT, "anchors_ref.npz"), **out)


def deep_sort():
  rng = np.random.default_rng(2024)
  out = {}
  D = 256
  # ---- cosine metric: raw helper + class with budget trimming
  centres = rng.standard_normal((12, D)).astype(np.float32)
  def feat(i):
    return (centres[i] + 0.1 * rng.standard_normal(D)).astype(np.float32)
  metric = nn_matching.NearestNeighborDistanceMetric("cosine", 0.5, budget=5)
  seen = set()
  for rnd in range(8):                       # 8 frames of updates, tracks 0..9
    f = [feat(t) for t in range(10) if (t + rnd) % 3 != 0]
    ids = [t for t in range(10) if (t + rnd) % 3 != 0]
    seen.update(ids)
    metric.partial_fit(np.asarray(f), np.asarray(ids), sorted(seen))
  dets = np.asarray([feat(i % 12) for i in range(30)], np.float32)
  targets = list(range(10))
  out["nn_dets"] = dets
  out["nn_targets"] = np.asarray(targets)
  out["nn_gallery"] = np.concatenate([np.asarray(metric.samples[t]) for t in targets])
  out["nn_seg"] = np.r_[0, np.cumsum([len(metric.samples[t]) for t in targets])].astype(np.int32)
  out["nn_cost"] = metric.distance(dets, targets)
  a = rng.standard_normal((4, D)).astype(np.float32); b = rng.standard_normal((6, D)).astype(np.float32)
  out["cos_a"] = a; out["cos_b"] = b
  out["cos_dist"] = nn_matching._cosine_distance(a, b)
  out["cos_nn"] = nn_matching._nn_cosine_distance(a, b)

  # ---- Detection + create_obj_infos
  N = 16
  xy = rng.uniform(0, 900, (N, 2)); wh = rng.uniform(10, 300, (N, 2))
  boxes = np.concatenate([xy, xy + wh], 1).astype(np.float32)
  probs = rng.uniform(0.5, 1.0, N).astype(np.float32)
  labels = rng.integers(1, 15, N).astype(np.int64)
  box_feats = rng.standard_normal((N, D, 7, 7)).astype(np.float32)
  names = ["BG", "Vehicle", "Person", "Parking_Meter", "Tree", "Skateboard", "Prop_Overshoulder",
           "Construction_Barrier", "Door", "Dumpster", "Push_Pulled_Object", "Construction_Vehicle",
           "Prop", "Bike", "Animal"]
  id2class = {i: n for i, n in enumerate(names)}
  out["coi_boxes"] = boxes; out["coi_probs"] = probs; out["coi_labels"] = labels
  out["coi_feats"] = box_feats
  for obj in ("Person", "Vehicle"):
    d = create_obj_infos(7, boxes.copy(), probs, labels, box_feats, id2class, [obj], 0.85, 0, 1.5)
    out["coi_%s_tlwh" % obj] = np.asarray([x.tlwh for x in d]).reshape(-1, 4)
    out["coi_%s_conf" % obj] = np.asarray([x.confidence for x in d])
    out["coi_%s_feat" % obj] = np.asarray([x.feature for x in d]).reshape(-1, D)
    out["coi_%s_tlbr" % obj] = np.asarray([x.to_tlbr() for x in d]).reshape(-1, 4)
    out["coi_%s_xyah" % obj] = np.asarray([x.to_xyah() for x in d]).reshape(-1, 4)

  # ---- tracker-side NMS (application_util/preprocessing.py:6-73)
  tl = np.concatenate([xy, wh], 1)
  out["tnms_boxes"] = tl; out["tnms_scores"] = probs
  out["tnms_keep"] = np.asarray(preprocessing.non_max_suppression(tl.copy(), 0.85, probs), np.int64)
  out["tnms_keep_05"] = np.asarray(preprocessing.non_max_suppression(tl.copy(), 0.5, probs), np.int64)

  # ---- a short tracking sequence through the reference Tracker (cosine metric): the cost
  # matrices it asked for and the resulting track table, to pin the HIP metric as a drop-in.
  metric = nn_matching.NearestNeighborDistanceMetric("cosine", 0.5, budget=5)
  calls = []
  orig = metric.distance
  def spy(features, targets):
    c = orig(features, targets)
    calls.append((np.asarray(features, np.float32), np.asarray(targets),
                  np.concatenate([np.asarray(metric.samples[t]) for t in targets]) if len(targets) else np.zeros((0, D), np.float32),
                  np.r_[0, np.cumsum([len(metric.samples[t]) for t in targets])].astype(np.int32), c.copy()))
    return c
  metric.distance = spy
  tracker = Tracker(metric, max_iou_distance=0.5)
  pos = rng.uniform(100, 800, (6, 2)); vel = rng.uniform(-6, 6, (6, 2))
  seq_tlwh, seq_feat, seq_tracks = [], [], []
  for fr in range(25):
    dets = []
    tl_f, ft_f = [], []
    for o in range(6):
      if (o + fr) % 7 == 0:
        continue
      p = pos[o] + vel[o] * fr + rng.normal(0, 1.0, 2)
      tlwh = np.r_[p, 40 + 5 * o, 90 + 8 * o]
      f = feat(o)
      dets.append(Detection(tlwh, 0.95, f)); tl_f.append(tlwh); ft_f.append(f)
    tracker.predict(); tracker.update(dets)
    seq_tlwh.append(np.asarray(tl_f)); seq_feat.append(np.asarray(ft_f))
    seq_tracks.append(np.asarray([[t.track_id] + list(t.to_tlwh()) for t in tracker.tracks
                                  if t.is_confirmed() and t.time_since_update <= 1]).reshape(-1, 5))
  out["seq_n"] = np.asarray([len(x) for x in seq_tlwh])
  out["seq_tlwh"] = np.concatenate(seq_tlwh); out["seq_feat"] = np.concatenate(seq_feat)
  out["seq_tracks_n"] = np.asarray([len(x) for x in seq_tracks])
  out["seq_tracks"] = np.concatenate(seq_tracks)
  out["seq_calls"] = np.asarray(len(calls))
  for i, (f, t, g, s, c) in enumerate(calls):
    out["call%d_feat" % i] = f; out["call%d_targets" % i] = t; out["call%d_gal" % i] = g
    out["call%d_seg" % i] = s; out["call%d_cost" % i] = c
  np.savez_compressed(os.path.join(OUT, "deep_sort_ref.npz"), **out)
  print("metric.distance calls recorded:", len(calls))


if __name__ == "__main__":
  anchors()
  deep_sort()
  for f in ("anchors_ref.npz", "deep_sort_ref.npz"):
    print(f, os.path.getsize(os.path.join(OUT, f)))
