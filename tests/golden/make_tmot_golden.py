#!/usr/bin/env python
"""Golden track tables from the reference's OWN tmot/multitracker.py (JDETracker), run here.

tmot imports three third-party modules that are not installed in the build container; they are
replaced by stubs that restate their published behaviour (everything else -- STrack, the Kalman
filter, the association cascade, list bookkeeping -- is the reference's unmodified code):
  * numba.jit                  -> identity decorator;
  * lap.lapjv(cost, extend_cost=True, cost_limit=t)  (lap 0.4.0 _lapjv.pyx: the cost matrix is
    extended to (n+m)x(n+m) with cost_limit/2 in the off-diagonal blocks and 0 in the lower-right
    block, solved exactly, and assignments into the extension are reported as -1) -> the same
    extension solved with scipy.optimize.linear_sum_assignment (exact; equal up to ties);
  * cython_bbox.bbox_overlaps  -> the py-faster-rcnn definition ("+1" widths/heights).
Writes tests/golden/tmot_ref.npz.  Needs /root/reference (build container only).
"""
import os, sys, types
import numpy as np
from scipy.optimize import linear_sum_assignment

REF = "/root/reference"
np.float = float; np.int = int


def lapjv(cost, extend_cost=False, cost_limit=np.inf, return_cost=True):
  cost = np.asarray(cost, np.float64)
  nr, nc = cost.shape
  n = nr + nc
  ext = np.empty((n, n), np.float64)
  ext[:] = cost_limit / 2.0 if cost_limit < np.inf else cost.max() + 1
  ext[nr:, nc:] = 0
  ext[:nr, :nc] = cost
  r, c = linear_sum_assignment(ext)
  x = np.full(n, -1, np.int64); y = np.full(n, -1, np.int64)
  x[r] = c; y[c] = r
  x = x[:nr].copy(); y = y[:nc].copy()
  x[x >= nc] = -1; y[y >= nr] = -1
  total = float(sum(cost[i, x[i]] for i in range(nr) if x[i] >= 0))
  return total, x, y


def bbox_overlaps(boxes, query):
  b = np.asarray(boxes, np.float64); q = np.asarray(query, np.float64)
  out = np.zeros((b.shape[0], q.shape[0]), np.float64)
  for k in range(q.shape[0]):
    qa = (q[k, 2] - q[k, 0] + 1) * (q[k, 3] - q[k, 1] + 1)
    for n in range(b.shape[0]):
      iw = min(b[n, 2], q[k, 2]) - max(b[n, 0], q[k, 0]) + 1
      if iw > 0:
        ih = min(b[n, 3], q[k, 3]) - max(b[n, 1], q[k, 1]) + 1
        if ih > 0:
          ua = (b[n, 2] - b[n, 0] + 1) * (b[n, 3] - b[n, 1] + 1) + qa - iw * ih
          out[n, k] = iw * ih / ua
  return out


def install_stubs():
  nb = types.ModuleType("numba"); nb.jit = lambda f=None, **kw: f if f is not None else (lambda g: g)
  lp = types.ModuleType("lap"); lp.lapjv = lapjv
  cb = types.ModuleType("cython_bbox"); cb.bbox_overlaps = bbox_overlaps
  sys.modules.update(numba=nb, lap=lp, cython_bbox=cb)
  if REF not in sys.path:
    sys.path.append(REF)


def make_sequence(seed=11, nobj=10, frames=45, D=64):
  """Objects that move, blink out for several frames (lost -> re-found / removed), cross each other
  and occasionally produce duplicate detections."""
  rng = np.random.default_rng(seed)
  centres = rng.standard_normal((nobj, D)).astype(np.float32)
  pos = rng.uniform(80, 900, (nobj, 2)); vel = rng.uniform(-10, 10, (nobj, 2))
  size = rng.uniform(40, 140, (nobj, 2))
  seq = []
  for fr in range(frames):
    dets = []
    for o in range(nobj):
      if (fr // 4 + o) % 5 == 0 or rng.uniform() < 0.08:
        continue
      p = pos[o] + vel[o] * fr + rng.normal(0, 2.0, 2)
      tlwh = np.r_[p, size[o] * (1 + 0.03 * rng.standard_normal(2))]
      f = (centres[o] + 0.3 * rng.standard_normal(D)).astype(np.float32)
      dets.append((tlwh, float(rng.uniform(0.5, 0.99)), f))
      if rng.uniform() < 0.05:                       # duplicate detection of the same object
        dets.append((tlwh + rng.normal(0, 1.0, 4), float(rng.uniform(0.3, 0.9)),
                     (centres[o] + 0.3 * rng.standard_normal(D)).astype(np.float32)))
    if fr % 9 == 8:                                  # clutter, sometimes below the birth threshold
      dets.append((np.r_[rng.uniform(0, 900, 2), 50, 90], float(rng.uniform(0.2, 0.9)),
                   rng.standard_normal(D).astype(np.float32)))
    seq.append(dets)
  return seq


def main():
  install_stubs()
  from tmot.multitracker import JDETracker
  from tmot.basetrack import BaseTrack
  seq = make_sequence()
  BaseTrack._count = 0
  trk = JDETracker(0.6, track_max_second_lost=2.0, emb_max_dist=0.7, iou_max_dist1=0.8, iou_max_dist2=0.9,
                   emb_smooth_alpha=0.9, frame_gap=8., frame_rate=30.)
  n, tlwh, conf, feat, out_n, out = [], [], [], [], [], []
  for dets in seq:
    n.append(len(dets))
    for t, c, f in dets:
      tlwh.append(t); conf.append(c); feat.append(f.copy())
    tracks = trk.update([(t.copy(), c, f.copy()) for t, c, f in dets])
    out_n.append(len(tracks))
    for t in tracks:
      out.append([t.track_id] + list(t.tlwh) + list(t.cur_det_tlwh) + [t.cur_det_conf, t.score,
                  t.tracklet_len, t.start_frame])
  here = os.path.dirname(os.path.abspath(__file__))
  np.savez_compressed(os.path.join(here, "tmot_ref.npz"), seq_n=np.asarray(n), seq_tlwh=np.asarray(tlwh),
                      seq_conf=np.asarray(conf), seq_feat=np.asarray(feat, np.float32),
                      out_n=np.asarray(out_n), out=np.asarray(out, np.float64),
                      n_lost=np.asarray([len(trk.lost_stracks)]), n_removed=np.asarray([len(trk.removed_stracks)]))
  print("frames", len(seq), "dets", len(tlwh), "track rows", len(out), "max id", int(np.asarray(out)[:, 0].max()),
        "lost", len(trk.lost_stracks), "removed", len(trk.removed_stracks))


if __name__ == "__main__":
  main()
