"""Cosine nearest-neighbour appearance metric, HIP-backed.

Same class / methods / sample bookkeeping as the reference's
deep_sort/nn_matching.py:99-177 (``NearestNeighborDistanceMetric``): ``samples`` maps a track
id to its last ``budget`` features; ``distance(features, targets)`` returns the float64
[len(targets), len(features)] matrix of the smallest cosine distance between each target's
gallery and each feature.  The reference loops over targets in Python calling numpy
(:174-177); here all galleries are concatenated and ONE call of the C ABI ``odt_nn_cosine``
computes the whole matrix on the GPU.  No CPU path.
"""
import numpy as np

from .. import ops


class NearestNeighborDistanceMetric(object):

  def __init__(self, metric, matching_threshold, budget=None, lib=None, device=0):
    if metric != "cosine":
      # the tracking scripts only ever use "cosine" (obj_detect_tracking.py:551-552)
      raise ValueError("only the 'cosine' metric is on the hot path")
    self.matching_threshold = matching_threshold
    self.budget = budget
    self.samples = {}
    self._lib = lib
    self._device = device

  def partial_fit(self, features, targets, active_targets):
    for feature, target in zip(features, targets):
      bucket = self.samples.setdefault(target, [])
      bucket.append(feature)
      if self.budget is not None and len(bucket) > self.budget:
        del bucket[:len(bucket) - self.budget]
    self.samples = {k: self.samples[k] for k in active_targets}

  def distance(self, features, targets):
    features = np.asarray(features, dtype=np.float32)
    cost = np.zeros((len(targets), len(features)))
    if len(targets) == 0 or len(features) == 0:
      return cost
    rows, seg = [], [0]
    for t in targets:
      g = self.samples[t]
      rows.extend(g)
      seg.append(seg[-1] + len(g))
    gallery = np.asarray(rows, dtype=np.float32).reshape(seg[-1], features.shape[1])
    return ops.nn_cosine(gallery, np.asarray(seg, np.int32), features, lib=self._lib,
                         device=self._device)
