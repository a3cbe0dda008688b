"""Pipelined ingest (odt_submit / odt_collect): results must be identical to
the blocking odt_forward for every batch of a stream, in order, with two batches in flight."""
import numpy as np
import pytest

from common import small_config, weights_for
from object_detection_tracking_amd import models
from object_detection_tracking_amd._lib import OdtError
from object_detection_tracking_amd.weights import synthetic_frames


def test_submit_collect_equals_forward(backend):
  name, lib = backend
  B, H, W = (1, 64, 96) if name == "emu" else (2, 256, 448)
  cfg = small_config(resnet_num_block=[1, 1, 1, 1], im_batch_size=B, rpn_test_post_nms_topk=16,
                     max_size=448)
  m = models.get_model(cfg, 0, weights=weights_for(cfg), lib=lib, is_multi=True)
  try:
    e = m.engine(B, H, W)
    batches = [synthetic_frames(B, H, W, seed=s) for s in (1, 2, 3)]
    batches[1] = batches[1].astype(np.float32)                 # float32 feed like the reference
    want = [e.forward(b, want_feats=True, want_pooled=True) for b in batches]
    got = list(e.forward_stream(batches, want_feats=True, want_pooled=True))
    assert len(got) == 3
    for g, w in zip(got, want):
      for a, b in zip(g, w):
        assert np.array_equal(a, b)
    # protocol errors come back as exceptions, not aborts
    t0 = e.submit(batches[0]); t1 = e.submit(batches[1])
    with pytest.raises(OdtError, match="outstanding"):
      e.submit(batches[2])
    with pytest.raises(OdtError, match="ticket"):
      e.collect(t1 + 5)
    r1 = e.collect(t1); r0 = e.collect(t0)                     # any order
    assert np.array_equal(r0[0], want[0][0]) and np.array_equal(r1[0], want[1][0])
    with pytest.raises(OdtError, match="ticket"):
      e.collect(t0)
  finally:
    m.close()
