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
    batches = [synthetic_frames(B, H, W, seed=s) for s in ((1, 2) if name == "emu" else (1, 2, 3))]
    batches[1] = batches[1].astype(np.float32)                 # float32 feed like the reference
    want = [e.forward(b, want_feats=True, want_pooled=True) for b in batches]
    got = list(e.forward_stream(batches, want_feats=True, want_pooled=True))
    assert len(got) == len(batches)
    for g, w in zip(got, want):
      for a, b in zip(g, w):
        assert np.array_equal(a, b)
    # protocol errors come back as exceptions, not aborts
    t0 = e.submit(batches[0]); t1 = e.submit(batches[1])
    with pytest.raises(OdtError, match="outstanding"):
      e.submit(batches[-1])
    with pytest.raises(OdtError, match="ticket"):
      e.collect(t1 + 5)
    r1 = e.collect(t1); r0 = e.collect(t0)                     # any order
    assert np.array_equal(r0[0], want[0][0]) and np.array_equal(r1[0], want[1][0])
    with pytest.raises(OdtError, match="ticket"):
      e.collect(t0)
    # the tracking loop's default: pooled features only (0.8 MB instead of 40 MB per 8-frame batch) -- the copies ride
    # right behind the forward's tail (side stream by default; compute stream with ODT_TAIL_OVERLAP=0, tested below)
    many = batches + batches
    got = list(e.forward_stream(many))
    assert len(got) == len(many)
    for g, w in zip(got, want + want):
      assert g[4] is None and np.array_equal(g[5], w[5])
      for a, b in zip(g[:4], w[:4]):
        assert np.array_equal(a, b)
    # alternating with tickets that do carry the [M,C,7,7] features (copy stream + event waits)
    ta = e.submit(batches[0], want_feats=True, want_pooled=False)
    tb = e.submit(batches[1], want_feats=False, want_pooled=True)
    ra = e.collect(ta, want_feats=True, want_pooled=False); rb = e.collect(tb, want_feats=False, want_pooled=True)
    assert np.array_equal(ra[4], want[0][4]) and np.array_equal(rb[5], want[1][5])
    with pytest.raises(OdtError, match="ODT_WANT_FEATS"):
      e.collect(e.submit(batches[0], want_feats=False, want_pooled=True), want_feats=True)
  finally:
    m.close()


def test_collect_defaults_follow_the_ticket_and_read_outputs(backend):
  """collect(t) returns what submit() asked for (collect's own defaults used to ask for the [M,C,7,7]
  features of a pooled-only ticket and raised); odt_read_outputs returns the last asynchronous forward's outputs; a
  decoder can write into the slot's pinned buffer and submit without a staging copy."""
  import ctypes as C
  from object_detection_tracking_amd._lib import ODT_DTYPE_U8
  name, lib = backend
  B, H, W = (1, 64, 96) if name == "emu" else (2, 256, 448)
  cfg = small_config(resnet_num_block=[1, 1, 1, 1], im_batch_size=B, rpn_test_post_nms_topk=16, max_size=448)
  m = models.get_model(cfg, 0, weights=weights_for(cfg), lib=lib, is_multi=True)
  try:
    e = m.engine(B, H, W)
    fr = synthetic_frames(B, H, W, seed=7)
    want = e.forward(fr, want_feats=True, want_pooled=True)
    r = e.collect(e.submit(fr, want_feats=False, want_pooled=True))        # defaults = as submitted
    assert r[4] is None and np.array_equal(r[5], want[5]) and np.array_equal(r[0], want[0])
    r = e.collect(e.submit(fr, want_feats=True, want_pooled=False))
    assert r[5] is None and np.array_equal(r[4], want[4])
    # frames written straight into the pinned ingest buffer, submit(None)
    buf = e.ingest_buffer(np.uint8)
    assert buf.shape == fr.shape and buf.dtype == np.uint8
    np.copyto(buf, fr)
    r = e.collect(e.submit(None, want_feats=False, want_pooled=True))
    assert np.array_equal(r[0], want[0]) and np.array_equal(r[5], want[5])
    # submit(None) without a buffer handed out for THIS ticket (the slot's pinned memory holds stale frames), or after
    # asking for the other dtype, is refused instead of silently inferring on whatever the buffer holds
    with pytest.raises(OdtError, match="odt_ingest_buffer"):
      e.submit(None, want_feats=False, want_pooled=True)
    e.ingest_buffer(np.float32)
    e._ingest_dtype = ODT_DTYPE_U8
    with pytest.raises(OdtError, match="odt_ingest_buffer"):
      e.submit(None, want_feats=False, want_pooled=True)
    np.copyto(e.ingest_buffer(np.uint8), fr)                                 # (re-armed: works again)
    r = e.collect(e.submit(None, want_feats=False, want_pooled=True))
    assert np.array_equal(r[0], want[0])
    # read_outputs after an asynchronous forward of another batch
    fr2 = synthetic_frames(B, H, W, seed=8)
    want2 = e.forward(fr2, want_feats=True, want_pooled=True)
    e.forward(fr)                                                            # something else in between
    if name == "hip":
      import torch
      d = torch.from_numpy(fr2).cuda(0)
      e.forward_device_async(d.data_ptr(), ODT_DTYPE_U8)
    else:
      e.lib.check(e.lib.dll.odt_forward_async(e.h, fr2.ctypes.data_as(C.c_void_p), ODT_DTYPE_U8, 0, None))
    got = e.read_outputs(want_feats=True, want_pooled=True)
    for a, b in zip(got, want2):
      assert np.array_equal(a, b)
    # a fresh handle has nothing to read
    e2 = m.engine(B, H + 32, W)
    with pytest.raises(OdtError, match="no forward"):
      e2.read_outputs()
  finally:
    m.close()


def test_tail_overlap_off_is_bit_identical(backend, monkeypatch):
  """ODT_TAIL_OVERLAP=0 (no test set it): the whole forward in stream order on the handle's stream, the
  small D2H copies enqueued behind it -- same results as the default (tail on the side stream) bit for bit."""
  name, lib = backend
  B, H, W = (1, 64, 96) if name == "emu" else (2, 256, 448)
  cfg = small_config(resnet_num_block=[1, 1, 1, 1], im_batch_size=B, rpn_test_post_nms_topk=16, max_size=448)
  batches = [synthetic_frames(B, H, W, seed=s) for s in (1, 2, 3)]
  res = {}
  for mode in ("1", "0"):
    monkeypatch.setenv("ODT_TAIL_OVERLAP", mode)
    m = models.get_model(cfg, 0, weights=weights_for(cfg), lib=lib, is_multi=True)
    try:
      e = m.engine(B, H, W)
      blocking = [e.forward(b, want_feats=False, want_pooled=True) for b in batches]
      streamed = list(e.forward_stream(batches * 2))                       # pooled only: the compute-stream D2H path
      for g, w in zip(streamed, blocking + blocking):
        for a, b in zip(g[:4], w[:4]):
          assert np.array_equal(a, b)
        assert np.array_equal(g[5], w[5])
      res[mode] = blocking
    finally:
      m.close()
  for a, b in zip(res["1"], res["0"]):
    for x, y in zip(a, b):
      assert (x is None and y is None) or np.array_equal(x, y)


# ---- device-side frame resize (reference obj_detect_tracking.py:597-608: astype(float32) +
# resizeImage on the host for every frame) ---------------------------------------------------------
def _raw_vs_host_resize(lib, src_hw, dtype, short_edge, max_size):
  from object_detection_tracking_amd.nn import get_new_hw, resizeImage
  cfg = small_config(resnet_num_block=[1, 1, 1, 1], short_edge_size=short_edge, max_size=max_size)
  w = weights_for(cfg)
  frame = synthetic_frames(1, src_hw[0], src_hw[1], seed=7)[0]
  if dtype == np.float32:
    frame = frame.astype(np.float32) + np.float32(0.25)
  m = models.get_model(cfg, 0, weights=w, lib=lib)
  try:
    # reference order: float32 first, then bilinear resize on the host, then the forward
    host = resizeImage(frame.astype(np.float32), short_edge, max_size)
    neww, newh = get_new_hw(src_hw[0], src_hw[1], short_edge, max_size)
    assert host.shape[:2] == (newh, neww) and (newh, neww) != tuple(src_hw)
    b0, l0, p0, f0 = m.predict(host)
    b1, l1, p1, f1, scale = m.predict_raw(frame)
    assert np.isclose(scale, (newh / src_hw[0] + neww / src_hw[1]) / 2)
    # the device resize follows the host restatement operation by operation: identical results
    assert np.array_equal(b0, b1) and np.array_equal(l0, l1) and np.array_equal(p0, p1)
    assert np.array_equal(f0, f1)
    assert len(b0) > 0
    # ... and against the ORACLE's resize (oracle/imgproc.py: float64, pixel by pixel; pinned in test_oracle_golden):
    # the frames differ by rounding only (<= 6.2e-5 on the 0..255 scale), so the detections agree as a set within the path's tolerance
    from oracle import imgproc
    from common import match_detections
    b2, l2, p2, f2 = m.predict(imgproc.resize_image(frame.astype(np.float32), short_edge, max_size))
    miss, extra = match_detections(b1, l1, p1, b2, l2, p2, 1e-3 * max(1.0, max(newh, neww) / 128.0), 1e-4)
    assert miss + extra <= max(1, len(b2) // 50), (miss, extra, len(b2))
  finally:
    m.close()


def test_device_resize_matches_host_resize(backend):
  name, lib = backend
  _raw_vs_host_resize(lib, (48, 80), np.uint8, 64, 128)         # upscale 4/3 (720p -> 1080p style)
  if name == "hip":                                             # (the simulator runs one case only)
    _raw_vs_host_resize(lib, (72, 128), np.uint8, 96, 256)
    _raw_vs_host_resize(lib, (150, 260), np.uint8, 96, 160)     # downscale, long edge bound by max_size
    _raw_vs_host_resize(lib, (72, 128), np.float32, 96, 256)    # float32 feed
    _raw_vs_host_resize(lib, (720, 1280), np.uint8, 1080, 1920)  # 720p stream into the 1080p plan


def test_device_resize_batch_and_pipeline(backend):
  from object_detection_tracking_amd.nn import resizeImage
  name, lib = backend
  if name == "emu":
    pytest.skip("covered by test_device_resize_matches_host_resize on the simulator; full case on the GPU")
  cfg = small_config(resnet_num_block=[1, 1, 1, 1], im_batch_size=2, rpn_test_post_nms_topk=48,
                     short_edge_size=96, max_size=256)
  w = weights_for(cfg)
  frames = synthetic_frames(2, 72, 128, seed=11)
  m = models.get_model(cfg, 0, weights=w, lib=lib, is_multi=True)
  try:
    host = np.stack([resizeImage(f.astype(np.float32), 96, 256) for f in frames])
    ref = m.predict_batch(host)
    got = m.predict_batch_raw(frames)
    for a, b in zip(ref, got[:5]):
      assert np.array_equal(a, b)
    # same frames through the pinned double-buffered ingest (odt_submit / odt_collect)
    e, _ = m.engine_for_raw(2, 72, 128)
    outs = list(e.forward_stream([frames, frames[::-1].copy(), frames]))
    assert np.array_equal(outs[0][0], ref[0]) and np.array_equal(outs[2][0], ref[0])
    assert np.array_equal(outs[1][0][::-1], ref[0])
    # switching back to plan-sized frames on the same handle
    e.set_source_size(e.height, e.width)
    assert np.array_equal(e.forward(host)[0], ref[0])
  finally:
    m.close()


# ---- handle life cycle / threading (one handle per (GPU, stream), the
# library is re-entrant across handles, ctypes releases the GIL during the forward) -----------------
@pytest.mark.gpu
def test_two_handles_in_two_threads_match_sequential(hip_lib):
  import threading
  cfg = small_config(resnet_num_block=[1, 1, 2, 3], im_batch_size=2, rpn_test_post_nms_topk=64,
                     max_size=448)
  w = weights_for(cfg)
  frames = [synthetic_frames(2, 256, 448, seed=s) for s in (21, 22)]
  ms = [models.get_model(cfg, 0, weights=w, lib=hip_lib, is_multi=True) for _ in range(2)]
  try:
    want = [ms[i].predict_batch(frames[i]) for i in range(2)]
    got = [None, None]
    def work(i):
      for _ in range(5):
        got[i] = ms[i].predict_batch(frames[i])
    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    [t.start() for t in th]; [t.join() for t in th]
    for i in range(2):
      for a, b in zip(want[i], got[i]):
        assert np.array_equal(a, b)
  finally:
    [m.close() for m in ms]


@pytest.mark.gpu
def test_create_destroy_does_not_leak_device_memory(hip_lib):
  import torch
  cfg = small_config(resnet_num_block=[1, 1, 2, 3], max_size=448)
  w = weights_for(cfg)
  fr = synthetic_frames(1, 256, 448)[0]
  def cycle():
    m = models.get_model(cfg, 0, weights=w, lib=hip_lib)
    m.predict(fr)
    e = m.engine(1, 256, 448)
    list(e.forward_stream([fr[None]] * 3))          # allocates the pinned ingest slots too
    m.close()
  cycle(); cycle()
  torch.cuda.synchronize()
  free0 = torch.cuda.mem_get_info(0)[0]
  for _ in range(6):
    cycle()
  torch.cuda.synchronize()
  free1 = torch.cuda.mem_get_info(0)[0]
  assert free0 - free1 < 8 << 20, "device memory shrank by %d bytes over 6 create/destroy cycles" % (free0 - free1)


@pytest.mark.gpu
def test_back_to_back_forwards_with_split_k_in_trunk_and_tail(hip_lib):
  """b=4 at 1080p: res5 (trunk) and the box-head FCs (tail) both run split-K, and the tail of forward i runs on the
  side stream under the trunk of forward i+1 -- the two groups must not share a partial-sum buffer.  Pipelined
  batches must equal the blocking forward bit for bit."""
  from common import make_config
  cfg = make_config(rpn_test_post_nms_topk=300, im_batch_size=4)
  m = models.get_model(cfg, 0, weights=weights_for(cfg), lib=hip_lib, is_multi=True)
  try:
    e = m.engine(4, 1080, 1920)
    d = e.describe()
    assert d["split_launches_by_family"]["of_split3_with_split_k"] >= 2, d
    batches = [synthetic_frames(4, 1080, 1920, seed=s) for s in (1, 2, 3)]
    want = [e.forward(b, want_feats=False, want_pooled=True) for b in batches]
    got = list(e.forward_stream(batches * 2))
    for g, w in zip(got, want + want):
      for a, b in zip(g[:4], w[:4]):
        assert np.array_equal(a, b)
      assert np.array_equal(g[5], w[5])
  finally:
    m.close()


@pytest.mark.gpu
def test_submit_large_batch_staged_by_several_threads(hip_lib):
  """odt_submit_ex copies a batch of pageable frames into the slot's pinned buffer with up to four threads (round 6: 50 MB of
  uint8 at 8 x 1080p; the call's host time 1.93 -> 1.09 ms).  2 x 1080p float32 frames = 50 MB: the chunked copy
  must be the blocking forward's input bit for bit (odd sizes: the last chunk is shorter)."""
  B, H, W = 2, 1080, 1920
  cfg = small_config(resnet_num_block=[1, 1, 1, 1], im_batch_size=B, rpn_test_post_nms_topk=32, max_size=1920, short_edge_size=1080)
  m = models.get_model(cfg, 0, weights=weights_for(cfg), lib=hip_lib, is_multi=True)
  try:
    e = m.engine(B, H, W)
    batches = [synthetic_frames(B, H, W, seed=s).astype(np.float32) for s in (11, 12)]
    want = [e.forward(b, want_feats=False, want_pooled=True) for b in batches]
    got = list(e.forward_stream(batches, want_feats=False, want_pooled=True))
    for g, w in zip(got, want):
      for a, b in zip(g, w):
        assert (a is None and b is None) or np.array_equal(a, b)
  finally:
    m.close()
