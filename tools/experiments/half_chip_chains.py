"""Experiment (round 6): two independent half-batch chains, each confined to half of the chip's CUs.

The b = 8 step is a chain of one-round launches whose workgroups all reach their HBM-bound phases (fused bottleneck tails,
epilogues, the res2 / res3 1x1 layers) together, while the matrix-bound phases run against the power budget.  Images of a
batch are independent, so the batch can run as TWO chains of four images; if each chain only owns half of the CUs (a CU-masked
HIP stream: hipExtStreamCreateWithCUMask) and the chains drift out of phase, one chain's HBM-bound phase lies under the other's
MFMA-bound phase: bandwidth and power budget are shared in time instead of being demanded in lock-step.

Prints frames/s for: one b=8 handle (the product), two b=4 handles on plain streams, two b=4 handles on half-chip streams
(three ways of cutting the CU mask), one b=4 handle alone on a half-chip stream (is the mask effective?).
Needs no library change: odt_forward_async takes a caller stream.
"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from object_detection_tracking_amd import models
from object_detection_tracking_amd._lib import ODT_DTYPE_U8
from object_detection_tracking_amd.config import make_config
from object_detection_tracking_amd.weights import synthetic_frames, synthetic_weights

hip = C.CDLL("libamdhip64.so")


def masked_stream(words):
  st = C.c_void_p()
  arr = (C.c_uint32 * len(words))(*words)
  rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), C.c_uint32(len(words)), arr)
  assert rc == 0, rc
  return st.value


def mask_words(pred, ncu=256):
  w = [0] * (ncu // 32)
  for i in range(ncu):
    if pred(i):
      w[i // 32] |= 1 << (i % 32)
  return w


def run(engs, devs, streams, steps, stagger_ms=0.0):
  for e in engs:
    e.synchronize()
  torch.cuda.synchronize()
  # warm-up
  for k in range(3):
    for e, d, s in zip(engs, devs, streams):
      e.forward_device_async(d[k % len(d)].data_ptr(), ODT_DTYPE_U8, stream=s)
  for e in engs:
    e.synchronize()
  t0 = time.perf_counter()
  for k in range(steps):
    for i, (e, d, s) in enumerate(zip(engs, devs, streams)):
      if k == 0 and i == 1 and stagger_ms > 0:
        time.sleep(stagger_ms * 1e-3)
      e.forward_device_async(d[k % len(d)].data_ptr(), ODT_DTYPE_U8, stream=s)
  for e in engs:
    e.synchronize()
  torch.cuda.synchronize()
  dt = time.perf_counter() - t0
  nfr = sum(d[0].shape[0] for d in devs) * steps
  return nfr / dt


def main():
  H, W, steps = 1080, 1920, int(os.environ.get("STEPS", "16"))
  torch.cuda.set_device(0)
  out = {}
  cfg8 = make_config(rpn_test_post_nms_topk=300, im_batch_size=8, max_size=W, short_edge_size=H, conv_split_family=0)
  cfg4 = make_config(rpn_test_post_nms_topk=300, im_batch_size=4, max_size=W, short_edge_size=H, conv_split_family=0)
  w = synthetic_weights(cfg8, seed=0)
  m8 = models.get_model(cfg8, 0, weights=w, is_multi=True)
  e8 = m8.engine(8, H, W)
  d8 = [torch.from_numpy(synthetic_frames(8, H, W, seed=1234 + 77 * r)).cuda(0) for r in range(4)]
  out["one_b8"] = run([e8], [d8], [None], steps)
  # the plan's tile policy is written for 256 CUs (a layer takes 256-row tiles when it has >= 256 of them ...): a chain that
  # owns 128 CUs wants the thresholds halved, or its b = 4 plan falls back to small tiles / split-K where b = 8 does not
  half_env = {"ODT_CONV_SPLIT_MINTILES": "128", "ODT_CONV_SPLIT3_MINTILES": "100", "ODT_STEM_GRID": "128"} if os.environ.get("HALF_POLICY", "1") == "1" else {}
  os.environ.update(half_env)
  m4 = [models.get_model(cfg4, 0, weights=w, is_multi=True) for _ in range(2)]
  e4 = [m.engine(4, H, W) for m in m4]
  for k in half_env:
    os.environ.pop(k)
  out["b4_handle"] = {k: e4[0].describe().get(k) for k in ("split_launches_by_family", "policy", "env_overrides_applied", "bottleneck_tails_fused")}
  d4 = [[d[:4].contiguous() for d in d8], [d[4:].contiguous() for d in d8]]
  out["one_b4_plain"] = run(e4[:1], d4[:1], [None], steps)
  out["two_b4_plain"] = run(e4, d4, [None, None], steps)
  cuts = {
      "lo_hi_128": (lambda i: i < 128, lambda i: i >= 128),                  # (bit i -> XCD i % 8: half of every XCD's CUs)
      "even_odd": (lambda i: i % 2 == 0, lambda i: i % 2 == 1),             # XCDs 0 2 4 6 / 1 3 5 7
      "xcd_lo_hi": (lambda i: i % 8 < 4, lambda i: i % 8 >= 4),             # XCDs 0-3 / 4-7
  }
  for name, (pa, pb) in cuts.items():
    sa, sb = masked_stream(mask_words(pa)), masked_stream(mask_words(pb))
    out["one_b4_half_" + name] = run(e4[:1], d4[:1], [sa], steps)
    for stag in (0.0, 6.0):
      out["two_b4_half_%s_stagger%g" % (name, stag)] = run(e4, d4, [sa, sb], steps, stag)
    # tail overlap off would need new handles; the tail stream is unmasked either way
    print(json.dumps(out), flush=True)
  out["one_b8_again"] = run([e8], [d8], [None], steps)
  print(json.dumps(out, indent=1))


if __name__ == "__main__":
  main()
