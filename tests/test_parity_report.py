"""Parity at BASELINE's full sizes, MEASURED and written down (the e2e tests passed with a
mismatch budget whose actual use was reported nowhere).

test_parity_report_1080p runs configs #2 (1920x1080, b=1, single graph) and #3 (b=8, batched graph) in the three
arithmetic modes of the conv path (exact-f32 MFMA; the split kernels the bench runs -- fp16x2 where a layer is eligible,
bf16x3 elsewhere; bf16x3 only) against the oracle (a CPU restatement of the reference's TF graph -- parity unpinned against
TensorFlow itself, oracle/__init__.py) and writes gpurun_out/r03_parity.json: per-stage max relative error, proposal /
detection mismatch counts at the test tolerance, the largest box / score difference among the matched ones, label and
valid-count equality.  The budgets asserted below ARE the measured values (see profiles/r03_parity.json, the
tracked copy of a run on the MI355X): anything worse is a regression.

The split kernel is also checked directly at the model's dominant shapes against float64.
"""
import json
import os

import numpy as np
import pytest

from common import weights_for
from object_detection_tracking_amd import models, ops
from common import make_config
from object_detection_tracking_amd.weights import synthetic_frames
from oracle.graph import OracleModel

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = np.float32
TRUNK = ["conv0", "pool0", "c2", "c3", "c4", "c5"]


def _rel(a, b):
  return float(np.abs(a - b).max() / max(1e-6, np.abs(b).max()))


def _stage_errors(e, ref):
  out = {}
  for name in TRUNK:
    out[name] = _rel(e.tap(name).transpose(0, 3, 1, 2), ref[name])
  for l in range(2, 7):
    r = ref["p%d" % l]
    out["p%d" % l] = _rel(e.tap("p%d" % l).transpose(0, 3, 1, 2)[:, :, :r.shape[2], :r.shape[3]], r)
    rp = e.tap("rpn%d" % l)
    out["rpn%d_logits" % l] = _rel(rp[..., :3], ref["rpn_logits%d" % l])
    out["rpn%d_deltas" % l] = _rel(rp[..., 3:15].reshape(rp.shape[:3] + (3, 4)), ref["rpn_deltas%d" % l])
  return out


def _set_diff(b1, l1, p1, b2, l2, p2, tol_box, tol_prob):
  """One-to-one nearest match (same label); returns unmatched counts at the tolerances and the largest box / score
  difference among the matched pairs."""
  used = np.zeros(len(b2), bool)
  miss, dbox, dprob = 0, 0.0, 0.0
  for i in range(len(b1)):
    d = np.abs(b2 - b1[i]).max(1) if len(b2) else np.zeros(0)
    ok = (~used) & (l2 == l1[i]) & (d <= tol_box) & (np.abs(p2 - p1[i]) <= tol_prob)
    j = np.where(ok)[0]
    if j.size:
      k = j[np.argmin(d[j])]
      used[k] = True
      dbox = max(dbox, float(d[k])); dprob = max(dprob, float(abs(p2[k] - p1[i])))
    else:
      miss += 1
  return miss, int((~used).sum()), dbox, dprob


def _measure(lib, cfg, B, H, W, ref, multi):
  w = weights_for(cfg)
  fr = synthetic_frames(B, H, W)
  import copy
  cfg = copy.copy(cfg)
  cfg.keep_taps = True                    # stage errors are read through the taps (the arena handle is compared bit for bit in test_e2e.py)
  m = models.get_model(cfg, 0, weights=w, lib=lib, is_multi=multi)
  try:
    side = max(H, W)
    tol_box = 1e-3 * side / 128           # the north_star's 1e-3 is quoted at O(100) px coordinates (test_e2e.py)
    rec = {}
    if multi:
      boxes, labels, probs, valid, feats = m.predict_batch(fr)
      e = m.engine(B, H, W)
      rec["valid_equal"] = bool(np.array_equal(valid, ref["final_valid_indices"]))
      rec["valid"] = [int(v) for v in valid]
      miss = extra = 0; dbox = dprob = 0.0; lab_eq = True
      for b in range(B):
        v, rv = int(valid[b]), int(ref["final_valid_indices"][b])
        mi, ex, db, dp = _set_diff(boxes[b, :v], labels[b, :v], probs[b, :v], ref["final_boxes"][b, :rv],
                                   ref["final_labels"][b, :rv], ref["final_probs"][b, :rv], tol_box, 1e-4)
        miss += mi; extra += ex; dbox = max(dbox, db); dprob = max(dprob, dp)
        lab_eq = lab_eq and v == rv and bool(np.array_equal(labels[b, :v], ref["final_labels"][b, :rv]))
      ndet = int(valid.sum())
      nprop = e.tap("nproposals").reshape(-1).astype(int)
      props = e.tap("proposals")
      pm = pe = 0; pd = 0.0
      for b in range(B):
        rp = ref["proposals"][ref["proposals"][:, 0] == b][:, 1:]       # oracle rows are (image, x1, y1, x2, y2)
        n = int(nprop[b])
        z = np.zeros(max(n, len(rp)))
        a, c, d, _ = _set_diff(props[0, b, :n], z[:n], z[:n], rp, z[:len(rp)], z[:len(rp)], tol_box, 1)
        pm += a; pe += c; pd = max(pd, d)
      rec["proposals"] = {"count": int(nprop.sum()), "oracle_count": int(len(ref["proposals"])), "unmatched_ours": pm, "unmatched_oracle": pe, "max_box_diff_px": pd}
    else:
      boxes, labels, probs, feats = m.predict(fr[0])
      e = m.engine(1, H, W)
      miss, extra, dbox, dprob = _set_diff(boxes, labels, probs, ref["final_boxes"], ref["final_labels"],
                                          ref["final_probs"], tol_box, 1e-4)
      lab_eq = len(labels) == len(ref["final_labels"]) and bool(np.array_equal(labels, ref["final_labels"]))
      ndet = len(boxes)
      n = int(e.tap("nproposals")[0])
      z = np.zeros(max(n, len(ref["proposals"])))
      a, c, d, _ = _set_diff(e.tap("proposals")[0, 0, :n], z[:n], z[:n], ref["proposals"], z[:len(ref["proposals"])],
                             z[:len(ref["proposals"])], tol_box, 1)
      rec["proposals"] = {"count": n, "oracle_count": int(len(ref["proposals"])), "unmatched_ours": a,
                          "unmatched_oracle": c, "max_box_diff_px": d}
    rec["stage_max_rel_err"] = _stage_errors(e, ref)
    rec["detections"] = {"count": ndet, "unmatched_ours": miss, "unmatched_oracle": extra, "max_box_diff_px": dbox,
                         "max_box_diff_rel_to_side": dbox / side, "max_prob_diff": dprob, "labels_equal_in_order": lab_eq}
    rec["box_tolerance_px"] = tol_box
    rec["split_conv_launches"] = sum(1 for nm, _, _, _ in e.profile_layers() if nm.endswith("[bf16x3]") or nm.endswith("[fp16x2]"))
    rec["fp16x2_conv_launches"] = sum(1 for nm, _, _, _ in e.profile_layers() if nm.endswith("[fp16x2]"))
    if rec["detections"]["unmatched_ours"] + rec["detections"]["unmatched_oracle"] == 0 and lab_eq:
      rec["fpn_box_feat_max_rel_err"] = _rel(feats, ref["fpn_box_feat"])
    return rec
  finally:
    m.close()


# measured on the MI355X (profiles/r03_parity.json); asserted with no slack on the counts
# measured: stage max 2.0e-6 .. 3.5e-6 (fp16x2 the lowest), proposals 300/300 and 2400/2400 matched (max 7.3e-4 px), detections
# 100/100 and 800/800 matched with labels equal in order (max 3.7e-4 px = 1.9e-7 of the frame side, max score diff 7.7e-6)
MODES = (("exact_f32", {"ODT_CONV_SPLIT": "0"}), ("split_default_fp16x2_bf16x3", {"ODT_CONV_SPLIT": "1"}),
         ("split_bf16x3_only", {"ODT_CONV_SPLIT": "1", "ODT_CONV_SPLIT_PIPE": "3"}))
BUDGET = dict(stage=1e-5, det_unmatched=0, prop_unmatched=0, box_px=1e-3)


@pytest.mark.gpu
def test_parity_report_1080p(hip_lib, monkeypatch):
  report = {"oracle": "oracle/graph.py: CPU restatement of the reference's TF graph (torch-CPU fp32 + numpy); "
                      "parity unpinned against TensorFlow itself",
            "inputs": "synthetic_frames(seed 0) 1920x1080 uint8, synthetic_weights(seed 0), K = 300, 15 classes",
            "configs": {}}
  failures = []
  for cname, B, multi in (("config2_b1_1080p", 1, False), ("config3_b8_1080p", 8, True)):
    cfg = make_config(rpn_test_post_nms_topk=300, im_batch_size=B)
    fr = synthetic_frames(B, 1080, 1920)
    om = OracleModel(cfg, weights_for(cfg))
    ref = om.forward_multi(fr) if multi else om.forward(fr[0])
    report["configs"][cname] = {}
    for mode, env in MODES:
      monkeypatch.delenv("ODT_CONV_SPLIT_PIPE", raising=False)
      for k, v in env.items():
        monkeypatch.setenv(k, v)
      rec = _measure(hip_lib, cfg, B, 1080, 1920, ref, multi)
      report["configs"][cname][mode] = rec
      bud = BUDGET
      worst = max(rec["stage_max_rel_err"].values())
      d, p = rec["detections"], rec["proposals"]
      if worst > bud["stage"]: failures.append((cname, mode, "stage", worst))
      if d["unmatched_ours"] + d["unmatched_oracle"] > bud["det_unmatched"]: failures.append((cname, mode, "detections", d))
      if p["unmatched_ours"] + p["unmatched_oracle"] > bud["prop_unmatched"]: failures.append((cname, mode, "proposals", p))
      if max(d["max_box_diff_px"], p["max_box_diff_px"]) > bud["box_px"]: failures.append((cname, mode, "box px", d["max_box_diff_px"], p["max_box_diff_px"]))
      if not d["labels_equal_in_order"]: failures.append((cname, mode, "label order", d))
      if multi and not rec["valid_equal"]: failures.append((cname, mode, "valid counts", rec["valid"]))
      if mode != "exact_f32" and rec["split_conv_launches"] == 0: failures.append((cname, mode, "split path not taken", 0))
      if mode == "split_default_fp16x2_bf16x3" and rec["fp16x2_conv_launches"] == 0: failures.append((cname, mode, "fp16x2 kernels not taken", 0))
      if mode == "split_bf16x3_only" and rec["fp16x2_conv_launches"] != 0: failures.append((cname, mode, "fp16x2 kernels taken", rec["fp16x2_conv_launches"]))
  os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
  with open(os.path.join(ROOT, "gpurun_out", "r03_parity.json"), "w") as fh:
    json.dump(report, fh, indent=1)
  assert not failures, failures


# ---- the split kernels at the model's dominant shapes, directly, against float64 -------------------------------
def _sampled_conv_errors(lib, B, H, W, Cin, Cout, k, res, rng, monkeypatch, pipe, nsample=3000):
  """max |y - f64| / (sum |a||w| + |bias| + |residual|) over sampled outputs, for the split kernel family `pipe` and
  for the exact-f32 MFMA kernel on the same data."""
  x = rng.standard_normal((B, H, W, Cin)).astype(F)
  x *= (rng.random((B, H, W, Cin)) > 0.4)                     # post-ReLU-like sparsity
  w = (rng.standard_normal((k, k, Cin, Cout)) / np.sqrt(k * k * Cin)).astype(F)
  b = rng.standard_normal(Cout).astype(F)
  r = rng.standard_normal((B, H, W, Cout)).astype(F) if res else None
  M = B * H * W
  # sampled output pixels: the first / last rows of the tensor, tile borders (multiples of 256 +- 1), random ones
  idx = np.unique(np.clip(np.concatenate([np.arange(0, 40), np.arange(M - 40, M), (np.arange(1, 60) * 256)[:, None].repeat(3, 1).reshape(-1)
                                          + np.tile([-1, 0, 1], 59), rng.integers(0, M, nsample)]), 0, M - 1))
  bi, hi, wi = idx // (H * W), (idx // W) % H, idx % W
  xp = np.pad(x, ((0, 0), (k // 2, k // 2), (k // 2, k // 2), (0, 0)))
  patches = np.stack([xp[bi, hi + dy, wi + dx] for dy in range(k) for dx in range(k)], 1).reshape(len(idx), -1).astype(np.float64)
  wm = w.reshape(-1, Cout).astype(np.float64)
  ref = patches @ wm + b
  mag = np.abs(patches) @ np.abs(wm) + np.abs(b)
  if res:
    ref += r.reshape(M, Cout)[idx]; mag += np.abs(r.reshape(M, Cout)[idx])
  errs = []
  for split in ("1", "0"):
    monkeypatch.setenv("ODT_CONV_SPLIT", split)
    monkeypatch.setenv("ODT_CONV_SPLIT_PIPE", pipe)
    y = ops.conv2d(x, w, b, 1, 1, k // 2, k // 2, (H, W), res=r, res_mode=1 if res else 0, relu=False, lib=lib)
    errs.append(float((np.abs(y.reshape(M, Cout)[idx] - ref) / mag).max()))
  return errs[0], errs[1]


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["3", "1", "2"])
def test_split_kernel_at_model_shapes_vs_f64(hip_lib, mode, monkeypatch):
  """res4 conv2 (M = 65 280, N = 256, K = 2304), the P2-level 3x3 (M = 1 044 480, N = 256, K = 2304: the FPN post-hoc
  and RPN convs) and res4 conv3 + residual (M = 65 280, N = 1024, K = 256): error of every sampled output against
  float64, relative to sum |a||w| + |bias| + |residual|.  An f32 dot product of this length sits at ~1e-7; the maximum
  over ~3 million sampled outputs measured 1.6e-7 (K = 2304) .. 6.1e-7 (K = 256 + bias + residual).  The bound is the
  exact-f32 MFMA kernel's own error on the same data (x 1.5, + 2^-23 for the three dropped piece products) and 1e-6."""
  rng = np.random.default_rng(20)
  shapes = [(8, 68, 120, 256, 256, 3, False), (8, 68, 120, 256, 1024, 1, True)]
  if mode in ("3", "2"):           # ("2": the fp16x2 kernels -- the stand-alone call records the input's range itself)
    shapes.append((8, 272, 480, 256, 256, 3, False))
  for sh in shapes:
    esp, e32 = _sampled_conv_errors(hip_lib, *sh, rng, monkeypatch, mode)
    # (the one-stage loop starts its accumulators at the residual, so its products are summed on top of
    # an O(1) value: same absolute bound, no relative one)
    assert esp < 1e-6 and (mode == "1" or esp <= 1.5 * e32 + 1.2e-7), (sh, esp, e32)


@pytest.mark.gpu
def test_split_kernel_stage_entry_two_sources_at_model_shape_vs_f64(hip_lib, monkeypatch):
  """res4 block0: conv3(t2: 256 ch) + convshortcut(x: 512 ch at stride 2) as one K-concatenated GEMM, M = 65 280,
  N = 1024, K = 768."""
  monkeypatch.setenv("ODT_CONV_SPLIT", "1")
  rng = np.random.default_rng(21)
  B, Ho, Wo = 8, 68, 120
  a = rng.standard_normal((B, Ho, Wo, 256)).astype(F)
  b2 = rng.standard_normal((B, 2 * Ho, 2 * Wo, 512)).astype(F)
  wa = (rng.standard_normal((256, 1024)) / 16).astype(F)
  wb = (rng.standard_normal((512, 1024)) / 23).astype(F)
  bias = rng.standard_normal(1024).astype(F)
  y = ops.conv2d_cat(a, b2, wa, wb, bias, stride_b=2, relu=False, lib=hip_lib)
  M = B * Ho * Wo
  idx = np.unique(np.concatenate([np.arange(0, 64), np.arange(M - 64, M), rng.integers(0, M, 4000)]))
  A = np.concatenate([a.reshape(M, 256)[idx], b2[:, ::2, ::2].reshape(M, 512)[idx]], 1).astype(np.float64)
  Wm = np.concatenate([wa, wb], 0).astype(np.float64)
  ref = A @ Wm + bias
  mag = np.abs(A) @ np.abs(Wm) + np.abs(bias)
  assert float((np.abs(y.reshape(M, 1024)[idx] - ref) / mag).max()) < 1e-6
