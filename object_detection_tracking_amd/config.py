"""Model configuration namespace for the detection hot path.

Mirrors the flat ``args`` namespace the reference builds in
``obj_detect_tracking.py:get_args`` (reference obj_detect_tracking.py:64-389):
the same attribute names with the same derived constants, so that the object
handed to :func:`object_detection_tracking_amd.models.get_model` can be either
the reference's own ``args`` or one made by :func:`make_config`.
"""
from __future__ import annotations

import math
from types import SimpleNamespace

import numpy as np

# class table of the 15-class ActEV model (reference class_ids.py:81-97)
ACTEV_CLASSES = [
    "BG", "Vehicle", "Person", "Parking_Meter", "Tree", "Skateboard",
    "Prop_Overshoulder", "Construction_Barrier", "Door", "Dumpster",
    "Push_Pulled_Object", "Construction_Vehicle", "Prop", "Bike", "Animal",
]


def make_config(**overrides):
  """Return a namespace with the reference's inference defaults (v3 model).

  Field names / values follow reference obj_detect_tracking.py:268-387.
  """
  c = SimpleNamespace()
  # flags (reference obj_detect_tracking.py:64-235 defaults, --version 3)
  c.version = 3
  c.num_class = 15
  c.im_batch_size = 1
  c.gpu = 1
  c.gpuid_start = 0
  c.max_size = 1920
  c.short_edge_size = 1080
  c.rpn_test_post_nms_topk = 1000
  c.threshold_conf = 0.0001
  c.model_path = None
  c.is_load_from_pb = False
  c.is_efficientdet = False
  c.is_coco_model = False
  c.use_partial_classes = False
  c.partial_classes = []
  c.add_mask = False
  c.use_gn = False
  c.use_se = False
  c.use_resnext = False
  c.use_deformable = False
  c.use_frcnn_class_agnostic = False
  c.use_conv_frcnn_head = False
  c.use_att_frcnn_head = False
  c.add_relation_nn = False
  c.use_dilations = True          # version 3 (obj_detect_tracking.py:270-271)
  # tracker flags
  c.min_confidence = 0.85
  c.min_detection_height = 0
  c.nms_max_overlap = 0.85
  c.max_iou_distance = 0.5
  c.max_cosine_distance = 0.5
  c.nn_budget = 5
  c.tracking_objs = "Person,Vehicle"
  c.frame_gap = 8
  c.conv_arith = None             # None / "default": split arithmetic where it pays | "f32": exact-f32 MFMA everywhere | "bf16x3"
  c.conv_split_family = "auto"    # "auto" (default): fp16x2 kernels, checked on the first forward(s) against a bf16x3-only twin handle;
                                  # the engine stays on bf16x3 when the pyramid / RPN tensors differ by more than conv_split_auto_tol
                                  # (models._Engine) | 0 / 2: fp16x2 kernels where eligible, UNGUARDED | 3: bf16x3 only | 1
  c.conv_split_auto_frames = 1    # "auto": forwards that are checked
  c.conv_split_auto_tol = 2e-5    # "auto": largest |difference| / |max| of a pyramid / RPN tensor that counts as f32 rounding
  c.keep_taps = False             # True: dedicated buffers for every stage tensor (engine.tap() of backbone stages; ~5x the activation memory)
  for k, v in overrides.items():
    setattr(c, k, v)
  return finalize_config(c)


def finalize_config(c):
  """Fill in the derived constants exactly as reference
  obj_detect_tracking.py:303-387 does (idempotent)."""
  d = c.__dict__
  d.setdefault("use_dilations", getattr(c, "version", 3) in (3, 4, 5))
  # versions 4-6: class-agnostic box regression (obj_detect_tracking.py:272-280)
  if getattr(c, "version", 3) in (4, 5, 6) and not d.get("_version_flags_applied", False):
    c.use_frcnn_class_agnostic = True
    if c.version in (4, 5):
      c.use_dilations = True
    if c.version == 6:
      c.use_se = True
    c._version_flags_applied = True
  c.is_train = False
  c.use_cpu_nms = False
  c.use_bg_score = False
  c.use_small_object_head = False
  c.no_obj_detect = False
  c.is_fpn = True
  c.rpn_min_size = 0
  c.rpn_proposal_nms_thres = 0.7
  c.anchor_strides = (4, 8, 16, 32, 64)
  c.fpn_resolution_requirement = float(c.anchor_strides[3])
  c.max_size = float(np.ceil(c.max_size / c.fpn_resolution_requirement) *
                     c.fpn_resolution_requirement)
  c.fpn_num_channel = 256
  c.fpn_frcnn_fc_head_dim = 1024
  d.setdefault("mrcnn_head_dim", 256)          # obj_detect_tracking.py:324
  d.setdefault("resnet_num_block", [3, 4, 23, 3])
  c.use_basic_block = False
  c.anchor_sizes = (32, 64, 128, 256, 512)
  c.anchor_ratios = (0.5, 1, 2)
  c.num_anchors = len(c.anchor_sizes) * len(c.anchor_ratios)
  # RPN decode clip uses max_size (obj_detect_tracking.py:372); the box head
  # uses decode_bbox_target's default log(1333/16) (nn.py:1518).
  c.bbox_decode_clip = float(np.log(c.max_size / 16.0))
  c.fastrcnn_bbox_reg_weights = np.array([10, 10, 5, 5], dtype="float32")
  c.rpn_test_pre_nms_topk = 6000   # dead in FPN mode (models.py:411-424)
  c.fastrcnn_nms_iou_thres = 0.5
  c.result_score_thres = getattr(c, "threshold_conf", 0.0001)
  d.setdefault("result_per_im", 100)           # obj_detect_tracking.py:375 (tests shrink it)
  if not hasattr(c, "classname2id"):
    names = ACTEV_CLASSES if c.num_class == 15 else \
        ["BG"] + ["class%d" % i for i in range(1, c.num_class)]
    c.classname2id = {n: i for i, n in enumerate(names)}
    c.classid2name = {i: n for i, n in enumerate(names)}
  return c


HEAD_DECODE_CLIP = float(math.log(1333 / 16.0))  # nn.py:1518 default
