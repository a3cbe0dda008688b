"""ctypes binding of the C ABI declared in include/odt.h.

The product path is libodt_hip.so (hand-written HIP for gfx950) and nothing
else: :func:`get_lib` raises if the library is missing or no GPU is visible --
there is no CPU fallback.  (Tests may bind another build of the *same* sources,
e.g. the HIP-on-CPU simulator under tests/emu/, by constructing
:class:`OdtLib` with an explicit path; the package itself never does.)
"""
from __future__ import annotations

import ctypes as C
import os

# (effective when this package is imported before anything has loaded libamdhip64 -- torch does: see INTEGRATION.md section 3,
# "several streams per GPU": eight hardware queues per stream priority instead of four; an explicit setting wins)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_HIP_PATH = os.path.join(HERE, "libodt_hip.so")

c_float_p = C.POINTER(C.c_float)
c_int_p = C.POINTER(C.c_int32)
c_i64_p = C.POINTER(C.c_int64)
c_double_p = C.POINTER(C.c_double)

ODT_DTYPE_U8, ODT_DTYPE_F32 = 0, 1
ODT_ARITH_DEFAULT, ODT_ARITH_F32, ODT_ARITH_BF16X3 = 0, 1, 2
ODT_GRAPH_SINGLE, ODT_GRAPH_MULTI, ODT_GRAPH_EFFNET = 0, 1, 2
RPN_CH = 16


class OdtConfig(C.Structure):
  _fields_ = [
      ("graph", C.c_int32), ("batch", C.c_int32), ("height", C.c_int32),
      ("width", C.c_int32), ("num_class", C.c_int32),
      ("num_blocks", C.c_int32 * 4), ("use_dilations", C.c_int32),
      ("fpn_channels", C.c_int32), ("head_dim", C.c_int32),
      ("rpn_topk", C.c_int32), ("result_per_im", C.c_int32),
      ("anchor_field", C.c_int32), ("rpn_nms_thresh", C.c_float),
      ("rpn_decode_clip", C.c_float), ("head_decode_clip", C.c_float),
      ("bbox_reg_weights", C.c_float * 4), ("result_score_thresh", C.c_float),
      ("head_nms_thresh", C.c_float), ("add_mask", C.c_int32), ("mask_dim", C.c_int32),
      ("eff_backbone", C.c_int32), ("eff_det", C.c_int32), ("eff_topk", C.c_int32),
      ("eff_image_scale", C.c_float), ("conv_arith", C.c_int32), ("conv_split_family", C.c_int32),
      ("keep_taps", C.c_int32),
      ("tail_overlap", C.c_int32),
  ]


class OdtOutputs(C.Structure):
  _fields_ = [("boxes", c_float_p), ("probs", c_float_p), ("labels", c_int_p),
              ("valid", c_int_p), ("feats", c_float_p), ("pooled", c_float_p),
              ("masks", c_float_p)]


def fptr(a):
  return a.ctypes.data_as(c_float_p)


def iptr(a):
  return a.ctypes.data_as(c_int_p)


def f32(a):
  return np.ascontiguousarray(a, dtype=np.float32)


def i32(a):
  return np.ascontiguousarray(a, dtype=np.int32)


class OdtError(RuntimeError):
  pass


class OdtLib(object):
  """Typed view of one build of the C ABI."""

  SYMBOLS = [
      "odt_last_error", "odt_device_count", "odt_create", "odt_destroy",
      "odt_load_tensor", "odt_finalize_weights", "odt_forward",
      "odt_forward_async", "odt_synchronize", "odt_read_outputs", "odt_describe", "odt_range_health", "odt_submit", "odt_submit_ex", "odt_collect",
      "odt_ingest_buffer", "odt_set_source_size", "odt_tap", "odt_profile_enable",
      "odt_profile_read", "odt_profile_layer", "odt_probe_mfma_bf16", "odt_nn_cosine", "odt_op_conv2d", "odt_op_conv2d_cat",
      "odt_op_bottleneck_tail", "odt_op_stem", "odt_op_preprocess",
      "odt_op_maxpool", "odt_op_topk", "odt_op_nms", "odt_op_proposals",
      "odt_op_roi_align", "odt_op_detections", "odt_op_class_nms", "odt_tracker_create", "odt_tracker_destroy",
      "odt_tracker_predict", "odt_tracker_update", "odt_tracker_tracks", "odt_lsap", "odt_tracker_nms",
      "odt_tmot_create", "odt_tmot_destroy", "odt_tmot_reset", "odt_tmot_update", "odt_tmot_tracks",
  ]

  def __init__(self, path):
    if not os.path.exists(path):
      raise OdtError("native library not found: %s (build it with "
                     "`python -m object_detection_tracking_amd.build`)" % path)
    self.path = path
    self.dll = C.CDLL(path)
    d = self.dll
    for s in self.SYMBOLS:
      if not hasattr(d, s):
        raise OdtError("%s does not export %s" % (path, s))
    d.odt_last_error.restype = C.c_char_p
    d.odt_create.argtypes = [C.POINTER(OdtConfig), C.c_int, C.POINTER(C.c_void_p)]
    d.odt_destroy.argtypes = [C.c_void_p]
    d.odt_load_tensor.argtypes = [C.c_void_p, C.c_char_p, c_float_p, c_i64_p, C.c_int]
    d.odt_finalize_weights.argtypes = [C.c_void_p]
    d.odt_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                              C.POINTER(OdtOutputs)]
    d.odt_forward_async.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    d.odt_synchronize.argtypes = [C.c_void_p]
    d.odt_read_outputs.argtypes = [C.c_void_p, C.POINTER(OdtOutputs)]
    d.odt_submit.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    d.odt_describe.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
    d.odt_range_health.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.c_char_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_longlong)]
    d.odt_submit_ex.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int)]
    d.odt_collect.argtypes = [C.c_void_p, C.c_int, C.POINTER(OdtOutputs)]
    d.odt_set_source_size.argtypes = [C.c_void_p, C.c_int, C.c_int]
    d.odt_ingest_buffer.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p),
                                    C.POINTER(C.c_size_t)]
    d.odt_tap.argtypes = [C.c_void_p, C.c_char_p, c_float_p, C.c_size_t, c_i64_p,
                          C.POINTER(C.c_int)]
    d.odt_profile_enable.argtypes = [C.c_void_p, C.c_int]
    d.odt_profile_read.argtypes = [C.c_void_p, c_double_p, c_double_p,
                                   C.POINTER(C.c_int), c_double_p]
    d.odt_profile_layer.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_int, c_double_p,
                                    c_double_p, c_i64_p, C.POINTER(C.c_int)]
    d.odt_probe_mfma_bf16.argtypes = [C.c_int, C.c_double, C.c_double, C.c_int, c_double_p, c_double_p, c_double_p,
                                      C.POINTER(C.c_int)]
    d.odt_nn_cosine.argtypes = [C.c_int, c_float_p, c_int_p, C.c_int, c_float_p, C.c_int,
                                C.c_int, c_double_p]
    d.odt_op_conv2d.argtypes = [C.c_int, c_float_p] + [C.c_int] * 4 + [c_float_p, c_float_p] + \
        [C.c_int] * 11 + [c_float_p, C.c_int, C.c_int, c_float_p]
    d.odt_op_conv2d_cat.argtypes = [C.c_int, c_float_p] + [C.c_int] * 4 + [c_float_p] + [C.c_int] * 4 + \
        [c_float_p, c_float_p, c_float_p, C.c_int, C.c_int, c_float_p]
    d.odt_op_bottleneck_tail.argtypes = [C.c_int, c_float_p] + [C.c_int] * 4 + [c_float_p, c_float_p, C.c_int, c_float_p, c_float_p,
                                         C.c_int, c_float_p, C.c_int, C.c_int, c_float_p]
    d.odt_op_stem.argtypes = [C.c_int, c_float_p] + [C.c_int] * 3 + [c_float_p, c_float_p, C.c_int, C.c_int, c_float_p]
    d.odt_op_preprocess.argtypes = [C.c_int, C.c_void_p] + [C.c_int] * 8 + [c_float_p]
    d.odt_op_maxpool.argtypes = [C.c_int, c_float_p] + [C.c_int] * 4 + [c_float_p]
    d.odt_op_topk.argtypes = [C.c_int, c_float_p, C.c_int, C.c_int, c_int_p]
    d.odt_op_nms.argtypes = [C.c_int, c_float_p, c_float_p, C.c_int, C.c_int, C.c_float,
                             c_int_p, C.POINTER(C.c_int)]
    d.odt_op_proposals.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, c_int_p, c_int_p, c_int_p,
                                   C.POINTER(c_float_p), C.POINTER(c_float_p), C.c_int, C.c_int,
                                   C.c_int, C.c_float, C.c_float, c_float_p, c_int_p]
    d.odt_op_roi_align.argtypes = [C.c_int, C.c_int, C.c_int, c_int_p, c_int_p,
                                   C.POINTER(c_float_p), c_float_p, c_float_p, c_int_p, C.c_int,
                                   c_float_p, c_float_p]
    d.odt_tracker_create.argtypes = [C.c_double, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int,
                                     C.POINTER(C.c_void_p)]
    d.odt_tracker_destroy.argtypes = [C.c_void_p]
    d.odt_tracker_predict.argtypes = [C.c_void_p]
    d.odt_tracker_update.argtypes = [C.c_void_p, c_double_p, c_double_p, c_float_p, C.c_int, C.c_int]
    d.odt_tracker_tracks.argtypes = [C.c_void_p, C.c_int, c_int_p, c_int_p, c_int_p, c_int_p, c_int_p,
                                     c_double_p, c_double_p, C.POINTER(C.c_int)]
    d.odt_lsap.argtypes = [c_double_p, C.c_int, C.c_int, c_int_p, c_int_p, C.POINTER(C.c_int)]
    d.odt_tracker_nms.argtypes = [c_double_p, c_double_p, c_int_p, C.c_int, C.c_double, c_int_p, C.POINTER(C.c_int)]
    d.odt_tmot_create.argtypes = [C.c_double] * 8 + [C.POINTER(C.c_void_p)]
    d.odt_tmot_destroy.argtypes = [C.c_void_p]
    d.odt_tmot_reset.argtypes = [C.c_void_p]
    d.odt_tmot_update.argtypes = [C.c_void_p, c_double_p, c_double_p, c_float_p, C.c_int, C.c_int,
                                  C.POINTER(C.c_int), C.POINTER(C.c_int)]
    d.odt_tmot_tracks.argtypes = [C.c_void_p, C.c_int, C.c_int, c_int_p, c_int_p, c_int_p, c_double_p,
                                  c_double_p, c_double_p, c_double_p, c_int_p, c_int_p, c_int_p,
                                  C.POINTER(C.c_int)]
    d.odt_op_detections.argtypes = [C.c_int] * 5 + [c_float_p, c_float_p, c_float_p, c_int_p,
                                                    C.c_int, C.c_int, c_float_p, C.c_float,
                                                    C.c_float, C.c_float, C.c_int, c_float_p,
                                                    c_float_p, c_int_p, c_int_p]
    d.odt_op_class_nms.argtypes = [C.c_int] * 5 + [c_float_p, c_float_p, c_int_p, C.c_float, C.c_float, C.c_int,
                                                   c_float_p, c_float_p, c_int_p, c_int_p]

  def check(self, rc):
    if rc != 0:
      msg = self.dll.odt_last_error()
      raise OdtError(msg.decode("utf-8", "replace") if msg else "odt error %d" % rc)

  def device_count(self):
    n = C.c_int(0)
    self.check(self.dll.odt_device_count(C.byref(n)))
    return n.value


_LIB = None


def get_lib():
  """The product library (HIP, gfx950).  Fails loudly; never falls back."""
  global _LIB
  if _LIB is None:
    lib = OdtLib(LIB_HIP_PATH)
    if lib.device_count() < 1:
      raise OdtError("libodt_hip.so loaded but no HIP device is visible")
    _LIB = lib
  return _LIB
