"""EfficientNet backbone on the HIP kernels (the part of the EfficientDet path that is built):
``EfficientNetBackbone(name, weights).features(frames)`` -> {level: [B,h,w,C] float32} for the
reduction_1..5 endpoints (reference efficientdet/efficientdet_arch.py:396-437 ``build_backbone``:
levels 3, 4, 5 feed the BiFPN)."""
import ctypes as C

import numpy as np

from .. import _lib
from .._lib import ODT_DTYPE_F32, ODT_DTYPE_U8, ODT_GRAPH_EFFNET, OdtConfig, c_i64_p, f32, fptr
from .arch import backbone_spec


class EfficientNetBackbone(object):

  def __init__(self, name, weights, batch, height, width, device=0, lib=None, det=None, num_classes=90,
               topk=5000, score_thresh=0.0, per_im=100, image_scale=1.0, keep_taps=True):
    self.lib = lib if lib is not None else _lib.get_lib()
    self.name, self.batch, self.height, self.width = name, batch, height, width
    self.src_height, self.src_width = height, width
    self.spec = backbone_spec(name)
    c = OdtConfig()
    c.graph = ODT_GRAPH_EFFNET; c.batch = batch; c.height = height; c.width = width
    c.eff_backbone = int(name[-1])
    c.eff_det = -1 if det is None else int(det[-1])      # "efficientdet-dN"
    c.num_class = num_classes; c.eff_topk = topk; c.result_score_thresh = score_thresh
    c.result_per_im = per_im; c.eff_image_scale = image_scale; c.head_nms_thresh = 0.5
    c.keep_taps = int(bool(keep_taps))     # (features() / tap() read stage tensors: the stand-alone backbone keeps them by default)
    self.h = C.c_void_p()
    self.lib.check(self.lib.dll.odt_create(C.byref(c), device, C.byref(self.h)))
    try:
      for k, a in weights.items():
        a = f32(a)
        shape = (C.c_int64 * a.ndim)(*a.shape)
        self.lib.check(self.lib.dll.odt_load_tensor(self.h, k.encode(), fptr(a), C.cast(shape, c_i64_p), a.ndim))
      self.lib.check(self.lib.dll.odt_finalize_weights(self.h))
    except Exception:
      self.lib.dll.odt_destroy(self.h); self.h = None
      raise

  def set_source_size(self, src_height, src_width):
    """Frames of [B, src_height, src_width, 3]; the reference's input scaling (dataloader.py:100-123)
    runs on the device and the output boxes are multiplied by image_scale_to_original."""
    self.lib.check(self.lib.dll.odt_set_source_size(self.h, int(src_height), int(src_width)))
    self.src_height, self.src_width = int(src_height), int(src_width)

  def close(self):
    if self.h is not None:
      self.lib.dll.odt_destroy(self.h); self.h = None

  def __del__(self):
    try:
      self.close()
    except Exception:
      pass

  def forward_async(self, frames):
    fr = np.ascontiguousarray(frames)
    dt = ODT_DTYPE_U8 if fr.dtype == np.uint8 else ODT_DTYPE_F32
    if dt == ODT_DTYPE_F32:
      fr = np.ascontiguousarray(fr, np.float32)
    assert fr.shape == (self.batch, self.src_height, self.src_width, 3), fr.shape
    self._keep = fr
    self.lib.check(self.lib.dll.odt_forward_async(self.h, fr.ctypes.data_as(C.c_void_p), dt, 0, None))

  def synchronize(self):
    self.lib.check(self.lib.dll.odt_synchronize(self.h))

  def describe(self):
    """What the handle runs (odt_describe): kernel families, launches, memory."""
    import json
    buf = C.create_string_buffer(16384)
    self.lib.check(self.lib.dll.odt_describe(self.h, buf, 16384))
    return json.loads(buf.value.decode())

  def tap(self, name):
    """Stage tensor in the device layout (NHWC, channel stride padded to 32), as numpy."""
    shape = (C.c_int64 * 4)(); rank = C.c_int()
    self.lib.check(self.lib.dll.odt_tap(self.h, name.encode(), None, 0, C.cast(shape, c_i64_p), C.byref(rank)))
    out = np.zeros([int(shape[i]) for i in range(rank.value)], np.float32)
    self.lib.check(self.lib.dll.odt_tap(self.h, name.encode(), fptr(out), out.size, C.cast(shape, c_i64_p),
                                        C.byref(rank)))
    return out

  def features(self, frames):
    """{level: NHWC float32 [B,h,w,C]} of reduction_1..5 (pad channels stripped)."""
    self.forward_async(frames); self.synchronize()
    out = {}
    for b in self.spec["blocks"]:
      if b["reduction"]:
        out[b["reduction"]] = np.ascontiguousarray(self.tap("reduction_%d" % b["reduction"])[..., :b["cout"]])
    return out
