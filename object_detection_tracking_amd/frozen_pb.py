"""Weights out of a frozen TensorFlow ``GraphDef`` (``.pb``) WITHOUT TensorFlow.

The reference's released models are frozen graphs (``--is_load_from_pb``, reference
obj_detect_tracking.py:83,244-245; models.py:198-263 ``Mask_RCNN_FPN_frozen`` imports the file
with ``tf.import_graph_def``).  ``graph_util.convert_variables_to_constants`` turns every variable
into a ``Const`` node that keeps the variable's name (``conv0/W``, ``group0/block0/conv1/bn/gamma``,
``.../mean/EMA`` ...), so the file is a weight container: this module walks the protobuf wire
format directly (GraphDef.node=1 -> NodeDef{name=1, op=2, attr=5{key=1, value=2}} ->
AttrValue.tensor=8 -> TensorProto{dtype=1, tensor_shape=2{dim=2{size=1}}, tensor_content=4,
float_val=5, double_val=6, half_val=13}) and returns {name: float32 array}.  The graph STRUCTURE is
not read from the file -- it is the reference's fixed architecture, rebuilt from the config as for
the ``.npz`` route; only ``Const`` payloads are used.

The field numbers are those of tensorflow/core/framework/{graph,node_def,attr_value,tensor,
tensor_shape}.proto; no real frozen model ships with the reference or is reachable offline, so the
reader is exercised against files produced by :func:`write_frozen_pb` (same wire format, written
by hand) and against GraphDefs serialised by Google's protobuf runtime from those message definitions
(tests/tf_protos.py, tests/test_drop_in.py: every payload form, op nodes and non-float constants in
between) -- the encoder is then the official one; a file written by TensorFlow itself has still not
been read.
"""
from __future__ import annotations

import struct

import numpy as np

DT_FLOAT, DT_DOUBLE, DT_INT32, DT_HALF = 1, 2, 3, 19


def _varint(buf, i):
  x = 0; s = 0
  while True:
    b = buf[i]; i += 1
    x |= (b & 0x7F) << s
    if not b & 0x80:
      return x, i
    s += 7


def _fields(buf):
  """Yield (field_number, wire_type, value) over one message; length-delimited values come back
  as memoryviews, fixed32/64 as raw bytes, varints as ints."""
  i, n = 0, len(buf)
  while i < n:
    key, i = _varint(buf, i)
    f, wt = key >> 3, key & 7
    if wt == 0:
      v, i = _varint(buf, i)
    elif wt == 1:
      v = bytes(buf[i:i + 8]); i += 8
    elif wt == 2:
      ln, i = _varint(buf, i)
      v = buf[i:i + ln]; i += ln
    elif wt == 5:
      v = bytes(buf[i:i + 4]); i += 4
    else:
      raise ValueError("unsupported protobuf wire type %d" % wt)
    yield f, wt, v


def _tensor(buf):
  dtype, shape, content = 0, [], None
  floats, doubles, halves = [], [], []
  for f, wt, v in _fields(buf):
    if f == 1:
      dtype = v
    elif f == 2:                                   # TensorShapeProto
      for f2, _, v2 in _fields(v):
        if f2 == 2:                                # Dim
          size = 0
          for f3, _, v3 in _fields(v2):
            if f3 == 1:
              size = v3 if v3 < (1 << 63) else v3 - (1 << 64)
          shape.append(size)
    elif f == 4:
      content = bytes(v)
    elif f == 5:                                   # float_val: packed or repeated fixed32
      floats.append(np.frombuffer(bytes(v), "<f4"))
    elif f == 6:
      doubles.append(np.frombuffer(bytes(v), "<f8"))
    elif f == 13:                                  # half_val: varints holding the 16 bit pattern
      if wt == 2:
        j, raw = 0, bytes(v)
        while j < len(raw):
          x, j = _varint(raw, j); halves.append(x)
      else:
        halves.append(v)
  if dtype not in (DT_FLOAT, DT_DOUBLE, DT_HALF):
    return None
  n = int(np.prod(shape)) if shape else 1
  np_dt = {DT_FLOAT: "<f4", DT_DOUBLE: "<f8", DT_HALF: "<f2"}[dtype]
  if content:
    a = np.frombuffer(content, np_dt)
  elif dtype == DT_FLOAT and floats:
    a = np.concatenate(floats)
  elif dtype == DT_DOUBLE and doubles:
    a = np.concatenate(doubles)
  elif dtype == DT_HALF and halves:
    a = np.asarray(halves, np.uint16).view("<f2")
  else:
    a = np.zeros((n,), np.float32)                 # all-default tensor
  if a.size == 1 and n > 1:
    a = np.full((n,), a[0])                        # TF stores a constant-filled tensor once
  if a.size != n:
    raise ValueError("tensor payload of %d values for shape %s" % (a.size, shape))
  return np.asarray(a, np.float32).reshape(shape)


def load_frozen_pb(path, min_rank=1):
  """{node name: float32 array} for every floating-point ``Const`` of rank >= ``min_rank``."""
  with open(path, "rb") as fh:
    buf = memoryview(fh.read())
  out = {}
  for f, wt, node in _fields(buf):
    if f != 1 or wt != 2:
      continue
    name, op, value = "", "", None
    for f2, _, v2 in _fields(node):
      if f2 == 1:
        name = bytes(v2).decode()
      elif f2 == 2:
        op = bytes(v2).decode()
      elif f2 == 5:                                # attr map entry
        key, av = "", None
        for f3, _, v3 in _fields(v2):
          if f3 == 1:
            key = bytes(v3).decode()
          elif f3 == 2:
            av = v3
        if key == "value" and av is not None:
          for f4, _, v4 in _fields(av):
            if f4 == 8:
              value = v4
    if op == "Const" and value is not None:
      t = _tensor(value)
      if t is not None and t.ndim >= min_rank:
        out[name] = t
  if not out:
    raise ValueError("%s: no floating-point Const nodes found (not a frozen GraphDef?)" % path)
  return out


# ---- writer (tests / tooling): the same wire format by hand ---------------------------------------
def _enc_varint(x):
  out = bytearray()
  while True:
    b = x & 0x7F; x >>= 7
    out.append(b | (0x80 if x else 0))
    if not x:
      return bytes(out)


def _ld(field, payload):
  return _enc_varint((field << 3) | 2) + _enc_varint(len(payload)) + payload


def _vi(field, x):
  return _enc_varint(field << 3) + _enc_varint(x)


def _node(name, op, attrs=(), inputs=()):
  body = _ld(1, name.encode()) + _ld(2, op.encode())
  for i in inputs:
    body += _ld(3, i.encode())
  for k, v in attrs:
    body += _ld(5, _ld(1, k.encode()) + _ld(2, v))
  return _ld(1, body)


def _tensor_proto(a, use_float_val=False, half=False):
  a = np.asarray(a)
  shape = b"".join(_ld(2, _vi(1, int(d))) for d in a.shape)
  if half:
    return _vi(1, DT_HALF) + _ld(2, shape) + _ld(4, a.astype("<f2").tobytes())
  if use_float_val:
    return _vi(1, DT_FLOAT) + _ld(2, shape) + _ld(5, a.astype("<f4").tobytes())
  return _vi(1, DT_FLOAT) + _ld(2, shape) + _ld(4, a.astype("<f4").tobytes())


def write_frozen_pb(path, weights, float_val_names=(), half_names=()):
  """Serialise {name: array} as a frozen-graph-shaped GraphDef: a uint8 ``image`` Placeholder,
  one ``Const`` per tensor (+ an ``Identity`` reader, as freezing leaves them) and the output
  identities the reference looks up by name (models.py:219-238)."""
  blob = _node("image", "Placeholder", [("dtype", _vi(6, 4))])
  for name, a in weights.items():
    tp = _tensor_proto(a, use_float_val=name in float_val_names, half=name in half_names)
    blob += _node(name, "Const", [("dtype", _vi(6, DT_HALF if name in half_names else DT_FLOAT)),
                                  ("value", _ld(8, tp))])
    blob += _node(name + "/read", "Identity", [("T", _vi(6, DT_FLOAT))], inputs=[name])
  # a non-float Const and a scalar Const, as real graphs contain (must be skipped)
  blob += _node("anchor_count", "Const", [("dtype", _vi(6, DT_INT32)),
                                          ("value", _ld(8, _vi(1, DT_INT32) + _ld(7, _enc_varint(15))))])
  blob += _node("bn_epsilon", "Const", [("dtype", _vi(6, DT_FLOAT)),
                                        ("value", _ld(8, _vi(1, DT_FLOAT) + _ld(2, b"") +
                                                      _ld(5, struct.pack("<f", 1e-5))))])
  for out in ("final_boxes", "final_labels", "final_probs", "fpn_box_feat"):
    blob += _node(out, "Identity", [("T", _vi(6, DT_FLOAT))])
  blob += _ld(4, _vi(1, 134))                      # VersionDef.producer
  with open(path, "wb") as fh:
    fh.write(blob)
