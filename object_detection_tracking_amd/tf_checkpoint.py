"""Variables out of a TensorFlow V2 checkpoint (``model-NNN.index`` + ``.data-00000-of-00001``)
WITHOUT TensorFlow.

The reference's main models (``obj_v3_model.tgz`` ...) are checkpoint directories restored with
``tf.train.Saver`` (reference obj_detect_tracking.py:404-416).  A V2 checkpoint ("tensor bundle")
is: an ``.index`` file in LevelDB table format (blocks of prefix-compressed key/value entries +
restart array, a 1+4 byte trailer per block, metaindex + index blocks, 48-byte footer ending in the
magic 0xdb4775248b80fb57) mapping "" -> BundleHeaderProto and each variable name ->
BundleEntryProto{dtype=1, shape=2, shard_id=3, offset=4, size=5, crc32c=6}, and data shards holding
the raw little-endian tensors.  TensorFlow writes the table uncompressed (tensor_bundle.cc,
BundleWriter: options.compression = kNoCompression); snappy blocks are rejected with a clear error.

No checkpoint ships with the reference and none is reachable offline, so the reader is exercised
against bundles produced by :func:`write_checkpoint` (same format, written by hand) -- unpinned
against TensorFlow's own writer.
"""
from __future__ import annotations

import os
import re
import struct

import numpy as np

from .frozen_pb import _enc_varint, _fields, _ld, _varint, _vi

_MAGIC = 0xdb4775248b80fb57
_NP = {1: "<f4", 2: "<f8", 3: "<i4", 9: "<i8", 19: "<f2"}      # DataType enum -> numpy


def _block(buf, offset, size):
  """Decode one table block -> [(key, value)]."""
  if buf[offset + size] != 0:
    raise ValueError("compressed table block (type %d): only uncompressed TensorFlow bundles are "
                     "supported" % buf[offset + size])
  blk = buf[offset:offset + size]
  nrestarts = struct.unpack_from("<I", blk, size - 4)[0]
  end = size - 4 - 4 * nrestarts
  out, i, key = [], 0, b""
  while i < end:
    shared, i = _varint(blk, i)
    non_shared, i = _varint(blk, i)
    vlen, i = _varint(blk, i)
    key = key[:shared] + bytes(blk[i:i + non_shared]); i += non_shared
    out.append((key, bytes(blk[i:i + vlen]))); i += vlen
  return out


def _handle(b, i=0):
  off, i = _varint(b, i)
  size, i = _varint(b, i)
  return off, size, i


def read_index(index_path):
  """{variable name: (dtype enum, shape, shard_id, offset, size)} from a bundle ``.index``."""
  with open(index_path, "rb") as fh:
    buf = fh.read()
  if len(buf) < 48 or struct.unpack_from("<Q", buf, len(buf) - 8)[0] != _MAGIC:
    raise ValueError("%s: not a TensorFlow bundle index (bad table magic)" % index_path)
  footer = buf[len(buf) - 48:]
  _, _, i = _handle(footer)                        # metaindex handle
  ioff, isize, _ = _handle(footer, i)              # index handle
  entries = {}
  for _, hv in _block(buf, ioff, isize):
    doff, dsize, _ = _handle(hv)
    for key, val in _block(buf, doff, dsize):
      if not key:
        continue                                   # BundleHeaderProto
      dtype, shape, shard, offset, size = 0, [], 0, 0, 0
      for f, wt, v in _fields(memoryview(val)):
        if f == 1:
          dtype = v
        elif f == 2:
          for f2, _, v2 in _fields(v):
            if f2 == 2:
              d = 0
              for f3, _, v3 in _fields(v2):
                if f3 == 1:
                  d = v3
              shape.append(d)
        elif f == 3:
          shard = v
        elif f == 4:
          offset = v
        elif f == 5:
          size = v
      entries[key.decode()] = (dtype, shape, shard, offset, size)
  return entries


def resolve_checkpoint(path):
  """Checkpoint prefix from a directory (its ``checkpoint`` state file, else the newest
  ``*.index``), a prefix, or an ``.index`` file -- what tf.train.get_checkpoint_state + restore do
  in reference obj_detect_tracking.py:404-416."""
  if os.path.isdir(path):
    state = os.path.join(path, "checkpoint")
    if os.path.exists(state):
      m = re.search(r'model_checkpoint_path:\s*"([^"]+)"', open(state).read())
      if m:
        p = m.group(1)
        return p if os.path.isabs(p) else os.path.join(path, os.path.basename(p))
    idx = sorted((f for f in os.listdir(path) if f.endswith(".index")),
                 key=lambda f: os.path.getmtime(os.path.join(path, f)))
    if not idx:
      raise ValueError("%s: no checkpoint state file and no *.index" % path)
    return os.path.join(path, idx[-1][:-len(".index")])
  return path[:-len(".index")] if path.endswith(".index") else path


def load_checkpoint(path, skip=("global_step", "learning_rate", "/Momentum", "/Adam", "/AccumGrad")):
  """{variable name: float32 array} of every floating-point variable (optimizer slots skipped)."""
  prefix = resolve_checkpoint(path)
  entries = read_index(prefix + ".index")
  nshards = 1 + max(e[2] for e in entries.values()) if entries else 1
  out, shards = {}, {}
  for name, (dtype, shape, shard, offset, size) in entries.items():
    if dtype not in (1, 2, 19) or not shape or any(s in name for s in skip):
      continue                                   # integer / scalar / optimizer-slot variables
    if shard not in shards:
      shards[shard] = np.memmap("%s.data-%05d-of-%05d" % (prefix, shard, nshards), np.uint8, "r")
    raw = shards[shard][offset:offset + size]
    a = np.frombuffer(raw.tobytes(), _NP[dtype])
    if a.size != int(np.prod(shape)) if shape else a.size != 1:
      raise ValueError("%s: %d values for shape %s" % (name, a.size, shape))
    out[name] = np.asarray(a, np.float32).reshape(shape)
  if not out:
    raise ValueError("%s: no floating-point variables" % prefix)
  return out


# ---- writer (tests / tooling) --------------------------------------------------------------------
def _table_block(entries, restart_interval=16):
  """LevelDB block with prefix compression and restart points (as table::BlockBuilder)."""
  body, restarts, prev, n = b"", [], b"", 0
  for key, val in entries:
    if n % restart_interval == 0:
      restarts.append(len(body)); shared = 0
    else:
      shared = 0
      while shared < min(len(prev), len(key)) and prev[shared] == key[shared]:
        shared += 1
    body += _enc_varint(shared) + _enc_varint(len(key) - shared) + _enc_varint(len(val)) + key[shared:] + val
    prev = key; n += 1
  if not restarts:
    restarts = [0]
  return body + b"".join(struct.pack("<I", r) for r in restarts) + struct.pack("<I", len(restarts))


def write_checkpoint(prefix, variables, per_block=7, encode_entry=None, encode_header=None):
  """Write {name: array} as a one-shard V2 checkpoint + the ``checkpoint`` state file.  ``encode_entry(dtype_enum,
  shape, offset, size) -> bytes`` / ``encode_header() -> bytes`` let a test substitute another serialiser for the
  BundleEntryProto / BundleHeaderProto values (the protobuf runtime, tests/tf_protos.py)."""
  data, recs = b"", []
  for name in sorted(variables):
    a = np.asarray(variables[name])
    dt = {np.dtype("float32"): 1, np.dtype("float64"): 2, np.dtype("int32"): 3, np.dtype("int64"): 9,
          np.dtype("float16"): 19}[a.dtype]
    raw = a.astype(a.dtype.newbyteorder("<")).tobytes()
    shape = b"".join(_ld(2, _vi(1, int(d))) for d in a.shape)
    entry = _vi(1, dt) + _ld(2, shape) + _vi(4, len(data)) + _vi(5, len(raw)) + \
        _enc_varint((6 << 3) | 5) + struct.pack("<I", 0)
    if encode_entry is not None:
      entry = encode_entry(dt, [int(d) for d in a.shape], len(data), len(raw))
    recs.append((name.encode(), entry)); data += raw
  header = _vi(1, 1) + _ld(3, _vi(1, 1))                         # BundleHeaderProto{num_shards=1, version}
  recs = [(b"", encode_header() if encode_header is not None else header)] + recs
  with open(prefix + ".data-00000-of-00001", "wb") as fh:
    fh.write(data)
  out, index_entries = b"", []
  def emit(block):
    nonlocal out
    off = len(out)
    out += block + b"\x00" + struct.pack("<I", 0)               # type 0 = uncompressed, crc (unchecked)
    return _enc_varint(off) + _enc_varint(len(block))
  for i in range(0, len(recs), per_block):
    chunk = recs[i:i + per_block]
    index_entries.append((chunk[-1][0] + b"\x00", emit(_table_block(chunk))))
  meta = emit(_table_block([]))
  index = emit(_table_block(index_entries, restart_interval=1))
  footer = meta + index
  out += footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", _MAGIC)
  with open(prefix + ".index", "wb") as fh:
    fh.write(out)
  with open(os.path.join(os.path.dirname(prefix) or ".", "checkpoint"), "w") as fh:
    fh.write('model_checkpoint_path: "%s"\nall_model_checkpoint_paths: "%s"\n' % (
        os.path.basename(prefix), os.path.basename(prefix)))
