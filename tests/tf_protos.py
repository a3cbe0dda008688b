"""TensorFlow's GraphDef message family declared for Google's protobuf runtime (test infrastructure).

The messages and field numbers are those of tensorflow/core/framework/{graph,node_def,attr_value,tensor,tensor_shape,
types,versions}.proto (public, stable since TF 1.0); only what a frozen graph uses is declared.  With them the protobuf
runtime -- not this repository's hand-written wire-format writer -- serialises the files the readers are tested on:
field order, packed repeated scalars, map entries and length prefixes are then the official encoder's.
"""
from google.protobuf import descriptor_pb2, descriptor_pool, message_factory

_T = descriptor_pb2.FieldDescriptorProto


def _field(msg, name, number, ftype, label=_T.LABEL_OPTIONAL, type_name=None, oneof=None, packed=None):
  f = msg.field.add()
  f.name, f.number, f.type, f.label = name, number, ftype, label
  if type_name:
    f.type_name = type_name
  if oneof is not None:
    f.oneof_index = oneof
  if packed is not None:
    f.options.packed = packed
  return f


def _build():
  fd = descriptor_pb2.FileDescriptorProto()
  fd.name = "odt_test_tf_graph.proto"; fd.package = "odt_tf"; fd.syntax = "proto3"
  # TensorShapeProto { message Dim { int64 size = 1; string name = 2; } repeated Dim dim = 2; bool unknown_rank = 3; }
  shp = fd.message_type.add(); shp.name = "TensorShapeProto"
  dim = shp.nested_type.add(); dim.name = "Dim"
  _field(dim, "size", 1, _T.TYPE_INT64); _field(dim, "name", 2, _T.TYPE_STRING)
  _field(shp, "dim", 2, _T.TYPE_MESSAGE, _T.LABEL_REPEATED, ".odt_tf.TensorShapeProto.Dim")
  _field(shp, "unknown_rank", 3, _T.TYPE_BOOL)
  # TensorProto
  ten = fd.message_type.add(); ten.name = "TensorProto"
  _field(ten, "dtype", 1, _T.TYPE_INT32)                    # enum DataType on the wire: a varint
  _field(ten, "tensor_shape", 2, _T.TYPE_MESSAGE, type_name=".odt_tf.TensorShapeProto")
  _field(ten, "version_number", 3, _T.TYPE_INT32)
  _field(ten, "tensor_content", 4, _T.TYPE_BYTES)
  _field(ten, "float_val", 5, _T.TYPE_FLOAT, _T.LABEL_REPEATED, packed=True)
  _field(ten, "double_val", 6, _T.TYPE_DOUBLE, _T.LABEL_REPEATED, packed=True)
  _field(ten, "int_val", 7, _T.TYPE_INT32, _T.LABEL_REPEATED, packed=True)
  _field(ten, "half_val", 13, _T.TYPE_INT32, _T.LABEL_REPEATED, packed=True)
  # AttrValue { message ListValue {...} oneof value { s=2 i=3 f=4 b=5 type=6 shape=7 tensor=8 list=1 } }
  av = fd.message_type.add(); av.name = "AttrValue"
  lv = av.nested_type.add(); lv.name = "ListValue"
  _field(lv, "s", 2, _T.TYPE_BYTES, _T.LABEL_REPEATED)
  _field(lv, "i", 3, _T.TYPE_INT64, _T.LABEL_REPEATED, packed=True)
  _field(lv, "f", 4, _T.TYPE_FLOAT, _T.LABEL_REPEATED, packed=True)
  av.oneof_decl.add().name = "value"
  _field(av, "list", 1, _T.TYPE_MESSAGE, type_name=".odt_tf.AttrValue.ListValue", oneof=0)
  _field(av, "s", 2, _T.TYPE_BYTES, oneof=0)
  _field(av, "i", 3, _T.TYPE_INT64, oneof=0)
  _field(av, "f", 4, _T.TYPE_FLOAT, oneof=0)
  _field(av, "b", 5, _T.TYPE_BOOL, oneof=0)
  _field(av, "type", 6, _T.TYPE_INT32, oneof=0)
  _field(av, "shape", 7, _T.TYPE_MESSAGE, type_name=".odt_tf.TensorShapeProto", oneof=0)
  _field(av, "tensor", 8, _T.TYPE_MESSAGE, type_name=".odt_tf.TensorProto", oneof=0)
  # NodeDef { string name = 1; string op = 2; repeated string input = 3; string device = 4; map<string, AttrValue> attr = 5; }
  nd = fd.message_type.add(); nd.name = "NodeDef"
  ent = nd.nested_type.add(); ent.name = "AttrEntry"; ent.options.map_entry = True
  _field(ent, "key", 1, _T.TYPE_STRING)
  _field(ent, "value", 2, _T.TYPE_MESSAGE, type_name=".odt_tf.AttrValue")
  _field(nd, "name", 1, _T.TYPE_STRING); _field(nd, "op", 2, _T.TYPE_STRING)
  _field(nd, "input", 3, _T.TYPE_STRING, _T.LABEL_REPEATED)
  _field(nd, "device", 4, _T.TYPE_STRING)
  _field(nd, "attr", 5, _T.TYPE_MESSAGE, _T.LABEL_REPEATED, ".odt_tf.NodeDef.AttrEntry")
  # VersionDef { int32 producer = 1; int32 min_consumer = 2; }   GraphDef { repeated NodeDef node = 1; VersionDef versions = 4; }
  ver = fd.message_type.add(); ver.name = "VersionDef"
  _field(ver, "producer", 1, _T.TYPE_INT32); _field(ver, "min_consumer", 2, _T.TYPE_INT32)
  gd = fd.message_type.add(); gd.name = "GraphDef"
  _field(gd, "node", 1, _T.TYPE_MESSAGE, _T.LABEL_REPEATED, ".odt_tf.NodeDef")
  _field(gd, "versions", 4, _T.TYPE_MESSAGE, type_name=".odt_tf.VersionDef")
  # tensorflow/core/protobuf/tensor_bundle.proto: the values of a V2 checkpoint's .index table
  # BundleHeaderProto { int32 num_shards = 1; Endianness endianness = 2; VersionDef version = 3; }
  bh = fd.message_type.add(); bh.name = "BundleHeaderProto"
  _field(bh, "num_shards", 1, _T.TYPE_INT32); _field(bh, "endianness", 2, _T.TYPE_INT32)
  _field(bh, "version", 3, _T.TYPE_MESSAGE, type_name=".odt_tf.VersionDef")
  # BundleEntryProto { DataType dtype = 1; TensorShapeProto shape = 2; int32 shard_id = 3; int64 offset = 4; int64 size = 5;
  #                    fixed32 crc32c = 6; repeated TensorSliceProto slices = 7; }
  be = fd.message_type.add(); be.name = "BundleEntryProto"
  _field(be, "dtype", 1, _T.TYPE_INT32)
  _field(be, "shape", 2, _T.TYPE_MESSAGE, type_name=".odt_tf.TensorShapeProto")
  _field(be, "shard_id", 3, _T.TYPE_INT32); _field(be, "offset", 4, _T.TYPE_INT64); _field(be, "size", 5, _T.TYPE_INT64)
  _field(be, "crc32c", 6, _T.TYPE_FIXED32)
  pool = descriptor_pool.DescriptorPool()
  pool.Add(fd)
  get = lambda n: message_factory.GetMessageClass(pool.FindMessageTypeByName("odt_tf." + n))
  return {n: get(n) for n in ("GraphDef", "NodeDef", "AttrValue", "TensorProto", "TensorShapeProto", "VersionDef",
                              "BundleHeaderProto", "BundleEntryProto")}


MESSAGES = _build()
DT_FLOAT, DT_DOUBLE, DT_INT32, DT_HALF = 1, 2, 3, 19


def const_node(graph, name, array, dtype=DT_FLOAT, how="content"):
  """Append a Const node the way graph_util.convert_variables_to_constants does (attrs dtype + value)."""
  import numpy as np
  n = graph.node.add(); n.name = name; n.op = "Const"
  n.attr["dtype"].type = dtype
  t = n.attr["value"].tensor
  t.dtype = dtype
  for d in array.shape:
    t.tensor_shape.dim.add().size = int(d)
  if dtype == DT_HALF:
    if how == "content":
      t.tensor_content = np.asarray(array, "<f2").tobytes()
    else:
      t.half_val.extend(int(v) for v in np.asarray(array, "<f2").reshape(-1).view(np.uint16))
  elif dtype == DT_DOUBLE:
    if how == "content":
      t.tensor_content = np.asarray(array, "<f8").tobytes()
    else:
      t.double_val.extend(float(v) for v in np.asarray(array, np.float64).reshape(-1))
  else:
    if how == "content":
      t.tensor_content = np.asarray(array, "<f4").tobytes()
    elif how == "splat":                       # a constant-filled tensor: one value, full shape
      t.float_val.append(float(np.asarray(array).reshape(-1)[0]))
    else:
      t.float_val.extend(float(v) for v in np.asarray(array, np.float32).reshape(-1))
  return n
