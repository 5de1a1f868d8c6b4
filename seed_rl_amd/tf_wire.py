"""Hand-built descriptors of the TensorFlow / SEED wire messages (no protoc, no TensorFlow in this image).

Field numbers and types as published in TF 2.4.1: tensorflow/core/framework/{tensor_shape,tensor}.proto,
tensorflow/core/protobuf/{struct,tensor_bundle}.proto, and /root/reference/grpc/service.proto:28-57.  Built with
google.protobuf.descriptor_pb2 into a PRIVATE descriptor pool and turned into message classes by the protobuf runtime;
used by grpc_service.py (transport) and tf_checkpoint.py (tf.train.Checkpoint files).
"""
from google.protobuf import descriptor_pb2, descriptor_pool, message_factory

_F = descriptor_pb2.FieldDescriptorProto
_T = dict(int32=_F.TYPE_INT32, int64=_F.TYPE_INT64, sint64=_F.TYPE_SINT64, uint32=_F.TYPE_UINT32, uint64=_F.TYPE_UINT64,
          fixed32=_F.TYPE_FIXED32, float=_F.TYPE_FLOAT, double=_F.TYPE_DOUBLE, bool=_F.TYPE_BOOL, string=_F.TYPE_STRING, bytes=_F.TYPE_BYTES)


def _msg(parent, name, fields, oneof=None):
  """fields: (name, number, type, repeated[, in_oneof]); type = scalar name or '.pkg.Message'."""
  m = parent.message_type.add() if hasattr(parent, 'message_type') else parent.nested_type.add()
  m.name = name
  if oneof:
    m.oneof_decl.add().name = oneof
  for f in fields:
    fname, num, typ, rep = f[:4]
    fd = m.field.add()
    fd.name, fd.number = fname, num
    fd.label = _F.LABEL_REPEATED if rep else _F.LABEL_OPTIONAL
    if typ.startswith('.'):
      fd.type, fd.type_name = _F.TYPE_MESSAGE, typ
    else:
      fd.type = _T[typ]
    if len(f) > 4 and f[4]:
      fd.oneof_index = 0
  return m


def _build_pool():
  pool = descriptor_pool.DescriptorPool()
  # tensorflow/core/framework/{tensor_shape,tensor}.proto + tensorflow/core/protobuf/struct.proto (TF 2.4.1)
  tf = descriptor_pb2.FileDescriptorProto(name='seed_rl_amd/tf_wire.proto', package='tensorflow', syntax='proto3')
  shp = _msg(tf, 'TensorShapeProto', [('dim', 2, '.tensorflow.TensorShapeProto.Dim', True), ('unknown_rank', 3, 'bool', False)])
  _msg(shp, 'Dim', [('size', 1, 'int64', False), ('name', 2, 'string', False)])
  _msg(tf, 'TensorProto', [
      ('dtype', 1, 'int32', False), ('tensor_shape', 2, '.tensorflow.TensorShapeProto', False),
      ('version_number', 3, 'int32', False), ('tensor_content', 4, 'bytes', False), ('half_val', 13, 'int32', True),
      ('float_val', 5, 'float', True), ('double_val', 6, 'double', True), ('int_val', 7, 'int32', True),
      ('string_val', 8, 'bytes', True), ('scomplex_val', 9, 'float', True), ('int64_val', 10, 'int64', True),
      ('bool_val', 11, 'bool', True), ('dcomplex_val', 12, 'double', True), ('uint32_val', 16, 'uint32', True),
      ('uint64_val', 17, 'uint64', True)])
  SV = '.tensorflow.StructuredValue'
  _msg(tf, 'StructuredValue', [
      ('none_value', 1, '.tensorflow.NoneValue', False, True), ('float64_value', 11, 'double', False, True),
      ('int64_value', 12, 'sint64', False, True), ('string_value', 13, 'string', False, True),
      ('bool_value', 14, 'bool', False, True), ('tensor_shape_value', 31, '.tensorflow.TensorShapeProto', False, True),
      ('tensor_dtype_value', 32, 'int32', False, True), ('tensor_spec_value', 33, '.tensorflow.TensorSpecProto', False, True),
      ('list_value', 51, '.tensorflow.ListValue', False, True), ('tuple_value', 52, '.tensorflow.TupleValue', False, True),
      ('dict_value', 53, '.tensorflow.DictValue', False, True),
      ('named_tuple_value', 54, '.tensorflow.NamedTupleValue', False, True)], oneof='kind')
  _msg(tf, 'NoneValue', [])
  _msg(tf, 'ListValue', [('values', 1, SV, True)])
  _msg(tf, 'TupleValue', [('values', 1, SV, True)])
  dv = _msg(tf, 'DictValue', [('fields', 1, '.tensorflow.DictValue.FieldsEntry', True)])
  ent = _msg(dv, 'FieldsEntry', [('key', 1, 'string', False), ('value', 2, SV, False)])
  ent.options.map_entry = True
  _msg(tf, 'PairValue', [('key', 1, 'string', False), ('value', 2, SV, False)])
  _msg(tf, 'NamedTupleValue', [('name', 1, 'string', False), ('values', 2, '.tensorflow.PairValue', True)])
  _msg(tf, 'TensorSpecProto', [('name', 1, 'string', False), ('shape', 2, '.tensorflow.TensorShapeProto', False),
                               ('dtype', 3, 'int32', False)])
  # tensorflow/core/protobuf/tensor_bundle.proto (+ framework/versions.proto, tensor_slice.proto)
  _msg(tf, 'VersionDef', [('producer', 1, 'int32', False), ('min_consumer', 2, 'int32', False), ('bad_consumers', 3, 'int32', True)])
  _msg(tf, 'BundleHeaderProto', [('num_shards', 1, 'int32', False), ('endianness', 2, 'int32', False),
                                 ('version', 3, '.tensorflow.VersionDef', False)])
  ts = _msg(tf, 'TensorSliceProto', [('extent', 1, '.tensorflow.TensorSliceProto.Extent', True)])
  _msg(ts, 'Extent', [('start', 1, 'int64', False), ('length', 2, 'int64', False)])
  _msg(tf, 'BundleEntryProto', [('dtype', 1, 'int32', False), ('shape', 2, '.tensorflow.TensorShapeProto', False),
                                ('shard_id', 3, 'int32', False), ('offset', 4, 'int64', False), ('size', 5, 'int64', False),
                                ('crc32c', 6, 'fixed32', False), ('slices', 7, '.tensorflow.TensorSliceProto', True)])
  # tensorflow/core/protobuf/trackable_object_graph.proto (TF 2.4.1): what tf.train.Checkpoint stores under the key
  # _CHECKPOINTABLE_OBJECT_GRAPH and walks in restore()
  og = _msg(tf, 'TrackableObjectGraph', [('nodes', 1, '.tensorflow.TrackableObjectGraph.TrackableObject', True)])
  T = '.tensorflow.TrackableObjectGraph.TrackableObject'
  to = _msg(og, 'TrackableObject', [('children', 1, T + '.ObjectReference', True), ('attributes', 2, T + '.SerializedTensor', True),
                                    ('slot_variables', 3, T + '.SlotVariableReference', True)])
  _msg(to, 'ObjectReference', [('node_id', 1, 'int32', False), ('local_name', 2, 'string', False)])
  _msg(to, 'SerializedTensor', [('name', 1, 'string', False), ('full_name', 2, 'string', False),
                                ('checkpoint_key', 3, 'string', False), ('optional_restore', 4, 'bool', False)])
  _msg(to, 'SlotVariableReference', [('original_variable_node_id', 1, 'int32', False), ('slot_name', 2, 'string', False),
                                     ('slot_variable_node_id', 3, 'int32', False)])
  pool.Add(tf)
  # grpc/service.proto:28-57
  sv = descriptor_pb2.FileDescriptorProto(name='seed_rl_amd/service.proto', package='seed_rl', syntax='proto3')
  _msg(sv, 'InitRequest', [])
  _msg(sv, 'MethodOutputSignature', [('name', 1, 'string', False), ('output_specs', 2, 'bytes', False)])
  _msg(sv, 'InitResponse', [('method_output_signature', 1, '.seed_rl.MethodOutputSignature', True)])
  _msg(sv, 'CallRequest', [('function', 1, 'string', False), ('tensor', 2, 'bytes', True)])
  _msg(sv, 'CallResponse', [('tensor', 1, 'bytes', True), ('status_code', 2, 'int32', False),
                            ('status_error_message', 3, 'string', False)])
  pool.Add(sv)
  return pool


POOL = _build_pool()


def message_class(name):
  return message_factory.GetMessageClass(POOL.FindMessageTypeByName(name))
