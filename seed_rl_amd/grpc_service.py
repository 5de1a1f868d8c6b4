"""gRPC front-end of the learner: wire-compatible with the reference's TensorService and its dynamic batching
(SURVEY.md 8(f) rank 2).

Reference: /root/reference/grpc/service.proto:28-57 (service TensorService { Init; stream Call }),
/root/reference/grpc/ops/grpc.cc:141-233 (TensorHandler: function table, round-robin over functions bound under one
name), :527-861 (verify_args / GetArgBatchSize / DynamicFn: server-side batching of N single calls -- or of client-side
batches of k with k | N -- into ONE invocation of the bound function, direct calls for exact-shape arguments, output
slicing back to the callers), :870-973 (GrpcServerBindOp / CanBatch), /root/reference/grpc/python/ops.py (Server / Client
surface); behaviour pinned by /root/reference/grpc/python/ops_test.py (replayed in tests/test_grpc_service.py).

Wire format.  The messages of service.proto carry `bytes` fields holding serialized tensorflow.TensorProto /
tensorflow.StructuredValue messages.  Neither protoc nor TensorFlow exists in this image, so the descriptors of those
messages (tensor.proto, tensor_shape.proto, struct.proto of TF 2.4.1: field numbers and types as published) are built
here with google.protobuf.descriptor_pb2 and turned into message classes by the protobuf runtime; the gRPC plumbing is
grpcio's generic handlers.  An unmodified reference actor (grpc_client_call op) therefore finds the same service name,
method names, message layout, status codes and error strings.

The transport is NOT the product's hot path; what it feeds is: `bind_inference` batches actor requests exactly like the
reference (inference_batch_size single-step requests per call, learner.py:339-349), packs them into ONE pinned host
buffer in `inference.request_layout` and replays the captured HIP graph of FusedInferenceState (two H2D copies + one
graph launch per batch).
"""
import asyncio
import collections
import concurrent.futures
import threading
import time

import grpc
import numpy as np

from seed_rl_amd import tf_wire

_cls = tf_wire.message_class
TensorProto, StructuredValue = _cls('tensorflow.TensorProto'), _cls('tensorflow.StructuredValue')
InitRequest, InitResponse = _cls('seed_rl.InitRequest'), _cls('seed_rl.InitResponse')
CallRequest, CallResponse = _cls('seed_rl.CallRequest'), _cls('seed_rl.CallResponse')
SERVICE = 'seed_rl.TensorService'

# --------------------------------------------------------------------------------------------------------------------- #
# TensorProto <-> numpy (tensorflow/core/framework/types.proto DataType values; Tensor::AsProtoTensorContent / FromProto).
# --------------------------------------------------------------------------------------------------------------------- #
DT_FLOAT, DT_DOUBLE, DT_INT32, DT_UINT8, DT_INT16, DT_INT8, DT_STRING, DT_INT64, DT_BOOL = 1, 2, 3, 4, 5, 6, 7, 9, 10
DT_UINT16, DT_HALF, DT_UINT32, DT_UINT64 = 17, 19, 22, 23
_NP_OF = {DT_FLOAT: np.float32, DT_DOUBLE: np.float64, DT_INT32: np.int32, DT_UINT8: np.uint8, DT_INT16: np.int16,
          DT_INT8: np.int8, DT_INT64: np.int64, DT_BOOL: np.bool_, DT_UINT16: np.uint16, DT_HALF: np.float16,
          DT_UINT32: np.uint32, DT_UINT64: np.uint64, DT_STRING: np.object_}
_DT_OF = {np.dtype(v): k for k, v in _NP_OF.items() if k != DT_STRING}
_DT_NAME = {DT_FLOAT: 'float', DT_DOUBLE: 'double', DT_INT32: 'int32', DT_UINT8: 'uint8', DT_INT16: 'int16',
            DT_INT8: 'int8', DT_STRING: 'string', DT_INT64: 'int64', DT_BOOL: 'bool', DT_UINT16: 'uint16',
            DT_HALF: 'half', DT_UINT32: 'uint32', DT_UINT64: 'uint64'}      # DataTypeString()
_VAL_FIELD = {DT_FLOAT: 'float_val', DT_DOUBLE: 'double_val', DT_INT32: 'int_val', DT_UINT8: 'int_val', DT_INT16: 'int_val',
              DT_INT8: 'int_val', DT_INT64: 'int64_val', DT_BOOL: 'bool_val', DT_UINT16: 'int_val', DT_HALF: 'half_val',
              DT_UINT32: 'uint32_val', DT_UINT64: 'uint64_val'}


def dtype_enum(dtype):
  """numpy / torch-style dtype or str/bytes -> DataType enum value."""
  if isinstance(dtype, str) and dtype == 'string':
    return DT_STRING
  if dtype is str or dtype is bytes or np.dtype(dtype).kind in ('O', 'S', 'U'):
    return DT_STRING
  return _DT_OF[np.dtype(dtype)]


def _as_array(value, dt=None):
  if isinstance(value, (str, bytes)):
    return np.array(value.encode() if isinstance(value, str) else value, dtype=np.object_)
  if hasattr(value, 'detach'):                       # torch tensor
    value = value.detach().cpu().numpy()
  a = np.asarray(value)
  if a.dtype.kind in ('U', 'S'):
    a = np.array([x.encode() if isinstance(x, str) else bytes(x) for x in a.reshape(-1)], dtype=np.object_).reshape(a.shape)
  elif a.dtype == np.float64 and dt is None and not isinstance(value, (np.ndarray, np.generic)):
    a = a.astype(np.float32)                         # Python floats are float32 tensors in TF
  elif a.dtype == np.int64 and dt is None and not isinstance(value, (np.ndarray, np.generic)):
    a = a.astype(np.int32)                           # Python ints are int32 tensors in TF
  if dt is not None and a.dtype.kind != 'O':
    a = a.astype(_NP_OF[dt], copy=False)
  return a


def _varint(n):
  out = bytearray()
  while True:
    b = n & 0x7F
    n >>= 7
    if n:
      out.append(b | 0x80)
    else:
      out.append(b)
      return bytes(out)


def encode_tensor(value, dt=None):
  """numpy array / scalar / bytes -> serialized TensorProto (tensor_content encoding, like AsProtoTensorContent)."""
  a = _as_array(value, dt)
  tp = TensorProto()
  for d in a.shape:
    tp.tensor_shape.dim.add().size = int(d)
  if a.dtype.kind == 'O':
    tp.dtype = DT_STRING
    items = [bytes(x) for x in a.reshape(-1)]
    tp.tensor_content = b''.join(_varint(len(x)) for x in items) + b''.join(items)     # port::EncodeStringList
  else:
    tp.dtype = _DT_OF[a.dtype]
    tp.tensor_content = np.ascontiguousarray(a).tobytes()
  return tp.SerializeToString()


def decode_tensor(data):
  """Serialized TensorProto -> (numpy array, DataType enum).  Accepts tensor_content and the typed *_val fields
  (with TF's "last value repeats" fill rule)."""
  tp = TensorProto()
  tp.ParseFromString(data)                            # raises google.protobuf.message.DecodeError
  shape = tuple(int(d.size) for d in tp.tensor_shape.dim)
  n = int(np.prod(shape, dtype=np.int64))
  dt = tp.dtype
  if dt not in _NP_OF:
    raise ValueError('unsupported tensor dtype %d' % dt)
  if dt == DT_STRING:
    if tp.tensor_content:
      buf, lens, pos = tp.tensor_content, [], 0
      for _ in range(n):
        v, s = 0, 0
        while True:
          b = buf[pos]; pos += 1
          v |= (b & 0x7F) << s
          s += 7
          if not b & 0x80:
            break
        lens.append(v)
      items = []
      for ln in lens:
        items.append(bytes(buf[pos:pos + ln])); pos += ln
    else:
      items = list(tp.string_val)
      items += [items[-1] if items else b''] * (n - len(items))
    a = np.empty(n, dtype=np.object_)
    a[:] = items[:n]
    return a.reshape(shape), dt
  npdt = np.dtype(_NP_OF[dt])
  if tp.tensor_content:
    return np.frombuffer(tp.tensor_content, dtype=npdt).reshape(shape).copy(), dt
  vals = list(getattr(tp, _VAL_FIELD[dt]))
  if dt == DT_HALF:
    a = np.array(vals, dtype=np.uint16).view(np.float16)
  else:
    a = np.array(vals).astype(npdt)
  if a.size < n:
    a = np.concatenate([a, np.full(n - a.size, a[-1] if a.size else 0, dtype=npdt)])
  return a[:n].reshape(shape), dt


# --------------------------------------------------------------------------------------------------------------------- #
# Nests and their StructuredValue coding (tensorflow/python/saved_model/nested_structure_coder.py).
# --------------------------------------------------------------------------------------------------------------------- #
class TensorSpec(collections.namedtuple('TensorSpec', 'shape dtype name')):
  """tf.TensorSpec stand-in: shape (tuple, None for an unknown dimension), dtype (numpy dtype or str for strings)."""

  def __new__(cls, shape, dtype, name=None):
    if isinstance(shape, int):
      shape = (shape,)
    return super(TensorSpec, cls).__new__(cls, None if shape is None else tuple(shape), dtype, name)


def _is_namedtuple(x):
  return isinstance(x, tuple) and hasattr(x, '_fields') and not isinstance(x, TensorSpec)


def flatten(nest):
  """tf.nest.flatten (dict values in sorted-key order)."""
  if nest is None:
    return []
  if isinstance(nest, TensorSpec) or not isinstance(nest, (list, tuple, dict)):
    return [nest]
  out = []
  for x in ([nest[k] for k in sorted(nest)] if isinstance(nest, dict) else nest):
    out.extend(flatten(x))
  return out


def pack_sequence_as(structure, flat):
  it = iter(flat)

  def rec(s):
    if s is None:
      return None
    if isinstance(s, TensorSpec) or not isinstance(s, (list, tuple, dict)):
      return next(it)
    if isinstance(s, dict):
      vals = {k: rec(s[k]) for k in sorted(s)}
      return type(s)((k, vals[k]) for k in s)
    if _is_namedtuple(s):
      return type(s)(*[rec(x) for x in s])
    return type(s)(rec(x) for x in s)
  return rec(structure)


def encode_structure(nest):
  v = StructuredValue()
  if nest is None:
    v.none_value.SetInParent()
  elif isinstance(nest, TensorSpec):
    ts = v.tensor_spec_value
    ts.name = nest.name or ''
    if nest.shape is None:
      ts.shape.unknown_rank = True
    else:
      for d in nest.shape:
        ts.shape.dim.add().size = -1 if d is None else int(d)
    ts.dtype = dtype_enum(nest.dtype)
  elif isinstance(nest, dict):
    for k in nest:
      v.dict_value.fields[k].CopyFrom(encode_structure(nest[k]))
  elif _is_namedtuple(nest):
    v.named_tuple_value.name = type(nest).__name__
    for k, x in zip(nest._fields, nest):
      p = v.named_tuple_value.values.add()
      p.key = k
      p.value.CopyFrom(encode_structure(x))
  elif isinstance(nest, tuple):
    v.tuple_value.SetInParent()
    for x in nest:
      v.tuple_value.values.add().CopyFrom(encode_structure(x))
  elif isinstance(nest, list):
    v.list_value.SetInParent()
    for x in nest:
      v.list_value.values.add().CopyFrom(encode_structure(x))
  else:
    raise TypeError('cannot encode %r as a StructuredValue' % (nest,))
  return v


def decode_structure(v):
  kind = v.WhichOneof('kind')
  if kind is None or kind == 'none_value':
    return None
  if kind == 'tensor_spec_value':
    ts = v.tensor_spec_value
    shape = None if ts.shape.unknown_rank else tuple(None if d.size < 0 else int(d.size) for d in ts.shape.dim)
    return TensorSpec(shape, 'string' if ts.dtype == DT_STRING else _NP_OF[ts.dtype], ts.name or None)
  if kind == 'list_value':
    return [decode_structure(x) for x in v.list_value.values]
  if kind == 'tuple_value':
    return tuple(decode_structure(x) for x in v.tuple_value.values)
  if kind == 'dict_value':
    return {k: decode_structure(v.dict_value.fields[k]) for k in v.dict_value.fields}
  if kind == 'named_tuple_value':
    keys = [p.key for p in v.named_tuple_value.values]
    return collections.namedtuple(v.named_tuple_value.name, keys)(*[decode_structure(p.value) for p in v.named_tuple_value.values])
  raise ValueError('unsupported StructuredValue kind %s' % kind)


# --------------------------------------------------------------------------------------------------------------------- #
# Status codes (tensorflow/core/lib/core/error_codes.proto == grpc status codes) and the errors they map to.
# --------------------------------------------------------------------------------------------------------------------- #
OK, CANCELLED, UNKNOWN, INVALID_ARGUMENT, NOT_FOUND, INTERNAL, UNAVAILABLE = 0, 1, 2, 3, 5, 13, 14


class OpError(Exception):
  code = UNKNOWN

  def __init__(self, message, code=None):
    super(OpError, self).__init__(message)
    self.message = message
    if code is not None:
      self.code = code


class CancelledError(OpError):
  code = CANCELLED


class InvalidArgumentError(OpError):
  code = INVALID_ARGUMENT


class InternalError(OpError):
  code = INTERNAL


class UnavailableError(OpError):
  code = UNAVAILABLE


_ERR = {CANCELLED: CancelledError, INVALID_ARGUMENT: InvalidArgumentError, INTERNAL: InternalError,
        UNAVAILABLE: UnavailableError}


def _shape_str(shape):
  return '[' + ','.join(str(int(d)) for d in shape) + ']'           # TensorShape::DebugString()


def function(input_signature, output_signature='infer'):
  """The role of `@tf.function(input_signature=...)` for a function bound to the server: attaches the nest of
  TensorSpec the arguments must match.  output_signature: nest of TensorSpec of the result, None for "no output", or
  'infer' -- the function is then called ONCE on zeros at bind time (what tracing does for a tf.function)."""
  def deco(fn):
    fn.input_signature = input_signature
    fn.output_signature = output_signature
    return fn
  return deco


# --------------------------------------------------------------------------------------------------------------------- #
# Server side: verify_args / GetArgBatchSize / DynamicFn (grpc.cc:527-861).
# --------------------------------------------------------------------------------------------------------------------- #
def _verify_args(types, shapes, batching_dims, args):
  if len(types) != len(args):
    return 'Expects %d arguments, but %d is provided' % (len(types), len(args))
  for i, (a, dt) in enumerate(args):
    if len(shapes[i]) + batching_dims != a.ndim:
      return 'Expects arg[%d] to have shape with %d dimension(s), but had shape %s' % (
          i, len(shapes[i]) + batching_dims, _shape_str(a.shape))
    if tuple(a.shape[a.ndim - len(shapes[i]):]) != tuple(shapes[i]):
      return 'Expects arg[%d] to have shape with suffix %s, but had shape %s' % (i, _shape_str(shapes[i]), _shape_str(a.shape))
    if types[i] != dt:
      return 'Expects arg[%d] to be %s but %s is provided' % (i, _DT_NAME[types[i]], _DT_NAME.get(dt, str(dt)))
  return None


class _Computation(object):
  __slots__ = ('request', 'callbacks', 'num_ready')

  def __init__(self, types, shapes, batch_size):
    self.request = [np.empty(s, dtype=_NP_OF[t]) for t, s in zip(types, shapes)]
    self.callbacks = [None] * batch_size
    self.num_ready = 0


class _DynamicFn(object):
  """One bound function with the reference's batching rules (grpc.cc:591-861)."""

  def __init__(self, fn, executor):
    self.fn, self.executor = fn, executor
    sig = fn.input_signature
    self.signature = sig
    specs = flatten(sig)
    self.types = [dtype_enum(s.dtype) for s in specs]
    self.shapes = [tuple(s.shape) for s in specs]
    out = getattr(fn, 'output_signature', 'infer')
    if isinstance(out, str) and out == 'infer':
      zeros = [np.zeros(s, dtype=_NP_OF[t]) if t != DT_STRING else np.full(s, b'', dtype=np.object_)
               for t, s in zip(self.types, self.shapes)]
      out = _map_to_specs(fn(*pack_sequence_as(sig, zeros)))
    self.output_specs = out
    out_shapes = [s.shape for s in flatten(out)]
    # CanBatch (grpc.cc:948-973): every input has rank >= 1 and the same leading dimension N; no output is a scalar or
    # has a known leading dimension other than N (unknown ranks / leading dimensions pass and are checked per call)
    self.batch_size = -1
    if self.shapes and all(len(s) > 0 and s[0] == self.shapes[0][0] for s in self.shapes) and \
        all(s is None or (len(s) > 0 and (s[0] is None or s[0] == self.shapes[0][0])) for s in out_shapes):
      self.batch_size = self.shapes[0][0]
    self.arg_shapes = [s[1:] for s in self.shapes] if self.batch_size != -1 else None
    self.mu = threading.Lock()
    self.next_index = 0
    self.current = _Computation(self.types, self.shapes, self.batch_size) if self.batch_size != -1 else None

  def _run(self, arrays):
    res = self.fn(*pack_sequence_as(self.signature, arrays))
    return [_as_array(x, dtype_enum(s.dtype)) for x, s in zip(flatten(res), flatten(self.output_specs))] \
        if self.output_specs is not None else []

  def __call__(self, args, callback):
    """args: [(array, dtype enum)]; callback(code, message, [arrays]).  Returns True when the call counted for the
    round-robin (a direct call was started, or a batch was filled and started)."""
    if self.batch_size == -1 or (args and tuple(args[0][0].shape) == self.shapes[0]):
      return self._direct(args, callback)
    batched = bool(args) and bool(self.arg_shapes) and args[0][0].ndim == len(self.arg_shapes[0]) + 1
    err = _verify_args(self.types, self.arg_shapes, 1 if batched else 0, args)
    if err is None and batched:
      n0 = args[0][0].shape[0]
      for i in range(1, len(args)):
        if args[i][0].shape[0] != n0:
          err = 'Expects arg[%d] to start with the batching dimension %d but had shape %s' % (
              i, n0, _shape_str(args[i][0].shape))
          break
    if err is not None:
      callback(INVALID_ARGUMENT, err, [])
      return False
    count = args[0][0].shape[0] if batched else 1
    with self.mu:
      index, comp = self.next_index, self.current
      if index + count > self.batch_size:
        callback(INVALID_ARGUMENT, 'Learner-side batch size exceeded', [])     # (a CHECK failure in the reference)
        return False
      self.next_index += count
      if self.next_index == self.batch_size:
        self.next_index = 0
        self.current = _Computation(self.types, self.shapes, self.batch_size)
    for (a, _), dst in zip(args, comp.request):
      if batched:
        dst[index:index + count] = a
      else:
        dst[index] = a
    with self.mu:
      comp.callbacks[index + count - 1] = (callback, index, batched)
      comp.num_ready += count
      full = comp.num_ready == self.batch_size
    if full:
      self.executor.submit(self._run_batch, comp)
      return True
    return False

  def _run_batch(self, comp):
    code, msg, outs = OK, '', []
    try:
      outs = self._run(comp.request)
      for o in outs:
        if o.ndim <= 0:
          code, msg = INVALID_ARGUMENT, 'Output must be at least rank 1 when batching is enabled'
          break
        if o.shape[0] != self.batch_size:
          code, msg = INVALID_ARGUMENT, ('All outputs must have the same batch size as the inputs when batching is '
                                         'enabled, expected: %d was: %d' % (self.batch_size, o.shape[0]))
          break
    except OpError as e:
      code, msg = e.code, e.message
    except Exception as e:                           # pylint: disable=broad-except
      code, msg = INVALID_ARGUMENT if isinstance(e, (ValueError, AssertionError)) else INTERNAL, '%s: %s' % (type(e).__name__, e)
    for j, entry in enumerate(comp.callbacks):
      if entry is None:
        continue
      cb, start, batched = entry
      rets = []
      if code == OK:
        rets = [o[start:j + 1] if batched else o[j] for o in outs]
      cb(code, msg, rets)

  def _direct(self, args, callback):
    err = _verify_args(self.types, self.shapes, 0, args)
    if err is not None:
      callback(INVALID_ARGUMENT, err, [])
      return False

    def run():
      try:
        callback(OK, '', self._run([a for a, _ in args]))
      except OpError as e:
        callback(e.code, e.message, [])
      except Exception as e:                         # pylint: disable=broad-except
        callback(INVALID_ARGUMENT if isinstance(e, (ValueError, AssertionError)) else INTERNAL,
                 '%s: %s' % (type(e).__name__, e), [])
    self.executor.submit(run)
    return True

  def shutdown(self):
    with self.mu:
      if self.current is not None:
        for entry in self.current.callbacks:
          if entry is not None:
            entry[0](CANCELLED, 'Server shutdown.', [])
        self.current = _Computation(self.types, self.shapes, self.batch_size)
        self.next_index = 0


def _map_to_specs(res):
  if res is None:
    return None
  if isinstance(res, (list, tuple, dict)):
    if isinstance(res, dict):
      return {k: _map_to_specs(v) for k, v in res.items()}
    if _is_namedtuple(res):
      return type(res)(*[_map_to_specs(x) for x in res])
    return type(res)(_map_to_specs(x) for x in res)
  a = _as_array(res)
  return TensorSpec(a.shape, 'string' if a.dtype.kind == 'O' else a.dtype)


class Server(object):
  """grpc/python/ops.py:Server -- `Server([addresses])`, `bind(fn | [fn, ...])`, `start()`, `shutdown()`."""

  def __init__(self, server_addresses, num_workers=8):
    if isinstance(server_addresses, (str, bytes)) or not hasattr(server_addresses, '__iter__'):
      raise InvalidArgumentError('server_addresses must be a vector, got shape: []')
    self._addresses = list(server_addresses)
    self._fns = collections.OrderedDict()            # name -> {'fn': [...], 'counter': int}
    self._specs = []                                 # [(name, output_specs)] in bind order
    self._executor = concurrent.futures.ThreadPoolExecutor(max_workers=num_workers, thread_name_prefix='batched_fn')
    self._loop = self._thread = self._server = None
    self._lock = threading.Lock()
    self._is_shutdown = False

  # ---- TensorHandler (grpc.cc:141-233) ---- #
  def bind(self, fn):
    fns = list(fn) if isinstance(fn, (list, tuple)) else [fn]
    for i, f in enumerate(fns):
      if getattr(f, 'input_signature', None) is None:
        raise ValueError('the bound function must have input_signature set (grpc_service.function)')
      name = f.__name__
      if i == 0 and name in self._fns:
        raise InvalidArgumentError("Function '%s' was bound twice." % name)
      dyn = _DynamicFn(f, self._executor)
      bucket = self._fns.setdefault(name, {'fn': [], 'counter': 0})
      if not bucket['fn']:
        self._specs.append((name, dyn.output_specs))
      bucket['fn'].append(dyn)

  def _call(self, request, done):
    """TensorHandler::Call: decode, dispatch round-robin, encode.  done(CallResponse)."""
    def callback(code, msg, rets):
      if self._is_shutdown:                          # grpc.cc:336-343: nothing is written once the server shuts down;
        return                                       # the client's read fails ("... is the server closed?")
      resp = CallResponse()
      if code == OK:
        for r in rets:
          resp.tensor.append(encode_tensor(r))
      else:
        resp.status_code, resp.status_error_message = code, msg
      done(resp)
    try:
      args = [decode_tensor(t) for t in request.tensor]
    except Exception:                                # pylint: disable=broad-except
      callback(INVALID_ARGUMENT, 'Cannot parse TensorProto.', [])
      return
    bucket = self._fns.get(request.function)
    if bucket is None:
      callback(INTERNAL, 'Function %s not found' % request.function, [])
      return
    with self._lock:
      dyn = bucket['fn'][bucket['counter'] % len(bucket['fn'])]
    if dyn(args, callback):
      with self._lock:
        bucket['counter'] += 1

  # ---- gRPC plumbing (grpc.aio, generic handlers: no generated stubs) ---- #
  def start(self):
    if not self._specs:
      raise UnavailableError('No function was bound')
    if self._server is not None:
      raise InvalidArgumentError('Server is already started')
    self._is_shutdown = False
    started = concurrent.futures.Future()

    async def init(request, context):
      del request, context
      resp = InitResponse()
      for name, specs in self._specs:
        sig = resp.method_output_signature.add()
        sig.name = name
        sig.output_specs = encode_structure(specs).SerializeToString()
      return resp

    async def call(request_iterator, context):
      del context
      loop = asyncio.get_running_loop()
      try:
        async for request in request_iterator:
          fut = loop.create_future()

          def done(resp, fut=fut):
            loop.call_soon_threadsafe(lambda: fut.done() or fut.set_result(resp))
          self._call(request, done)
          yield await fut
      except asyncio.CancelledError:                 # stream cut by shutdown / a vanished client: nothing to report
        return

    async def main():
      server = grpc.aio.server(options=[('grpc.max_receive_message_length', -1), ('grpc.max_send_message_length', -1)])
      handler = grpc.method_handlers_generic_handler(SERVICE, {
          'Init': grpc.unary_unary_rpc_method_handler(init, request_deserializer=InitRequest.FromString,
                                                      response_serializer=InitResponse.SerializeToString),
          'Call': grpc.stream_stream_rpc_method_handler(call, request_deserializer=CallRequest.FromString,
                                                        response_serializer=CallResponse.SerializeToString)})
      server.add_generic_rpc_handlers((handler,))
      for a in self._addresses:
        server.add_insecure_port(a)
      await server.start()
      self._server = server
      self._stop = asyncio.Event()
      started.set_result(True)
      await self._stop.wait()
      await server.stop(0)

    def run():
      self._loop = asyncio.new_event_loop()
      asyncio.set_event_loop(self._loop)
      try:
        self._loop.run_until_complete(main())
      except Exception as e:                         # pylint: disable=broad-except
        if not started.done():
          started.set_exception(e)
      finally:
        pending = [t for t in asyncio.all_tasks(self._loop) if not t.done()]
        for t in pending:                            # stream handlers cut off by stop(0)
          t.cancel()
        if pending:
          self._loop.run_until_complete(asyncio.gather(*pending, return_exceptions=True))
        self._loop.close()
    self._thread = threading.Thread(target=run, name='grpc_service', daemon=True)
    self._thread.start()
    started.result(timeout=30)

  def shutdown(self):
    if self._server is None:
      return
    self._is_shutdown = True
    for bucket in self._fns.values():                # pending (unfilled) batches: Cancelled "Server shutdown."
      for dyn in bucket['fn']:
        dyn.shutdown()
    self._loop.call_soon_threadsafe(self._stop.set)
    self._thread.join(timeout=30)
    self._server = self._loop = self._thread = None

  def __del__(self):
    try:
      self.shutdown()
    except Exception:                                # pylint: disable=broad-except
      pass


class Client(object):
  """grpc/python/ops.py:Client -- connects (wait_for_ready), reads the method signatures from Init and exposes one
  Python method per bound function; every call is one message on ONE long-lived bidirectional `Call` stream."""

  def __init__(self, server_address, timeout=None):
    if not isinstance(server_address, str):
      raise InvalidArgumentError('server_address must be a scalar, got shape: %s' % _shape_str(np.shape(server_address)))
    # Init waits for the server (wait_for_ready), like the reference's client.  With no caller deadline the wait is cut into
    # attempts on FRESH channels: a channel that started connecting while a previous server on the same address was going
    # down can sit in its reconnect back-off long after the new server is up (seen once as a 300 s test timeout, r4).
    opts = [('grpc.max_receive_message_length', -1), ('grpc.max_send_message_length', -1)]
    deadline = None if timeout is None else time.time() + timeout
    while True:
      self._channel = grpc.insecure_channel(server_address, options=opts)
      init = self._channel.unary_unary('/%s/Init' % SERVICE, request_serializer=InitRequest.SerializeToString,
                                       response_deserializer=InitResponse.FromString)
      attempt = 10.0 if deadline is None else max(0.05, min(10.0, deadline - time.time()))
      try:
        resp = init(InitRequest(), wait_for_ready=True, timeout=attempt)
        break
      except grpc.RpcError as e:
        retry = e.code() == grpc.StatusCode.DEADLINE_EXCEEDED and (deadline is None or time.time() < deadline)
        self._channel.close()
        if not retry:
          raise UnavailableError(e.details() or 'server closed')
    self._mu = threading.Lock()
    self._queue = collections.deque()
    self._cv = threading.Condition()
    self._closed = False
    stream = self._channel.stream_stream('/%s/Call' % SERVICE, request_serializer=CallRequest.SerializeToString,
                                         response_deserializer=CallResponse.FromString)
    self._responses = stream(self._requests())
    for m in resp.method_output_signature:
      v = StructuredValue()
      v.ParseFromString(m.output_specs)
      self._add_method(m.name, decode_structure(v))

  def _requests(self):
    while True:
      with self._cv:
        while not self._queue and not self._closed:
          self._cv.wait()
        if self._closed and not self._queue:
          return
        req = self._queue.popleft()
      yield req

  def _add_method(self, name, output_specs):
    def call(*inputs):
      req = CallRequest()
      req.function = name
      for x in flatten(list(inputs)):
        req.tensor.append(encode_tensor(x))
      with self._mu:
        with self._cv:
          self._queue.append(req)
          self._cv.notify()
        try:
          resp = next(self._responses)
        except (grpc.RpcError, StopIteration):
          raise UnavailableError('Read failed, is the server closed?')
      if resp.status_code != OK:
        raise _ERR.get(resp.status_code, OpError)(resp.status_error_message, resp.status_code)
      if output_specs is None:
        return None
      return pack_sequence_as(output_specs, [decode_tensor(t)[0] for t in resp.tensor])
    setattr(self, name, call)

  def close(self):
    with self._cv:
      self._closed = True
      self._cv.notify()
    self._channel.close()


# --------------------------------------------------------------------------------------------------------------------- #
# The inference function of the learner behind the service (learner.py:339-405).
# --------------------------------------------------------------------------------------------------------------------- #
def bind_inference(server, fused_states, inference_batch_size, observation_shape, action_dtype=np.int64, lock=None,
                   observation_dtype=np.uint8, stream=None, gate=None):
  """Binds `inference(env_ids, run_ids, env_outputs, raw_rewards) -> actions` with the reference's input signature
  (learner.py:339-349: env_id int32, run_id int64, EnvOutput(reward f32, done bool, observation uint8 [...], abandoned
  bool, episode_step int32), raw_reward f32 -- every spec with the leading inference_batch_size dimension), one function
  per FusedInferenceState (the reference's one-per-inference-device list: round-robin, learner.py:406-414).  A filled
  batch is packed into one pinned host buffer (`inference.request_layout`) + the pinned frames and handed to the
  state's captured HIP graph: two H2D copies, one graph launch, one D2H of the actions.  `lock`: a lock that also orders
  the caller's own submissions to the device (a training thread dequeuing completed unrolls) against inference calls."""
  import torch
  from seed_rl_amd import inference as inf, utils
  n = inference_batch_size
  sig = (TensorSpec((n,), np.int32, 'env_id'), TensorSpec((n,), np.int64, 'run_id'),
         utils.EnvOutput(TensorSpec((n,), np.float32, 'reward'), TensorSpec((n,), np.bool_, 'done'),
                         TensorSpec((n,) + tuple(observation_shape), observation_dtype, 'observation'),
                         TensorSpec((n,), np.bool_, 'abandoned'), TensorSpec((n,), np.int32, 'episode_step')),
         TensorSpec((n,), np.float32, 'raw_reward'))
  fns = []
  for st in (fused_states if isinstance(fused_states, (list, tuple)) else [fused_states]):
    graphed = st.graphed(n, observation_shape)
    lay = inf.request_layout(n)
    req_pinned = torch.zeros(lay['bytes'], dtype=torch.uint8).pin_memory()
    t_obs = {np.dtype(np.uint8): torch.uint8, np.dtype(np.uint16): torch.int16, np.dtype(np.int16): torch.int16,
             np.dtype(np.float32): torch.float32}[np.dtype(observation_dtype)]       # env_output_specs' dtype
    obs_pinned = torch.zeros((n,) + tuple(observation_shape), dtype=t_obs).pin_memory()
    st_lock = lock if lock is not None else threading.Lock()      # `lock`: one shared with a training thread (LearnerServer)
    s_inf = stream if stream is not None else torch.cuda.current_stream(st.device)

    def inference(env_ids, run_ids, env_outputs, raw_rewards, graphed=graphed, req_pinned=req_pinned,
                  obs_pinned=obs_pinned, lock=st_lock, st=st, s_inf=s_inf):
      if gate is not None:
        gate.admit()                                 # back-pressure: the device batch must have room for this call
      with lock:                                     # one batch at a time per device state
        inf.pack_request(n, env_ids, run_ids, env_outputs.reward, raw_rewards, env_outputs.done, env_outputs.abandoned,
                         env_outputs.episode_step, out=req_pinned.numpy())
        obs_pinned.numpy()[...] = np.asarray(env_outputs.observation).view(obs_pinned.numpy().dtype)
        with torch.cuda.device(st.device), torch.cuda.stream(s_inf):
          actions = graphed.replay_packed(req_pinned, obs_pinned)
          token = gate.submitted() if gate is not None else None
          out = actions.cpu().numpy().astype(action_dtype, copy=False)      # the D2H orders after both H2D copies
      if gate is not None:
        gate.completed(token)
      return out
    inference.__name__ = 'inference'
    fns.append(function(sig, TensorSpec((n,), action_dtype, 'action'))(inference))
  server.bind(fns)
  return fns
