"""The NATIVE transport front-end (libseedserve.so through seed_rl_amd/grpc_native.py) against the behaviour the
reference pins in /root/reference/grpc/python/ops_test.py, for the part of the surface it covers (batchable functions
over numeric tensors -- the inference path): same service / message layout (the grpcio `Client` built from the same
descriptors talks to it over real sockets), DynamicFn batching (grpc/ops/grpc.cc:591-861), verify_args error strings
(:527-589), round-robin over functions bound under one name (:193-205), Init signatures, shutdown semantics.
Host code only: runs on CPU."""
import collections
import concurrent.futures as futures
import os
import tempfile
import time
import uuid

import numpy as np
import pytest

from seed_rl_amd import grpc_native as gn
from seed_rl_amd import grpc_service as gs
from seed_rl_amd.grpc_service import TensorSpec


@pytest.fixture
def address():
  path = os.path.join(tempfile.gettempdir(), 'seedrl_n_' + uuid.uuid4().hex[:12])
  yield 'unix:' + path
  if os.path.exists(path):
    os.remove(path)


def _serve(address, *fns, **kw):
  server = gn.NativeServer([address], num_io_threads=2)
  for f in fns:
    server.bind(f, **kw)
  server.start()
  return server


def test_library_exports_header_symbols():
  import re
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  hdr = open(os.path.join(root, 'include', 'seedserve.h')).read()
  declared = sorted(set(re.findall(r'\b(seedserve_[a-z_0-9]+)\s*\(', hdr)))
  assert declared == sorted(gn.SIGNATURES) and len(declared) == 12
  l = gn.lib()
  for s in declared:
    assert hasattr(l, s)
  assert l.seedserve_abi_version() == gn.ABI_VERSION


def test_batched_calls_and_slices(address):                          # ops_test.py:543-610, 760-801
  calls = []

  @gs.function([TensorSpec((4,), np.int32)], TensorSpec((4,), np.int32))
  def foo(x):
    calls.append(x.copy())
    return x + 1
  server = _serve(address, foo)
  c = gs.Client(address)
  # exact-shape arguments run directly, as a computation of their own
  assert c.foo(np.array([1, 2, 3, 4], np.int32)).tolist() == [2, 3, 4, 5] and len(calls) == 1
  # client-side batches of 2 from two clients fill ONE server-side batch of 4; each gets its slice back
  with futures.ThreadPoolExecutor(max_workers=2) as ex:
    f1 = ex.submit(lambda: gs.Client(address).foo(np.array([42, 43], np.int32)))
    f2 = ex.submit(lambda: gs.Client(address).foo(np.array([142, 143], np.int32)))
    assert f1.result(timeout=30).tolist() == [43, 44] and f2.result(timeout=30).tolist() == [143, 144]
  assert len(calls) == 2 and sorted(calls[1].tolist()) == [42, 43, 142, 143]
  # four single-element calls (rank - 1 arguments) -> one batch; scalars come back
  with futures.ThreadPoolExecutor(max_workers=4) as ex:
    fs = [ex.submit(lambda v=v: gs.Client(address).foo(v)) for v in (1, 2, 3, 4)]
    res = [f.result(timeout=30) for f in fs]
    assert sorted(int(r) for r in res) == [2, 3, 4, 5] and all(np.shape(r) == () for r in res)
  assert len(calls) == 3
  st = server.stats()
  assert st['batches'] == 3 and st['calls'] == 7 and st['errors'] == 0
  server.shutdown()


def test_argument_errors(address):                                   # ops_test.py:303-354, 612-630; grpc.cc:527-589
  @gs.function([TensorSpec((2, 3), np.int32), TensorSpec((2,), np.float32)], TensorSpec((2,), np.int32))
  def ident(x, y):
    return x[:, 0]

  @gs.function([TensorSpec((1,), np.int32)], TensorSpec((1,), np.int32))
  def failing(x):
    assert x[0] == 1, 'assertion failed'
    return x
  server = _serve(address, ident, failing)
  client = gs.Client(address)
  with pytest.raises(gs.InvalidArgumentError, match='Expects 2 arguments, but 1 is provided'):
    _raw_call(client, _req('ident', np.zeros(3, np.int32)))
  with pytest.raises(gs.InvalidArgumentError, match=r'Expects arg\[1\] to be float but int32 is provided'):
    client.ident(np.zeros(3, np.int32), np.int32(1))
  with pytest.raises(gs.InvalidArgumentError,
                     match=r'Expects arg\[0\] to have shape with suffix \[3\], but had shape \[4\]'):
    client.ident(np.zeros(4, np.int32), np.float32(1))
  with pytest.raises(gs.InvalidArgumentError,
                     match=r'Expects arg\[0\] to have shape with 1 dimension\(s\), but had shape \[1,1,3\]'):
    client.ident(np.zeros((1, 1, 3), np.int32), np.float32(1))
  with pytest.raises(gs.InvalidArgumentError,
                     match=r'Expects arg\[1\] to start with the batching dimension 1 but had shape \[2\]'):
    _raw_call(client, _req('ident', np.zeros((1, 3), np.int32), np.zeros(2, np.float32)))
  with pytest.raises(gs.InvalidArgumentError, match='Learner-side batch size exceeded'):
    _raw_call(client, _req('ident', np.zeros((3, 3), np.int32), np.zeros(3, np.float32)))
  with pytest.raises(gs.InvalidArgumentError, match='assertion failed'):
    client.failing(42)
  with pytest.raises(gs.InternalError, match='Function nope not found'):
    client._add_method('nope', None)
    client.nope(1)
  with pytest.raises(gs.InvalidArgumentError, match='Cannot parse TensorProto.'):
    req = gs.CallRequest(); req.function = 'ident'; req.tensor.append(b'\xff\xff\xff'); req.tensor.append(b'\x08')
    _raw_call(client, req)
  # the stream survives error responses
  assert client.failing(1) == 1
  assert server.stats()['errors'] >= 8
  server.shutdown()


def _req(function, *arrays):
  req = gs.CallRequest()
  req.function = function
  for a in arrays:
    req.tensor.append(gs.encode_tensor(a))
  return req


def _raw_call(client, req):
  """One message on the client's Call stream with hand-built contents."""
  with client._mu:
    with client._cv:
      client._queue.append(req)
      client._cv.notify()
    resp = next(client._responses)
  if resp.status_code != gs.OK:
    raise gs._ERR.get(resp.status_code, gs.OpError)(resp.status_error_message, resp.status_code)
  return [gs.decode_tensor(t)[0] for t in resp.tensor]


def test_typed_value_fields_and_widening(address):
  """TensorProtos written with typed *_val fields (tf.make_tensor_proto style, "last value repeats") are accepted like
  tensor_content ones; an int32 argument bound with widen_to_int64 lands as int64 in the batch buffer."""
  seen = {}
  n = 4
  ibuf = [np.zeros(n, np.int64), np.zeros((n, 2), np.float32), np.zeros(n, np.bool_)]
  obuf = [np.zeros(n, np.int64)]
  server = gn.NativeServer([address], num_io_threads=1)

  def compute(slot):
    seen['ids'] = ibuf[0].copy(); seen['x'] = ibuf[1].copy(); seen['b'] = ibuf[2].copy()
    obuf[0][...] = ibuf[0] * 10
  server.bind_buffers('f', [((n,), gs.DT_INT32, True), ((n, 2), gs.DT_FLOAT), ((n,), gs.DT_BOOL)], [((n,), gs.DT_INT64)], 1,
                      [[a.ctypes.data for a in ibuf]], [[a.ctypes.data for a in obuf]], compute,
                      output_nest=TensorSpec((n,), np.int64))
  server.start()
  client = gs.Client(address)
  tp_id = gs.TensorProto(); tp_id.dtype = gs.DT_INT32; tp_id.tensor_shape.dim.add().size = n; tp_id.int_val.extend([7, -3])
  tp_x = gs.TensorProto(); tp_x.dtype = gs.DT_FLOAT
  tp_x.tensor_shape.dim.add().size = n; tp_x.tensor_shape.dim.add().size = 2; tp_x.float_val.extend([1.5, 2.5, 3.5])
  tp_b = gs.TensorProto(); tp_b.dtype = gs.DT_BOOL; tp_b.tensor_shape.dim.add().size = n; tp_b.bool_val.extend([True, False, True])
  req = gs.CallRequest(); req.function = 'f'
  for tp in (tp_id, tp_x, tp_b):
    req.tensor.append(tp.SerializeToString())
  out, = _raw_call(client, req)
  assert seen['ids'].tolist() == [7, -3, -3, -3] and seen['ids'].dtype == np.int64
  assert seen['x'].reshape(-1).tolist() == [1.5, 2.5, 3.5, 3.5, 3.5, 3.5, 3.5, 3.5]
  assert seen['b'].tolist() == [True, False, True, True]
  assert out.tolist() == [70, -30, -30, -30] and out.dtype == np.int64
  server.shutdown()


def test_malformed_payload_never_reaches_a_batch(address):
  """ADVICE r3 (seedserve.cpp place()): a TensorProto that parses as a message but whose BYTES are wrong -- tensor_content
  of the wrong length, a truncated float_val run -- is answered INVALID_ARGUMENT 'Cannot parse TensorProto.' (the
  reference's Tensor::FromProto failure, grpc.cc:176-182) BEFORE any row is reserved: the batch it would have joined is
  later filled by well-formed calls only and computes on their bytes, not on whatever an earlier batch left in the
  pinned buffer."""
  import threading
  n = 2
  ibuf = [np.zeros((n, 3), np.float32)]
  obuf = [np.zeros(n, np.float32)]
  seen = []
  server = gn.NativeServer([address], num_io_threads=1)

  def compute(slot):
    seen.append(ibuf[0].copy())
    obuf[0][...] = ibuf[0].sum(axis=1)
  server.bind_buffers('f', [((n, 3), gs.DT_FLOAT)], [((n,), gs.DT_FLOAT)], 1, [[a.ctypes.data for a in ibuf]],
                      [[a.ctypes.data for a in obuf]], compute, output_nest=TensorSpec((n,), np.float32))
  server.start()
  client, other = gs.Client(address), gs.Client(address)
  # a full batch first, so that the pinned rows hold "an earlier batch's bytes"
  out, = _raw_call(client, _req('f', np.full((2, 3), 9.0, np.float32)))
  assert out.tolist() == [27.0, 27.0]
  errors0 = server.stats()['errors']
  bad = gs.TensorProto(); bad.dtype = gs.DT_FLOAT; bad.tensor_shape.dim.add().size = 3
  bad.tensor_content = b'\x00' * 8                                   # 3 floats announced, 8 bytes sent
  req = gs.CallRequest(); req.function = 'f'; req.tensor.append(bad.SerializeToString())
  with pytest.raises(gs.InvalidArgumentError, match='Cannot parse TensorProto.'):
    _raw_call(client, req)
  trunc = gs.encode_tensor(np.zeros(3, np.float32))
  tp = gs.TensorProto(); tp.ParseFromString(trunc); tp.tensor_content = b''
  raw = tp.SerializeToString() + b'\x2a\x06' + b'\x00\x00\x80\x3f\x00\x00'   # packed float_val: one value + 2 stray bytes
  req = gs.CallRequest(); req.function = 'f'; req.tensor.append(raw)
  with pytest.raises(gs.InvalidArgumentError, match='Cannot parse TensorProto.'):
    _raw_call(client, req)
  assert server.stats()['errors'] == errors0 + 2
  # no row was reserved by the rejected calls: two single-row calls fill ONE batch with their own bytes
  res = {}
  t = threading.Thread(target=lambda: res.setdefault('b', _raw_call(other, _req('f', np.array([4.0, 5.0, 6.0], np.float32)))))
  t.start()
  a, = _raw_call(client, _req('f', np.array([1.0, 2.0, 3.0], np.float32)))
  t.join(10)
  assert sorted([float(a), float(res['b'][0])]) == [6.0, 15.0]
  assert len(seen) == 2 and sorted(seen[1].sum(axis=1).tolist()) == [6.0, 15.0]
  server.shutdown()


def test_init_signatures_nests_and_round_robin(address):             # ops.py:80-83, grpc.cc:191-205; ops_test.py:356-382
  Out = collections.namedtuple('Out', 'action value')

  def mk(tag):
    @gs.function((TensorSpec((2,), np.int32, 'x'), {'k': TensorSpec((2, 3), np.uint8, 'obs')}),
                 Out(TensorSpec((2,), np.int64, 'action'), TensorSpec((2,), np.float32, 'value')))
    def which(x, d):
      return Out(np.full(2, tag, np.int64), d['k'].sum(-1).astype(np.float32) + x)
    return which
  server = gn.NativeServer([address], num_io_threads=2)
  server.bind([mk(0), mk(1), mk(2)])
  server.start()
  client = gs.Client(address)
  obs = np.arange(6, dtype=np.uint8).reshape(2, 3)
  tags = []
  for _ in range(7):
    out = client.which(np.array([10, 20], np.int32), {'k': obs})
    assert type(out).__name__ == 'Out' and out.value.tolist() == [13.0, 32.0] and out.action.dtype == np.int64
    tags.append(int(out.action[0]))
  assert tags == [0, 1, 2, 0, 1, 2, 0]
  server.shutdown()


def test_bind_and_start_errors(address):                             # ops_test.py:258-301
  with pytest.raises(gs.InvalidArgumentError, match='server_addresses must be a vector'):
    gn.NativeServer(address)
  server = gn.NativeServer([address])
  with pytest.raises(gs.UnavailableError, match='No function was bound'):
    server.start()

  @gs.function([TensorSpec((2,), np.int32)], TensorSpec((2,), np.int32))
  def foo(x):
    return x + 1
  server.bind(foo)
  with pytest.raises(gs.InvalidArgumentError, match="Function 'foo' was bound twice."):
    server.bind(foo)

  @gs.function([TensorSpec((), np.int32)])
  def scalar(x):
    return x
  with pytest.raises(gs.InvalidArgumentError, match='batchable'):
    server.bind(scalar)
  server.start()
  with pytest.raises(gs.InvalidArgumentError, match='Server is already started'):
    server.start()
  assert int(gs.Client(address).foo(np.array([1, 2], np.int32))[1]) == 3
  server.shutdown()
  with pytest.raises(gs.UnavailableError):
    gn.NativeServer(['unix:/nonexistent_dir_%s/sock' % uuid.uuid4().hex])


def test_tcp_listener_and_large_tensor():                            # ops_test.py:119-134 (40 MB there; 24 MB here)
  t = np.arange(6 * 1024 * 1024, dtype=np.int32).reshape(2, 3, 1024, 1024)

  @gs.function([TensorSpec(t.shape, np.int32)], TensorSpec(t.shape, np.int32))
  def foo(x):
    return x + 1
  server = gn.NativeServer(['localhost:0'], num_io_threads=1)
  server.bind(foo, num_slots=1)
  server.start()
  assert server.ports[0] > 0
  client = gs.Client('localhost:%d' % server.ports[0])
  assert np.array_equal(client.foo(t), t + 1)
  # one row at a time: two calls of 12 MB fill the batch
  with futures.ThreadPoolExecutor(max_workers=2) as ex:
    c2 = gs.Client('localhost:%d' % server.ports[0])
    f1, f2 = ex.submit(client.foo, t[0]), ex.submit(c2.foo, t[1])
    r1, r2 = f1.result(timeout=60), f2.result(timeout=60)
  assert {int(r1[0, 0, 0]), int(r2[0, 0, 0])} == {1, int(t[1, 0, 0, 0]) + 1}
  server.shutdown()


def test_shutdown_behaviour(address):                                # ops_test.py:384-421, 483-501, 524-541
  @gs.function([TensorSpec((2,), np.int32)], TensorSpec((2,), np.int32))
  def batched(x):
    return x + 1
  server = _serve(address, batched)
  client = gs.Client(address)
  with futures.ThreadPoolExecutor(max_workers=1) as ex:
    f = ex.submit(client.batched, 42)                                # half a batch: blocks
    time.sleep(0.5)
    assert not f.done()
    server.shutdown()                                                # waiting for a full batch
    with pytest.raises(gs.UnavailableError, match='server closed'):
      f.result(timeout=30)
  with pytest.raises(gs.UnavailableError, match='server closed'):    # call after shutdown
    client.batched(42)
  server.shutdown()                                                  # idempotent


def test_more_calls_than_slots_queue_up(address):
  """Calls that find every slot busy wait for one (the reference blocks in its queue); nothing is dropped."""
  @gs.function([TensorSpec((1,), np.int32)], TensorSpec((1,), np.int32))
  def slow(x):
    time.sleep(0.05)
    return x + 1
  server = _serve(address, slow, num_slots=1)
  with futures.ThreadPoolExecutor(max_workers=8) as ex:
    fs = [ex.submit(lambda v=v: int(gs.Client(address).slow(v))) for v in range(8)]
    assert sorted(f.result(timeout=60) for f in fs) == list(range(1, 9))
  server.shutdown()


def test_stress(address):                                            # ops_test.py:632-664
  """Many streams, single-step calls with a frame each, batches of 5 filled by 5 lock-stepped clients at a time (two
  such groups on two functions so that batches of different functions interleave on the I/O threads)."""
  def mk(name):
    @gs.function([TensorSpec((5,), np.int32), TensorSpec((5, 64, 64), np.uint8)], TensorSpec((5,), np.int32))
    def fn(x, frames):
      return x + frames[:, 0, 0]
    fn.__name__ = name
    return fn
  server = _serve(address, mk('foo'), mk('bar'))
  num_clients, num_calls = 10, 100
  clients = [gs.Client(address) for _ in range(num_clients)]

  def do_calls(k, client):
    frame = np.full((64, 64), k, np.uint8)
    call = client.foo if k < 5 else client.bar
    for i in range(num_calls):
      assert int(call(i, frame)) == i + k
  with futures.ThreadPoolExecutor(max_workers=num_clients) as ex:
    fs = [ex.submit(do_calls, k, c) for k, c in enumerate(clients)]
    for f in fs:
      f.result(timeout=120)
  st = server.stats()
  assert st['calls'] == num_clients * num_calls and st['batches'] == num_clients * num_calls // 5 and st['errors'] == 0
  # grpcio pools the channels of one process onto one connection: the ten clients are ten pairs of HTTP/2 streams
  assert st['connections'] >= 1 and st['streams'] == 2 * num_clients
  server.shutdown()


def test_many_connections_large_messages(address):
  """Twelve CONNECTIONS (one channel each: grpc.use_local_subchannel_pool) spread over three I/O threads, 150 calls of
  3 rows x 64 KB each per connection: exercises the cross-thread response path (12 pending answers per batch, posted from
  the compute thread to three I/O threads) and flow control of large DATA frames.  Every caller must get ITS rows back."""
  import queue
  import grpc
  N, k, C, calls = 36, 3, 12, 150              # every batch = one call of every connection (closed loop, lock step)
  @gs.function([TensorSpec((N,), np.int32), TensorSpec((N, 65536), np.uint8)], TensorSpec((N,), np.int64))
  def foo(x, payload):
    return x.astype(np.int64) * 1000 + payload[:, -1]
  server = gn.NativeServer([address], num_io_threads=3)
  server.bind(foo, num_slots=2)
  server.start()
  errors = []

  def conn(cid):
    try:
      ch = grpc.insecure_channel(address, options=[('grpc.max_receive_message_length', -1),
                                                   ('grpc.max_send_message_length', -1),
                                                   ('grpc.use_local_subchannel_pool', 1)])
      ident = lambda b: b
      ch.unary_unary('/%s/Init' % gs.SERVICE, request_serializer=ident, response_deserializer=ident)(
          b'', wait_for_ready=True, timeout=60)
      q = queue.SimpleQueue()

      def gen():
        while True:
          item = q.get()
          if item is None:
            return
          yield item
      responses = ch.stream_stream('/%s/Call' % gs.SERVICE, request_serializer=ident, response_deserializer=ident)(gen())
      payload = np.zeros((k, 65536), np.uint8)
      for i in range(calls):
        x = np.array([cid * 10 + j for j in range(k)], np.int32)
        payload[:, -1] = (i + cid) % 251
        q.put(_req('foo', x, payload).SerializeToString())
        resp = gs.CallResponse.FromString(next(responses))
        assert resp.status_code == 0, resp.status_error_message
        out = gs.decode_tensor(resp.tensor[0])[0]
        assert out.tolist() == [int(v) * 1000 + (i + cid) % 251 for v in x], (cid, i, out.tolist())
      q.put(None)
      ch.close()
    except Exception as e:                            # pylint: disable=broad-except
      errors.append((cid, repr(e)))
  with futures.ThreadPoolExecutor(max_workers=C) as ex:
    list(ex.map(conn, range(C)))
  st = server.stats()
  server.shutdown()
  assert not errors, errors[:3]
  assert st['connections'] == C and st['calls'] == C * calls and st['batches'] == C * calls * k // N and st['errors'] == 0
