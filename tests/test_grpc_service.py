"""The gRPC front-end (seed_rl_amd/grpc_service.py) against the behaviour the reference pins in
/root/reference/grpc/python/ops_test.py: same service / message layout (a client built from the same descriptors talks
to it over a real gRPC channel), the DynamicFn batching rules of grpc/ops/grpc.cc:591-861 and the reference's error
strings (ops_test.py:268-335, 503-630).  Runs on CPU (the transport is host code); the end-to-end path into
FusedInferenceState is in tests/test_gpu_grpc_inference.py."""
import collections
import concurrent.futures as futures
import os
import tempfile
import threading
import time
import uuid

import numpy as np
import pytest

from seed_rl_amd import grpc_service as gs
from seed_rl_amd.grpc_service import TensorSpec

Some = collections.namedtuple('Some', 'a b')


@pytest.fixture
def address():
  path = os.path.join(tempfile.gettempdir(), 'seedrl_' + uuid.uuid4().hex[:12])
  yield 'unix:' + path
  if os.path.exists(path):
    os.remove(path)


def _serve(address, *fns):
  server = gs.Server([address])
  for f in fns:
    server.bind(f)
  server.start()
  return server


def test_tensor_proto_round_trip_and_typed_fields():
  for a in (np.arange(12, dtype=np.int32).reshape(3, 4), np.float32(2.5), np.array([True, False]),
            np.zeros((0, 3), np.uint8), np.arange(5, dtype=np.int64), np.array([1.5, 2.5], np.float64)):
    b, dt = gs.decode_tensor(gs.encode_tensor(a))
    assert b.dtype == np.asarray(a).dtype and b.shape == np.shape(a) and np.array_equal(a, b)
    assert dt == gs.dtype_enum(np.asarray(a).dtype)
  s, dt = gs.decode_tensor(gs.encode_tensor(np.array([b'ab', b'', b'xyz'], dtype=object)))
  assert dt == gs.DT_STRING and list(s) == [b'ab', b'', b'xyz']
  # a TensorProto written with typed value fields (tf.make_tensor_proto style), incl. the "last value repeats" rule
  tp = gs.TensorProto()
  tp.dtype = gs.DT_FLOAT
  tp.tensor_shape.dim.add().size = 3
  tp.float_val.append(7.0)
  a, _ = gs.decode_tensor(tp.SerializeToString())
  assert a.tolist() == [7.0, 7.0, 7.0]
  tp = gs.TensorProto()
  tp.dtype = gs.DT_INT64
  tp.int64_val.extend([5])
  a, _ = gs.decode_tensor(tp.SerializeToString())
  assert a.shape == () and int(a) == 5
  # wire layout is the published one: dtype = field 1 varint, shape = field 2, content = field 4
  raw = gs.encode_tensor(np.int32(1))
  assert raw[:2] == bytes([0x08, gs.DT_INT32]) and raw[-6:] == bytes([0x22, 4, 1, 0, 0, 0])


def test_structured_value_round_trip():
  nest = (TensorSpec((), np.int32, 'arg1'), Some(TensorSpec((2,), np.float32, 'x'), [TensorSpec((None, 3), np.uint8), None]),
          {'k': TensorSpec((), 'string')})
  back = gs.decode_structure(gs.StructuredValue.FromString(gs.encode_structure(nest).SerializeToString()))
  assert back[0] == nest[0] and type(back[1]).__name__ == 'Some' and back[1].a == nest[1].a
  assert back[1].b[0].shape == (None, 3) and back[1].b[1] is None
  assert back[2]['k'].dtype == 'string'


def test_simple_two_calls_and_upvalue(address):                      # ops_test.py:41-72, 204-219
  a = 2

  @gs.function([TensorSpec((), np.int32)])
  def foo(x):
    return x + 1

  @gs.function([TensorSpec((), np.int32)])
  def half(x):
    return (x // a).astype(np.int32)
  server = _serve(address, foo, half)
  client = gs.Client(address)
  assert client.foo(42) == 43 and client.foo(43) == 44 and client.half(42) == 21
  server.shutdown()


def test_empty_input_output_no_output_string(address):               # ops_test.py:74-117, 243-256
  @gs.function([])
  def forty_two():
    return 42

  @gs.function([TensorSpec((), np.int32)])
  def empty(x):
    return []

  @gs.function([TensorSpec((), np.int32)])
  def nothing(x):
    pass

  @gs.function([TensorSpec((), 'string')])
  def hello(x):
    return x.item() + b' world'
  server = _serve(address, forty_two, empty, nothing, hello)
  client = gs.Client(address)
  assert client.forty_two() == 42
  assert client.empty(42) == []
  assert client.nothing(42) is None
  assert client.hello('hello').item() == b'hello world'
  server.shutdown()


def test_large_tensor(address):                                      # ops_test.py:119-134 (40 MB there; 16 MB here)
  t = np.ones((4, 1024, 1024), np.int32)

  @gs.function([TensorSpec(t.shape, np.int32)], TensorSpec(t.shape, np.int32))
  def foo(x):
    return x + 1
  server = _serve(address, foo)
  assert np.array_equal(gs.Client(address).foo(t), t + 1)
  server.shutdown()


def test_wait_for_server(address):                                   # ops_test.py:160-202
  @gs.function([TensorSpec((), np.int32)])
  def foo(x):
    return x + 1
  server = gs.Server([address])
  server.bind(foo)
  with futures.ThreadPoolExecutor(max_workers=1) as ex:
    f = ex.submit(lambda: gs.Client(address).foo(42))
    time.sleep(0.5)
    server.start()
    assert f.result(timeout=30) == 43
  server.shutdown()


def test_bind_and_start_errors(address):                             # ops_test.py:258-301
  with pytest.raises(gs.InvalidArgumentError, match='server_addresses must be a vector'):
    gs.Server(address)
  with pytest.raises(gs.InvalidArgumentError, match='server_address must be a scalar'):
    gs.Client([address])
  server = gs.Server([address])
  with pytest.raises(gs.UnavailableError, match='No function was bound'):
    server.start()

  @gs.function([TensorSpec((), np.int32)])
  def foo(x):
    return x + 1
  server.bind(foo)
  with pytest.raises(gs.InvalidArgumentError, match="Function 'foo' was bound twice."):
    server.bind(foo)
  server.start()
  with pytest.raises(gs.InvalidArgumentError, match='Server is already started'):
    server.start()
  server.shutdown()


def test_argument_errors(address):                                   # ops_test.py:303-354, 612-630
  @gs.function([TensorSpec((), np.int32)])
  def foo(x):
    return x + 1

  @gs.function([TensorSpec((), np.int32)], TensorSpec((), np.int32))      # (no trace call at bind: it would fail)
  def failing(x):
    assert x == 1, 'assertion failed'
    return x

  @gs.function([TensorSpec((4, 3), np.int32)], TensorSpec((4, 3), np.int32))
  def ident(x):
    return x
  server = _serve(address, foo, failing, ident)
  client = gs.Client(address)
  with pytest.raises(gs.InvalidArgumentError, match='Expects 1 arguments, but 2 is provided'):
    client.foo([42, 43])
  with pytest.raises(gs.InvalidArgumentError, match=r'Expects arg\[0\] to be int32 but string is provided'):
    client.foo('foo')
  with pytest.raises(gs.InvalidArgumentError, match='assertion failed'):
    client.failing(42)
  with pytest.raises(gs.InvalidArgumentError, match=r'Expects arg\[0\] to have shape with suffix \[3\], but had shape \[3,4\]'):
    client.ident(np.zeros((3, 4), np.int32))
  with pytest.raises(gs.InternalError, match='Function nope not found'):
    client._add_method('nope', None)
    client.nope(1)
  assert client.foo(1) == 2                                          # the stream survives error responses
  server.shutdown()


def test_nests(address):                                             # ops_test.py:356-382
  signature = (TensorSpec((), np.int32, 'arg1'),
               Some(TensorSpec((), np.int32, 'arg2'), [TensorSpec((), np.int32, 'arg3'), TensorSpec((), np.int32, 'arg4')]))

  @gs.function(signature)
  def foo(*args):
    return gs.pack_sequence_as(args, [t + 1 for t in gs.flatten(list(args))])
  server = _serve(address, foo)
  out = gs.Client(address).foo((1, Some(2, [3, 4])))
  assert isinstance(out, tuple) and type(out[1]).__name__ == 'Some' and isinstance(out[1].b, list)
  assert [int(x) for x in gs.flatten(list(out))] == [2, 3, 4, 5]
  server.shutdown()


def test_shutdown_behaviour(address):                                # ops_test.py:384-421, 483-501, 524-541
  waiting = threading.Event()

  @gs.function([TensorSpec((), np.int32)])
  def slow(x):
    waiting.set()
    time.sleep(1)
    return x + 1

  @gs.function([TensorSpec((2,), np.int32)], TensorSpec((2,), np.int32))
  def batched(x):
    return x + 1
  server = _serve(address, slow, batched)
  client = gs.Client(address)
  with futures.ThreadPoolExecutor(max_workers=1) as ex:
    f = ex.submit(client.slow, 42)
    assert waiting.wait(10)
    server.shutdown()                                                # while in a call
    with pytest.raises(gs.UnavailableError, match='server closed'):
      f.result(timeout=30)
  with pytest.raises(gs.UnavailableError, match='server closed'):    # call after shutdown
    client.slow(42)
  server.start()                                                     # shutdown + start: serves again
  client = gs.Client(address)
  with futures.ThreadPoolExecutor(max_workers=1) as ex:
    f = ex.submit(client.batched, 42)                                # half a batch: blocks
    time.sleep(0.5)
    assert not f.done()
    server.shutdown()                                                # waiting for a full batch
    with pytest.raises(gs.UnavailableError, match='server closed'):
      f.result(timeout=30)


def test_batching_rules(address):                                    # ops_test.py:503-522, 543-610, 760-801
  calls = []

  @gs.function([TensorSpec((2,), np.int32), TensorSpec((2,), np.int32)], TensorSpec((), np.int32))
  def rank0(unused_x, unused_y):
    return np.int32(1)

  @gs.function([TensorSpec((4,), np.int32)], TensorSpec((4,), np.int32))
  def foo(x):
    calls.append(x.copy())
    return x + 1

  @gs.function([TensorSpec((2,), np.int32)], TensorSpec(None, np.float32))          # shape known only when run
  def unspecified(x):
    return np.zeros((), np.float32) if x[0] == 0 else np.zeros(2 if x[0] == 1 else 1, np.float32)
  server = _serve(address, rank0, foo, unspecified)
  # no batching when an output is a scalar: the single-element call is then just a wrong-rank direct call
  with pytest.raises(gs.InvalidArgumentError, match=r'Expects arg\[0\] to have shape with 1 dimension\(s\), but had shape \[\]'):
    gs.Client(address).rank0(1, 1)
  # exact-shape arguments run directly, un-batched (batch auto-detection)
  c = gs.Client(address)
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
    assert sorted(int(f.result(timeout=30)) for f in fs) == [2, 3, 4, 5]
  assert len(calls) == 3
  # outputs checked per call when batching: rank 0 / wrong batch size
  with futures.ThreadPoolExecutor(max_workers=2) as ex:
    def bad(v, msg):
      with pytest.raises(gs.InvalidArgumentError, match=msg):
        gs.Client(address).unspecified(v)
    fs = [ex.submit(bad, 0, 'Output must be at least rank 1 when batching is enabled') for _ in range(2)]
    [f.result(timeout=30) for f in fs]
    fs = [ex.submit(bad, 2, 'All outputs must have the same batch size as the inputs when batching is enabled, '
                            'expected: 2 was: 1') for _ in range(2)]
    [f.result(timeout=30) for f in fs]
  server.shutdown()


def test_round_robin_over_bound_functions(address):                  # ops.py:80-83, grpc.cc:191-197
  def mk(tag):
    @gs.function([TensorSpec((), np.int32)])
    def which(x):
      return np.int32(tag)
    return which
  server = gs.Server([address])
  server.bind([mk(0), mk(1), mk(2)])
  server.start()
  client = gs.Client(address)
  assert [int(client.which(0)) for _ in range(7)] == [0, 1, 2, 0, 1, 2, 0]
  server.shutdown()


def test_stress(address):                                            # ops_test.py:632-664
  @gs.function([TensorSpec((5,), np.int32)], TensorSpec((5,), np.int32))
  def foo(x):
    return x + 1
  server = _serve(address, foo)
  num_clients, num_calls = 10, 50
  clients = [gs.Client(address) for _ in range(num_clients)]

  def do_calls(client):
    for i in range(num_calls):
      assert int(client.foo(i)) == i + 1
  with futures.ThreadPoolExecutor(max_workers=num_clients) as ex:
    fs = [ex.submit(do_calls, c) for c in clients]
    for i, f in enumerate(futures.as_completed(fs)):
      try:
        f.result()
      except gs.UnavailableError:
        assert i > num_clients // 2                 # only the clients cut off by the shutdown below
      if i == num_clients // 2:
        # as in the reference test: shut down once half the clients are through -- the last batch may never fill
        server.shutdown()
