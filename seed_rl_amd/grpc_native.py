"""Native transport front-end: the `Server` surface of /root/reference/grpc/python/ops.py over libseedserve.so
(include/seedserve.h; csrc/serve/seedserve.cpp = the reference's grpc/ops/grpc.cc rebuilt on epoll + libnghttp2).

The wire protocol, batching rules, status codes and error strings are those of `grpc_service.Server` (the asyncio
implementation, which stays as the general-purpose one: string tensors, non-batchable functions, nests with unknown
dimensions); what moves to C++ is everything per MESSAGE: HTTP/2, gRPC framing, CallRequest / TensorProto parsing,
argument verification, row reservation, the one copy of the tensor bytes into the batch buffer, response encoding.
Python runs once per BATCH: a compute thread per bound function takes a filled slot, runs the function on views of
the slot's buffers and completes it.

  server = grpc_native.NativeServer(['unix:/tmp/seed', 'localhost:8686'])
  server.bind(inference)             # a function decorated with grpc_service.function(input_signature, ...)
  server.start(); ...; server.shutdown()

`bind_inference(server, fused_state, ...)` lays the slot buffers out as `inference.request_layout` in PINNED memory, so
a filled batch goes to the device with two asynchronous copies and one HIP-graph replay (learner.py:339-414).
"""
import ctypes
import os
import queue
import threading
import time

import numpy as np

from seed_rl_amd import grpc_service as gs

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'lib', 'libseedserve.so')
ABI_VERSION = 1
MAX_RANK = 8


class Spec(ctypes.Structure):
  """seedserve_spec."""
  _fields_ = [('dtype', ctypes.c_int32), ('rank', ctypes.c_int32), ('dims', ctypes.c_int64 * MAX_RANK),
              ('widen_to_int64', ctypes.c_int32)]


class Stats(ctypes.Structure):
  """seedserve_stats."""
  _fields_ = [(n, ctypes.c_uint64) for n in 'connections streams calls batches bytes_in bytes_out errors'.split()]


P, c_int = ctypes.c_void_p, ctypes.c_int
SIGNATURES = {
    'seedserve_last_error': (ctypes.c_char_p, []),
    'seedserve_abi_version': (c_int, []),
    'seedserve_create': (P, [c_int]),
    'seedserve_listen': (c_int, [P, ctypes.c_char_p]),
    'seedserve_bind': (c_int, [P, ctypes.c_char_p, c_int, P, c_int, P, c_int, P, P]),
    'seedserve_set_init_response': (c_int, [P, P, ctypes.c_size_t]),
    'seedserve_start': (c_int, [P]),
    'seedserve_next_batch': (c_int, [P, c_int, c_int]),
    'seedserve_complete': (c_int, [P, c_int, c_int, c_int, ctypes.c_char_p]),
    'seedserve_shutdown': (c_int, [P]),
    'seedserve_destroy': (None, [P]),
    'seedserve_get_stats': (c_int, [P, P]),
}
_lib = None


def lib():
  """Loads libseedserve.so once (built by seed_rl_amd.build); no Python fallback is substituted silently."""
  global _lib
  if _lib is None:
    if not os.path.exists(LIB_PATH):
      raise RuntimeError('libseedserve.so not built (%s). Run `python -m seed_rl_amd.build`.' % LIB_PATH)
    l = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
      fn = getattr(l, name)
      fn.restype, fn.argtypes = res, args
    if l.seedserve_abi_version() != ABI_VERSION:
      raise RuntimeError('%s reports ABI version %d, this binding was written against %d: rebuild it'
                         % (LIB_PATH, l.seedserve_abi_version(), ABI_VERSION))
    _lib = l
  return _lib


def _err():
  m = lib().seedserve_last_error()
  return m.decode() if m else ''


def _spec(shape, dtype_enum, widen=False):
  if len(shape) > MAX_RANK:
    raise ValueError('tensors of rank > %d are not supported by the native server' % MAX_RANK)
  s = Spec()
  s.dtype, s.rank, s.widen_to_int64 = int(dtype_enum), len(shape), int(bool(widen))
  for i, d in enumerate(shape):
    s.dims[i] = int(d)
  return s


class _Bound(object):
  __slots__ = ('fn_id', 'name', 'compute', 'thread', 'keep', 'inflight', 'finisher')


class NativeServer(object):
  """grpc/python/ops.py:Server -- `NativeServer([addresses])`, `bind(fn | [fn, ...])`, `start()`, `shutdown()`."""

  def __init__(self, server_addresses, num_io_threads=None):
    if isinstance(server_addresses, (str, bytes)) or not hasattr(server_addresses, '__iter__'):
      raise gs.InvalidArgumentError('server_addresses must be a vector, got shape: []')
    n = num_io_threads or max(2, min(16, (os.cpu_count() or 4) // 4))
    self._h = lib().seedserve_create(n)
    if not self._h:
      raise RuntimeError('seedserve_create failed: %s' % _err())
    self.ports = []
    for a in server_addresses:
      port = lib().seedserve_listen(self._h, a.encode())
      if port < 0:
        msg = _err()
        lib().seedserve_destroy(self._h)
        self._h = None
        raise gs.UnavailableError(msg)
      self.ports.append(port)
    self._bound, self._specs, self._names = [], [], set()
    self._started = self._stopped = False

  # ---- binding ---------------------------------------------------------------------------------------------------- #
  def bind_buffers(self, name, in_specs, out_specs, num_slots, in_ptrs, out_ptrs, compute, output_nest=None, keep=None):
    """Low level: `in_specs` / `out_specs` lists of (shape, DataType enum[, widen]) with the leading batch dimension,
    `in_ptrs[slot][i]` / `out_ptrs[slot][o]` addresses of caller-owned buffers, `compute(slot)` run by a dedicated
    thread for every filled slot (raise grpc_service.OpError / any exception to fail the batch's calls)."""
    if self._started:
      raise gs.InvalidArgumentError('Server is already started')
    ins = (Spec * len(in_specs))(*[_spec(*s) for s in in_specs])
    outs = (Spec * max(len(out_specs), 1))(*[_spec(*s) for s in out_specs])
    ip = (P * (num_slots * len(in_specs)))(*[p for slot in in_ptrs for p in slot])
    op = (P * max(num_slots * len(out_specs), 1))(*[p for slot in out_ptrs for p in slot])
    fid = lib().seedserve_bind(self._h, name.encode(), len(in_specs), ins, len(out_specs), outs, num_slots, ip, op)
    if fid < 0:
      raise gs.InvalidArgumentError(_err())
    b = _Bound()
    b.fn_id, b.name, b.compute, b.thread, b.keep = fid, name, compute, None, keep
    b.inflight, b.finisher = queue.Queue(), None
    self._bound.append(b)
    if name not in self._names:
      self._names.add(name)
      self._specs.append((name, output_nest))
    return fid

  def bind(self, fn, num_slots=4):
    """server.bind(fn) for batchable functions with numeric tensors: the function is called once per full batch with
    numpy views of the batch buffers (packed like its input_signature) and returns the batched outputs."""
    fns = list(fn) if isinstance(fn, (list, tuple)) else [fn]
    for i, f in enumerate(fns):
      sig = getattr(f, 'input_signature', None)
      if sig is None:
        raise ValueError('the bound function must have input_signature set (grpc_service.function)')
      name = f.__name__
      if i == 0 and name in self._names:
        raise gs.InvalidArgumentError("Function '%s' was bound twice." % name)
      specs = gs.flatten(sig)
      types = [gs.dtype_enum(s.dtype) for s in specs]
      shapes = [tuple(s.shape) for s in specs]
      if any(t == gs.DT_STRING for t in types) or not shapes or any(len(s) == 0 or s[0] != shapes[0][0] for s in shapes):
        raise gs.InvalidArgumentError('NativeServer binds batchable functions over numeric tensors (every input leads '
                                      'with the same batch dimension); use grpc_service.Server for %s' % name)
      N = shapes[0][0]
      out = getattr(f, 'output_signature', 'infer')
      if isinstance(out, str) and out == 'infer':
        zeros = [np.zeros(s, dtype=gs._NP_OF[t]) for t, s in zip(types, shapes)]   # pylint: disable=protected-access
        out = gs._map_to_specs(f(*gs.pack_sequence_as(sig, zeros)))                  # pylint: disable=protected-access
      ospecs = gs.flatten(out)
      for s in ospecs:
        if s.shape is None or len(s.shape) == 0 or s.shape[0] != N or any(d is None for d in s.shape):
          raise gs.InvalidArgumentError('NativeServer needs fully known output shapes leading with the batch dimension')
      otypes = [gs.dtype_enum(s.dtype) for s in ospecs]
      ibufs = [[np.zeros(s, dtype=gs._NP_OF[t]) for t, s in zip(types, shapes)] for _ in range(num_slots)]   # pylint: disable=protected-access
      obufs = [[np.zeros(s.shape, dtype=gs._NP_OF[t]) for t, s in zip(otypes, ospecs)] for _ in range(num_slots)]   # pylint: disable=protected-access

      def compute(slot, f=f, sig=sig, ibufs=ibufs, obufs=obufs, ospecs=ospecs, N=N):
        res = f(*gs.pack_sequence_as(sig, ibufs[slot]))
        flat = gs.flatten(res) if ospecs else []
        for o, dst in zip(flat, obufs[slot]):
          a = np.asarray(o.detach().cpu().numpy() if hasattr(o, 'detach') else o)
          if a.ndim <= 0:
            raise gs.InvalidArgumentError('Output must be at least rank 1 when batching is enabled')
          if a.shape[0] != N:
            raise gs.InvalidArgumentError('All outputs must have the same batch size as the inputs when batching is '
                                          'enabled, expected: %d was: %d' % (N, a.shape[0]))
          dst[...] = a
      self.bind_buffers(name, list(zip(shapes, types)), [(tuple(s.shape), t) for s, t in zip(ospecs, otypes)], num_slots,
                        [[a.ctypes.data for a in slot] for slot in ibufs],
                        [[a.ctypes.data for a in slot] for slot in obufs], compute, output_nest=out, keep=(ibufs, obufs))

  # ---- life cycle --------------------------------------------------------------------------------------------------- #
  def _guarded(self, b, slot, fn, *a):
    """Runs one submission step of a batch; a failure answers the batch's callers and returns None."""
    try:
      return fn(*a), True
    except gs.OpError as e:
      code, msg = e.code, str(e.message).encode()
    except Exception as e:                           # pylint: disable=broad-except
      code = gs.INVALID_ARGUMENT if isinstance(e, (ValueError, AssertionError)) else gs.INTERNAL
      msg = ('%s: %s' % (type(e).__name__, e)).encode()
    lib().seedserve_complete(self._h, b.fn_id, slot, code, msg)
    return None, False

  def _compute_loop(self, b):
    """One thread per bound function: takes filled slots and submits them.  A `compute` with .stage / .launch halves
    (bind_inference) is run one batch ahead: when the next full batch is already queued, its host->device copies are
    issued before the current batch's launch; when nothing is queued the current batch is launched at once."""
    l, h = lib(), self._h
    stage, launch = getattr(b.compute, 'stage', None), getattr(b.compute, 'launch', None)
    would_block = getattr(b.compute, 'would_block', None)
    pending = None                                   # (slot, token): staged, not launched

    def launch_pending():
      slot, token = pending
      finish, ok = self._guarded(b, slot, launch, token)
      if ok:
        b.inflight.put((slot, finish))

    while True:
      slot = l.seedserve_next_batch(h, b.fn_id, 0 if pending is not None else 200)
      if slot == -2:
        return
      if slot < 0:
        if pending is not None:
          launch_pending()
          pending = None
        continue
      if stage is not None:
        if pending is not None and would_block is not None and would_block():
          launch_pending()                           # do not hold an admitted batch while the next one waits at the gate
          pending = None
        token, ok = self._guarded(b, slot, stage, slot, 0 if pending is None else 1)
        if pending is not None:
          launch_pending()
        pending = (slot, token) if ok else None
        continue
      finish, ok = self._guarded(b, slot, b.compute, slot)   # None: done; a callable: finishes the batch later
      if not ok:
        continue
      if finish is None:
        l.seedserve_complete(h, b.fn_id, slot, gs.OK, b'')
      else:
        b.inflight.put((slot, finish))

  def _finish_loop(self, b):
    """Completes pipelined batches in submission order: the compute thread keeps SUBMITTING the next batches to the
    device while this thread waits for the oldest one's actions and answers its callers."""
    l, h = lib(), self._h
    while True:
      item = b.inflight.get()
      if item is None:
        return
      slot, finish = item
      code, msg = gs.OK, b''
      try:
        finish()
      except Exception as e:                         # pylint: disable=broad-except
        code, msg = gs.INTERNAL, ('%s: %s' % (type(e).__name__, e)).encode()
      l.seedserve_complete(h, b.fn_id, slot, code, msg)

  def start(self):
    if not self._bound:
      raise gs.UnavailableError('No function was bound')
    if self._started:
      raise gs.InvalidArgumentError('Server is already started')
    resp = gs.InitResponse()
    for name, specs in self._specs:
      sig = resp.method_output_signature.add()
      sig.name = name
      sig.output_specs = gs.encode_structure(specs).SerializeToString()
    blob = resp.SerializeToString()
    lib().seedserve_set_init_response(self._h, blob, len(blob))
    if lib().seedserve_start(self._h) != 0:
      raise gs.UnavailableError(_err())
    self._started = True
    for b in self._bound:
      b.thread = threading.Thread(target=self._compute_loop, args=(b,), name='seedserve_compute_%s' % b.name, daemon=True)
      b.finisher = threading.Thread(target=self._finish_loop, args=(b,), name='seedserve_finish_%s' % b.name, daemon=True)
      b.thread.start()
      b.finisher.start()

  def stats(self):
    s = Stats()
    lib().seedserve_get_stats(self._h, ctypes.byref(s))
    return {n: int(getattr(s, n)) for n, _ in Stats._fields_}

  def shutdown(self):
    if self._h is None or self._stopped:
      return
    self._stopped = True
    lib().seedserve_shutdown(self._h)
    for b in self._bound:
      if b.thread is not None:
        b.thread.join(timeout=10)
        b.inflight.put(None)
        b.finisher.join(timeout=10)

  def __del__(self):
    try:
      self.shutdown()
      if self._h is not None:
        lib().seedserve_destroy(self._h)
        self._h = None
    except Exception:                                # pylint: disable=broad-except
      pass


# --------------------------------------------------------------------------------------------------------------------- #
# The learner's inference function behind the native server (agents/vtrace/learner.py:339-414).
# --------------------------------------------------------------------------------------------------------------------- #
def inference_signature(n, observation_shape, observation_dtype=np.uint8):
  """learner.py:339-349 with the leading inference batch dimension."""
  from seed_rl_amd import utils
  T = gs.TensorSpec
  return (T((n,), np.int32, 'env_id'), T((n,), np.int64, 'run_id'),
          utils.EnvOutput(T((n,), np.float32, 'reward'), T((n,), np.bool_, 'done'),
                          T((n,) + tuple(observation_shape), observation_dtype, 'observation'),
                          T((n,), np.bool_, 'abandoned'), T((n,), np.int32, 'episode_step')),
          T((n,), np.float32, 'raw_reward'))


import collections as _collections
PROF = _collections.defaultdict(float)      # host seconds per phase of the two-phase submission (tools/bench_serving.py)


def bind_inference(server, fused_states, inference_batch_size, observation_shape, action_dtype=np.int64, num_slots=4,
                   observation_dtype=np.uint8, stream=None, gate=None, lock=None, pipeline=2):
  """Binds `inference(env_ids, run_ids, env_outputs, raw_rewards) -> actions`, one instance per FusedInferenceState
  (the reference's one-per-inference-device list, round-robin: learner.py:406-414).  Each slot is ONE pinned byte
  buffer in `inference.request_layout` (the C++ side writes every argument at its offset, env ids widened to int64)
  plus the pinned observations: a filled batch costs two H2D copies, one HIP-graph replay and one D2H of the actions,
  all on `stream` (default: a high-priority stream per state, so that inference runs beside the train step instead of
  queueing behind it).  `gate` (learner_server.BatchGate): back-pressure -- `admit()` blocks while the device batch
  could overflow (the reference blocks in unroll_queue.enqueue_many), `submitted()` / `completed(token)` keep its fill
  estimate exact.  `lock`: held
  while a batch is SUBMITTED to the stream (not while it runs): a training thread that dequeues completed unrolls on
  the same stream takes it too, so that its count read and column moves are not interleaved with a batch.
  `pipeline`: graph instances with their own static inputs.  1: copies, replay and read-back on the one inference
  stream.  2 (default): the copies of batch i+1 go to a copy stream while the graph of batch i runs, the two streams
  ordered by HOST waits of the submitting thread -- device-side cross-stream waits in front of the graph launches stalled
  that thread for the whole copy (1.30 M env-steps/s against 2.0 M on one stream and 2.2-2.26 M host-ordered, learner
  training alongside on MI355X / ROCm 7: tools/bench_serving.py).  >= 3 (LearnerServer's default): the compute thread
  additionally stages batch i+1 BEFORE it launches batch i whenever batch i+1 is already queued
  (NativeServer._compute_loop), taking the PCIe copy off the submitting thread's critical path: equal to 2 up to
  2 048-row batches (the device is the bound there), 2.9 -> 3.6 M env-steps/s at 4 096 rows (28.9 MB per copy)."""
  import torch
  from seed_rl_amd import inference as inf
  n = inference_batch_size
  sig = inference_signature(n, observation_shape, observation_dtype)
  order = ['ids', 'runs', 'reward', 'done', None, 'abandoned', 'episode_step', 'raw']     # flattened signature order
  lay = inf.request_layout(n)
  t_obs = {np.dtype(np.uint8): torch.uint8, np.dtype(np.uint16): torch.int16, np.dtype(np.int16): torch.int16,
           np.dtype(np.float32): torch.float32}[np.dtype(observation_dtype)]
  specs = gs.flatten(sig)
  in_specs = [(tuple(s.shape), gs.dtype_enum(s.dtype), s.name == 'env_id') for s in specs]
  out_specs = [((n,), gs.dtype_enum(action_dtype))]
  fids = []
  for st in (fused_states if isinstance(fused_states, (list, tuple)) else [fused_states]):
    dev = st.device
    with torch.cuda.device(dev):
      s_inf = stream or torch.cuda.Stream(device=dev, priority=-1)
      s_copy = torch.cuda.Stream(device=dev, priority=-1) if pipeline > 1 else s_inf
      with torch.cuda.stream(s_inf):
        graphs = [st.graphed(n, observation_shape, input_slot=k) for k in range(pipeline)]
      s_inf.synchronize()
    req = [torch.zeros(lay['bytes'], dtype=torch.uint8).pin_memory() for _ in range(num_slots)]
    obs = [torch.zeros((n,) + tuple(observation_shape), dtype=t_obs).pin_memory() for _ in range(num_slots)]
    act = [torch.zeros(n, dtype=torch.int64).pin_memory() for _ in range(num_slots)]
    out = [np.zeros(n, dtype=action_dtype) for _ in range(num_slots)]
    in_ptrs = [[(obs[k].data_ptr() if name is None else req[k].data_ptr() + lay[name][0]) for name in order]
               for k in range(num_slots)]
    out_ptrs = [[(act[k].data_ptr() if np.dtype(action_dtype) == np.int64 else out[k].ctypes.data)] for k in range(num_slots)]
    st_lock = lock if lock is not None else threading.Lock()
    done = [torch.cuda.Event() for _ in range(num_slots)]
    staged = [torch.cuda.Event() for _ in range(pipeline)]
    ran = [None] * pipeline                          # event after the last replay of graph k (its inputs are free again)
    counter = [0]

    def stage(slot, ahead=0, graphs=graphs, req=req, obs=obs, st=st, s_copy=s_copy, staged=staged, ran=ran,
              counter=counter):
      """First half of a batch: its host->device copies, on the copy stream.  They wait only for the previous replay of
      the SAME graph instance (by the host: device-side cross-stream waits in front of graph launches were measured to
      stall the submitting thread for the whole copy).  `ahead`: batches staged before this one and not launched yet."""
      t0 = time.perf_counter()
      if gate is not None:
        gate.admit(ahead)
      k = counter[0] % len(graphs)
      counter[0] += 1
      t1 = time.perf_counter()
      with torch.cuda.device(st.device):
        if ran[k] is not None:
          ran[k].synchronize()                       # the previous replay of this instance has read its inputs
        t2 = time.perf_counter()
        with torch.cuda.stream(s_copy):
          graphs[k].stage(req[slot], obs[slot])
          staged[k].record(s_copy)
      t3 = time.perf_counter()
      PROF['admit'] += t1 - t0; PROF['wait_ran'] += t2 - t1; PROF['stage'] += t3 - t2; PROF['calls'] += 1
      return slot, k

    def launch(token, graphs=graphs, act=act, out=out, st=st, s_inf=s_inf, st_lock=st_lock, done=done, staged=staged,
               ran=ran):
      """Second half: replay + action read-back on the inference stream; returns `finish`, which the server's
      completion thread runs to wait for the actions."""
      slot, k = token
      t0 = time.perf_counter()
      with torch.cuda.device(st.device):
        staged[k].synchronize()                      # meanwhile the previous batch's graph runs on the inference stream
        t1 = time.perf_counter()
        with st_lock:
          t2 = time.perf_counter()
          with torch.cuda.stream(s_inf):
            actions = graphs[k].launch()
            t3 = time.perf_counter()
            ev = torch.cuda.Event()
            ev.record(s_inf)
            ran[k] = ev
            act[slot].copy_(actions, non_blocking=True)
            token = gate.submitted() if gate is not None else None
            done[slot].record(s_inf)
      t4 = time.perf_counter()
      PROF['wait_staged'] += t1 - t0; PROF['lock'] += t2 - t1; PROF['graph_launch'] += t3 - t2; PROF['after'] += t4 - t3

      def finish():
        done[slot].synchronize()                     # the actions are on the host: the callers can be answered
        if gate is not None:
          gate.completed(token)
        if np.dtype(action_dtype) != np.int64:
          out[slot][...] = act[slot].numpy()
      return finish

    def one_stream(slot, graphs=graphs, req=req, obs=obs, act=act, out=out, st=st, s_inf=s_inf, st_lock=st_lock,
                   done=done, counter=counter):
      """pipeline == 1: copies, replay and read-back in order on one stream."""
      if gate is not None:
        gate.admit()
      g = graphs[counter[0] % len(graphs)]
      counter[0] += 1
      with torch.cuda.device(st.device), st_lock, torch.cuda.stream(s_inf):
        g.stage(req[slot], obs[slot])
        actions = g.launch()
        act[slot].copy_(actions, non_blocking=True)
        token = gate.submitted() if gate is not None else None
        done[slot].record(s_inf)

      def finish():
        done[slot].synchronize()
        if gate is not None:
          gate.completed(token)
        if np.dtype(action_dtype) != np.int64:
          out[slot][...] = act[slot].numpy()
      return finish

    if s_copy is s_inf:
      compute = one_stream
    else:
      def compute(slot, stage=stage, launch=launch):
        return launch(stage(slot))
      # two-phase submission: the server's compute thread stages batch i+1 BEFORE it launches batch i whenever batch
      # i+1 is already waiting, so the PCIe copy runs under the host's graph-launch time (a lone batch is never held)
      if len(graphs) >= 3:                           # with two instances staging batch i+2 would wait for batch i's replay
        compute.stage, compute.launch = stage, launch
        #  staging the next batch may wait at the gate: the compute loop asks first and launches the batch it
        # holds before it blocks -- an admitted batch's callers are never kept behind the back-pressure of the next one
        compute.would_block = (lambda: gate.would_block(1)) if gate is not None else None
    fids.append(server.bind_buffers('inference', in_specs, out_specs, num_slots, in_ptrs, out_ptrs, compute,
                                    output_nest=gs.TensorSpec((n,), action_dtype, 'action'),
                                    keep=(req, obs, act, out, graphs, s_inf, s_copy)))
  return fids
