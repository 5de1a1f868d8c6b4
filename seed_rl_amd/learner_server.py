"""The data path of the reference's `learner_loop` around the hot path (agents/vtrace/learner.py:170-483):
actors --gRPC--> server-side batching --> central inference on the device store --> completed unrolls (time-major, in
HBM) --> dequeue(batch_size) --> Learner.minimize.  No checkpoint manager, logger thread or tf.data pipeline (control
plane, out of scope: SURVEY.md section 2) -- what is here is exactly what turns the pieces into a servable learner:

  server = LearnerServer(agent, learner, unroll_length=20, batch_size=512, inference_batch_size=256, num_envs=1024,
                         observation_shape=(84, 84, 1), server_addresses=['unix:/tmp/seed', 'localhost:8686'])
  server.start()
  while training: loss, session = server.train_step()     # blocks until batch_size unrolls are complete
  server.shutdown()

or, with the reference's entry-point signature (learner.py:170-187):

  learner_loop(create_env_fn, create_agent_fn, create_optimizer_fn, config=LoopConfig(...))

Concurrency (round 3).  Inference and training share ONE GPU the way the reference shares a TPU host between its
inference and training cores: inference batches run on a HIGH-PRIORITY stream through an `inference_twin()` of the
agent (same parameter buffer, own workspaces), the train step on its own stream, replayed from a HIP graph -- nothing
serialises a whole train step against inference any more.  The only ordered hand-over is the dequeue: it is submitted
to the inference stream under the lock that also orders the inference submissions (the column copies and the count
update, no host read: BatchGate), writes into one of two static training unrolls, and the train stream waits for its event.  The
transport is the native front-end (grpc_native / libseedserve.so) unless transport='python'.
"""
import collections
import threading
import time

import torch

from seed_rl_amd import grpc_service, inference, learner as learner_lib, networks, utils
from seed_rl_amd.unroll_store import Spec


class BatchGate(object):
  """Back-pressure between central inference and the learner on one device batch (the reference blocks in
  `unroll_queue.enqueue_many` when its queue of capacity 1 is full, learner.py:325-327,396-397) WITHOUT any blocking
  read of the device's column count (r6; until r5 the dequeue read it, which waited for every replay in flight on the
  inference stream while holding the submission lock: 70-120 us per inference call of the closed loop).

  Everything that changes the count is SUBMITTED in stream order under `lock` and numbered: an inference batch of n rows
  adds between 0 and n columns, a dequeue removes exactly B.  Each batch copies the count into its own pinned word
  behind itself; when its event has completed, that word is the exact count after submission number `base_seq`.  Then
    lower bound = base_fill - B * (dequeues after base_seq)                   -- what the learner may rely on,
    upper bound = lower bound + n * (batches after base_seq)                  -- what admission must assume,
  and both are exact whenever nothing is in flight."""

  def __init__(self, state, n, lock, ring=1024, mirrors=None):
    self.state, self.n, self.lock = state, n, lock
    self.mirrors = torch.zeros(ring, dtype=torch.int32).pin_memory() if mirrors is None else mirrors
    self.seq, self.base_seq, self.base_fill = 0, 0, 0
    self.after = collections.deque()                 # (seq, +n | -B) of the submissions behind base_seq
    self.up, self.low = 0, 0                         # the two bounds (plain ints: read without the lock)
    self.inflight, self.waits, self.open = 0, 0, True

  def _bounds(self):
    low = self.base_fill + sum(d for _, d in self.after if d < 0)
    self.low, self.up = low, low + sum(d for _, d in self.after if d > 0)

  @property
  def fill(self):
    """Completed unrolls the learner can rely on (a lower bound of the device's count; exact when nothing is in flight)."""
    return self.low

  def would_block(self, ahead=0):
    """True while admit(ahead) would wait (grpc_native's compute loop launches an already staged batch first)."""
    return self.open and self.up + self.n * (ahead + 1) > self.state.cap

  def admit(self, ahead=0, poll_s=0.0001):
    """Blocks while this batch, on top of everything submitted and the `ahead` batches staged before it, could overflow
    the device batch.  Once the gate is closed (shutdown) nothing is admitted any more: the batch's callers get
    CANCELLED, as the reference's pending calls do when its server shuts down -- letting batches through ungated then
    would overflow the device batch nobody consumes any longer."""
    while self.open and self.up + self.n * (ahead + 1) > self.state.cap:
      self.waits += 1
      time.sleep(poll_s)
    if not self.open:
      raise grpc_service.CancelledError('Server shutdown.')

  def submitted(self):
    """Under the submission lock, on the inference stream, right after the batch was enqueued."""
    self.seq += 1
    k = self.seq % self.mirrors.numel()
    self.mirrors[k:k + 1].copy_(self.state.batch_count, non_blocking=True)
    self.after.append((self.seq, self.n))
    self.inflight += 1
    self._bounds()
    return self.seq

  def completed(self, token):
    """After the batch's event completed: its mirror word has landed and is the exact count behind submission `token`."""
    with self.lock:
      self.inflight -= 1
      if token > self.base_seq:
        self.base_seq, self.base_fill = token, int(self.mirrors[token % self.mirrors.numel()])
        while self.after and self.after[0][0] <= token:
          self.after.popleft()
        self._bounds()

  def dequeued(self, batch_size):
    """Under the lock, right after a dequeue of batch_size columns was enqueued on the inference stream."""
    self.seq += 1
    self.after.append((self.seq, -batch_size))
    self._bounds()


class LearnerServer(object):

  def __init__(self, agent, learner, unroll_length, batch_size, inference_batch_size, num_envs, observation_shape,
               server_addresses, observation_dtype=torch.uint8, device='cuda', graphed=True, batch_capacity=None,
               transport='native', num_unroll_slots=2, num_io_threads=None, inference_slots=4, inference_pipeline=3,
               cu_split=None):
    self.agent, self.learner = agent, learner
    self.T, self.B, self.n = unroll_length, batch_size, inference_batch_size
    self.device = dev = torch.device(device)
    self.obs_shape, self.obs_dtype = tuple(observation_shape), observation_dtype
    A = agent._num_actions                            # pylint: disable=protected-access
    env_specs = utils.EnvOutput(Spec((), torch.float32), Spec((), torch.bool), Spec(self.obs_shape, observation_dtype),
                                Spec((), torch.bool), Spec((), torch.int32))
    ao_specs = networks.AgentOutput(Spec((), torch.int64), Spec((A,), torch.float32), Spec((), torch.float32))
    # room for every env completing an unroll at once (lock-stepped actors do) on top of a batch being assembled and
    # the inference batches in flight: back-pressure (BatchGate) then only engages when the learner really lags
    #  a full training batch plus every inference batch that can be admitted or staged ahead must fit, or
    # neither the learner (fill < B) nor inference (no room) could ever make progress
    need = batch_size + inference_batch_size * (inference_slots + 2)
    cap = batch_capacity or max(need, 2 * batch_size + num_envs + inference_batch_size * (inference_slots + 1))
    if not batch_capacity:
      cap = -(-cap // batch_size) * batch_size        # the ring head then only visits cap / batch_size positions
    if cap < need:
      raise ValueError('batch_capacity %d is too small: need >= batch_size + inference_batch_size * (inference_slots + 2) = %d'
                       % (cap, need))
    # Inference and training on DISJOINT compute units (cu_split = k: the first k of the 32 CUs of EVERY XCD serve
    # inference, the other 32 - k the train step; None / 0: one shared pool, inference on a high-priority stream).  On a
    # shared pool an inference kernel only gets the CU slots that the train step's workgroups vacate, and the first conv's
    # kernels keep theirs for the whole launch (persistent grids, ~130 us): in the closed loop every inference kernel ran
    # ~3x its stand-alone time.  Mask bit 8 j + x is CU j of XCD x (measured: tools/cu_mask_check.py -- a mask that leaves
    # an XCD empty is completed by the runtime, so "every other bit" keeps all 256 CUs while "bits 0..7 of every 16" keeps
    # 16 CUs per XCD): a per-XCD prefix is bits [0, 8 k).
    import os
    if cu_split is None and os.environ.get('SEEDRL_CU_SPLIT'):
      cu_split = int(os.environ['SEEDRL_CU_SPLIT'])
    self.cu_split = int(cu_split) if cu_split and 0 < int(cu_split) < 32 else None
    with torch.cuda.device(dev):
      if self.cu_split:
        from seed_rl_amd import ops
        k_ = self.cu_split
        self.infer_stream, self.infer_cus = ops.cu_mask_stream(dev, lambda i: i < 8 * k_)
        self.train_stream, self.train_cus = ops.cu_mask_stream(dev, lambda i: i >= 8 * k_)
      else:
        self.infer_stream = torch.cuda.Stream(device=dev, priority=-1)
        self.train_stream = torch.cuda.Stream(device=dev)
    self.infer_agent = agent.inference_twin() if hasattr(agent, 'inference_twin') else agent
    self.state = inference.FusedInferenceState(self.infer_agent, num_envs, unroll_length, env_specs, ao_specs,
                                               batch_capacity=cap, device=dev)
    self.lock = threading.Lock()                      # orders SUBMISSIONS to the inference stream (never held over a step)
    self.transport = transport
    self.gate = BatchGate(self.state, inference_batch_size, self.lock)
    np_obs = {torch.uint8: 'uint8', torch.int16: 'int16', torch.float32: 'float32'}[observation_dtype]
    if transport == 'native':
      from seed_rl_amd import grpc_native
      self.server = grpc_native.NativeServer(list(server_addresses), num_io_threads=num_io_threads)
      grpc_native.bind_inference(self.server, self.state, inference_batch_size, self.obs_shape,
                                 num_slots=inference_slots, observation_dtype=np_obs, stream=self.infer_stream,
                                 lock=self.lock, gate=self.gate, pipeline=inference_pipeline)
    else:
      self.server = grpc_service.Server(list(server_addresses))
      grpc_service.bind_inference(self.server, self.state, inference_batch_size, self.obs_shape, lock=self.lock,
                                  observation_dtype=np_obs, stream=self.infer_stream, gate=self.gate)
    # static training unrolls (inputs of the train step and of its HIP graph): two, so that the dequeue for step i+1
    # can be filled while step i still reads the other one
    T1 = unroll_length + 1
    z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)
    self.unrolls, self._graphs, self._done = [], [], []
    for s in range(max(1, num_unroll_slots)):
      init = agent.initial_state(batch_size)
      env = utils.EnvOutput(z((T1, batch_size), torch.float32), z((T1, batch_size), torch.bool),
                            self._obs_buffer(T1, batch_size, s), z((T1, batch_size), torch.bool),
                            z((T1, batch_size), torch.int32))
      ao = networks.AgentOutput(z((T1, batch_size), torch.int64), z((T1, batch_size, A), torch.float32),
                                z((T1, batch_size), torch.float32))
      self.unrolls.append(learner_lib.Unroll(init, z((T1, batch_size), torch.int64), env, ao))
      self._graphs.append(None)
      self._done.append(None)
    self._use_graph = bool(graphed) and getattr(learner.optimizer, 'capturable', False)
    self.unroll = self.unrolls[0]
    self.steps = 0
    self._ready = torch.cuda.Event()

  def _obs_buffer(self, T1, B, slot):
    # Atari agents read their frames from frames_buffer(T1, B)[3:]: dequeue straight into it (no copy in the step)
    shape, dtype = self.obs_shape, self.obs_dtype
    if hasattr(self.agent, 'frames_buffer') and len(shape) == 3 and shape[2] == 1 and dtype == torch.uint8:
      self.agent.frames_slot = slot
      buf = self.agent.frames_buffer(T1, B)[3:].view((T1, B) + tuple(shape))
      self.agent.frames_slot = 0
      return buf
    return torch.zeros((T1, B) + tuple(shape), dtype=dtype, device=self.device)

  def prepare(self):
    """Captures every slot's train-step graph BEFORE actors are served: GraphedStep's warm-up runs real
    optimizer steps on the shared flat parameter buffer and only then restores it, and capture synchronises the device
    -- neither belongs in the middle of serving, where the inference twin would act on perturbed weights."""
    if not self._use_graph:
      return
    with torch.cuda.device(self.device), torch.cuda.stream(self.train_stream):
      for slot in range(len(self.unrolls)):
        if self._graphs[slot] is None:
          if hasattr(self.agent, 'frames_slot'):
            self.agent.frames_slot = slot
          self._graphs[slot] = learner_lib.GraphedStep(self.learner, self.unrolls[slot])
    torch.cuda.synchronize(self.device)

  def start(self):
    self.prepare()
    self.server.start()

  def shutdown(self):
    self.gate.open = False                            # releases an inference batch waiting for room
    self.server.shutdown()

  def _step(self, slot):
    if hasattr(self.agent, 'frames_slot'):
      self.agent.frames_slot = slot
    if self._use_graph:
      if self._graphs[slot] is None:
        self._graphs[slot] = learner_lib.GraphedStep(self.learner, self.unrolls[slot])
      return self._graphs[slot]()
    return self.learner.minimize(self.unrolls[slot])

  def train_step(self, timeout=None, poll_s=0.0002):
    """learner.py:435-470: dequeue batch_size completed unrolls, minimize.  Returns (loss, session) or None on timeout.
    The call returns once the step is ENQUEUED on the train stream (the loss is a device scalar)."""
    t0 = time.time()
    slot = self.steps % len(self.unrolls)
    if self._done[slot] is not None:
      self._done[slot].synchronize()                  # the step that last read this unroll (two steps ago) is through
    while True:
      # wait on the HOST copy of the fill count (exact as of the last completed inference batch): no lock, no device
      # read while the batch is still filling -- the dequeue below is the only thing that orders against inference
      while self.gate.fill < self.B:
        if timeout is not None and time.time() - t0 > timeout:
          return None
        time.sleep(poll_s)
      with self.lock:
        ready = self.gate.fill >= self.B               # (re-read under the lock: another consumer may have been faster)
        if ready:
          with torch.cuda.device(self.device), torch.cuda.stream(self.infer_stream):
            self.state.dequeue_async(self.unrolls[slot], self.B)
            self.gate.dequeued(self.B)
            self._ready.record(self.infer_stream)
      if ready:
        break
    with torch.cuda.device(self.device), torch.cuda.stream(self.train_stream):
      self.train_stream.wait_event(self._ready)
      out = self._step(slot)
      ev = torch.cuda.Event()
      ev.record(self.train_stream)
      self._done[slot] = ev
    self.unroll = self.unrolls[slot]
    self.steps += 1
    if self.steps % 64 == 0:
      self.state.check_errors()                       # id / overflow flags of the inference bookkeeping (one host read)
    return out

  def synchronize(self):
    self.train_stream.synchronize()
    self.infer_stream.synchronize()


# --------------------------------------------------------------------------------------------------------------------- #
# The reference's entry point (agents/vtrace/learner.py:170-187): learner_loop(create_env_fn, create_agent_fn,
# create_optimizer_fn).  Flags become a config object; everything else is the signature and the order of operations of
# :189-483 restricted to the data path (no checkpoint manager / summary writer / tf.data).
# --------------------------------------------------------------------------------------------------------------------- #
class LoopConfig(object):
  """The absl flags learner_loop reads (learner.py:38-62, common_flags.py:25-55), flag defaults."""

  def __init__(self, server_address='unix:/tmp/agent_grpc', batch_size=32, inference_batch_size=-1, unroll_length=100,
               total_environment_frames=int(1e9), num_envs=4, num_action_repeats=1, device='cuda', transport='native',
               loss=None, max_steps=None, step_timeout=30.0):
    self.server_address, self.batch_size, self.unroll_length = server_address, batch_size, unroll_length
    self.inference_batch_size = inference_batch_size
    self.total_environment_frames, self.num_envs = total_environment_frames, num_envs
    self.num_action_repeats, self.device, self.transport = num_action_repeats, device, transport
    self.loss, self.max_steps, self.step_timeout = loss, max_steps, step_timeout


LoopResult = collections.namedtuple('LoopResult', 'iterations num_env_frames last_loss last_session server')


def learner_loop(create_env_fn, create_agent_fn, create_optimizer_fn, config=None, on_step=None):
  """Main learner loop (agents/vtrace/learner.py:170-483), data path only.

  Args (as in the reference):
    create_env_fn: `create_env_fn(task)` -> an environment with `.observation_space` (shape, dtype) and
      `.action_space` (`.n` discrete actions); used once to read the specs (learner.py:189-196) and closed.
    create_agent_fn: `create_agent_fn(action_space, observation_space, parametric_action_distribution)` -> agent.
    create_optimizer_fn: `create_optimizer_fn(final_iteration)` -> (optimizer, learning_rate_fn) (learner.py:217-222).
  config: LoopConfig (the reference's flags).  on_step(iterations, loss, session): optional callback per train step.
  Serves actors on config.server_address and trains until total_environment_frames (or config.max_steps) is reached.
  Returns LoopResult; the server is shut down."""
  import numpy as np
  from seed_rl_amd import parametric_distribution as pd
  cfg = config or LoopConfig()
  env = create_env_fn(0)
  obs_space, action_space = env.observation_space, env.action_space
  if hasattr(env, 'close'):
    env.close()                                                                        # learner.py:196
  dist = pd.get_parametric_distribution_for_action_space(action_space) \
      if hasattr(pd, 'get_parametric_distribution_for_action_space') else pd.categorical_distribution(int(action_space.n))
  agent = create_agent_fn(action_space, obs_space, dist)
  iter_frame_ratio = cfg.batch_size * cfg.unroll_length * cfg.num_action_repeats        # learner.py:236-237
  final_iteration = int(np.ceil(cfg.total_environment_frames / iter_frame_ratio))       # :238-239
  optimizer, _ = create_optimizer_fn(final_iteration)
  lrn = learner_lib.Learner(agent, optimizer, dist, config=cfg.loss)
  n = cfg.inference_batch_size
  if n == -1:
    n = max(1, cfg.num_envs // 2)                                                       # learner.py:306-308
  obs_dtype = {np.dtype(np.uint8): torch.uint8, np.dtype(np.float32): torch.float32,
               np.dtype(np.int16): torch.int16, np.dtype(np.uint16): torch.int16}[np.dtype(obs_space.dtype)]
  server = LearnerServer(agent, lrn, cfg.unroll_length, cfg.batch_size, n, cfg.num_envs, tuple(obs_space.shape),
                         [cfg.server_address], observation_dtype=obs_dtype, device=cfg.device,
                         transport=cfg.transport, graphed=getattr(optimizer, 'capturable', False))
  server.start()
  iterations, loss, session = 0, None, None
  try:
    while iterations < final_iteration and (cfg.max_steps is None or iterations < cfg.max_steps):   # :435
      out = server.train_step(timeout=cfg.step_timeout)
      if out is None:
        break                                                                           # actors went away
      loss, session = out
      iterations += 1
      if on_step is not None:
        on_step(iterations, loss, session)
    server.synchronize()
  finally:
    server.shutdown()
  return LoopResult(iterations, iterations * iter_frame_ratio, loss, session, server)
