"""Device-resident prioritized replay: mirror of PrioritizedReplay in /root/reference/common/utils.py:260-370.

The buffer lives in HBM (Atari R2D2: 1e5 unrolls x 0.85 MB = 85 GB of the MI355X's 288 GB; the reference keeps it
in host-side TF variables), FIFO insertion and sampled gathers are one multi-field row move each (csrc/store.hip /
inference.hip), sampling is csrc/replay.hip.  `num_inserted` is kept on the host (insert sizes are host constants),
so no call reads the device.
"""
import torch

from seed_rl_amd import ops, unroll_store, utils


class PrioritizedReplay(object):

  def __init__(self, size, specs, importance_sampling_exponent, device='cuda', name='PrioritizedReplay'):
    self._name, self._size = name, size
    self.device = torch.device(device)
    self._priorities = torch.zeros(size, dtype=torch.float32, device=self.device)
    self._buffer = unroll_store._map_specs(
        lambda s: torch.zeros((size,) + tuple(s.shape), dtype=s.dtype, device=self.device), specs)
    self.num_inserted = 0
    self._importance_sampling_exponent = importance_sampling_exponent
    self._ws = torch.empty(ops.replay_sample_workspace_bytes(size) // 4 + 4, dtype=torch.float32, device=self.device)

  def insert(self, values, priorities):
    """FIFO insertion with wrap-around (utils.py:277-307).  Returns the int64 slot indices."""
    flat_v = utils.flatten(values)
    n = flat_v[0].shape[0]
    if n > self._size:                 # duplicate slots would race in the row mover (the reference's scatter is undefined there too)
      raise ValueError('insert of %d values into a replay buffer of size %d' % (n, self._size))
    idx = (torch.arange(self.num_inserted, self.num_inserted + n, dtype=torch.int64, device=self.device)
           % self._size).contiguous()
    bufs = utils.flatten(self._buffer)
    ops.rows_move_multi(bufs, [v.to(b.dtype).contiguous() for v, b in zip(flat_v, bufs)],
                        [unroll_store._row_bytes(b, 1) for b in bufs], idx, None, n)
    self.num_inserted += n
    self._priorities[idx] = torch.as_tensor(priorities, device=self.device).to(torch.float32)
    return idx

  def sample(self, num_samples, priority_exp, uniforms=None):
    """utils.py:309-357.  Returns (indices int64[num_samples], weights f32[num_samples], sampled values).
    `uniforms` (f32[num_samples] in [0,1)) may be supplied for reproducibility; default torch.rand on the device."""
    if self.num_inserted <= 0:
      raise ValueError('Cannot sample if replay buffer is empty')
    limit = min(self._size, self.num_inserted)
    if priority_exp == 0:                                                       # uniform (:338-340)
      indices = torch.randint(0, limit, (num_samples,), dtype=torch.int64, device=self.device)
      weights = torch.ones(num_samples, dtype=torch.float32, device=self.device)
    else:
      if uniforms is None:
        uniforms = torch.rand(num_samples, dtype=torch.float32, device=self.device)
      indices = torch.empty(num_samples, dtype=torch.int64, device=self.device)
      weights = torch.empty(num_samples, dtype=torch.float32, device=self.device)
      ops.replay_sample(self._priorities, limit, priority_exp, self._importance_sampling_exponent,
                        uniforms.contiguous(), indices, weights, self._ws)
    def gather(b):
      return torch.empty((num_samples,) + tuple(b.shape[1:]), dtype=b.dtype, device=self.device)
    out = utils.map_structure(gather, self._buffer)
    bufs = utils.flatten(self._buffer)
    ops.rows_move_multi(utils.flatten(out), bufs, [unroll_store._row_bytes(b, 1) for b in bufs], None, indices,
                        num_samples)
    return indices, weights, out

  def update_priorities(self, indices, priorities):
    """utils.py:359-370 (duplicate indices: which priority wins is unspecified, as in the reference)."""
    idx = torch.as_tensor(indices, device=self.device).to(torch.int64)
    self._priorities[idx] = torch.as_tensor(priorities, device=self.device).to(torch.float32)


class UnrollReplay(PrioritizedReplay):
  """PrioritizedReplay of R2D2 `Unroll`s (agents/r2d2/learner.py:95-96, 659-668) whose per-timestep fields enter and
  leave TIME-MAJOR: the learner-side unroll store emits completed unrolls as [T1, n, ...] and the train step consumes
  [T1, B, ...], so the reference's batch-major rows + `utils.make_time_major` transposes (learner.py:453-457) become
  part of the one row move per field that inserts / gathers anyway (row = one STEP of one unroll; indices from
  csrc/replay.hip: replay_time_rows).  Buffer layout and slot arithmetic are the reference's ([size, T1, ...] rows,
  FIFO wrap-around), so `insert` / `sample` of the base class see the same buffer batch-major.

  specs: an `Unroll`-shaped structure (any namedtuple with fields agent_state, priority, prev_actions, env_outputs,
  agent_outputs) of per-ROW Specs: the time fields carry the leading T1."""

  TIME_FIELDS = ('prev_actions', 'env_outputs', 'agent_outputs')

  def __init__(self, size, specs, importance_sampling_exponent, device='cuda', name='UnrollReplay'):
    super(UnrollReplay, self).__init__(size, specs, importance_sampling_exponent, device, name)
    self._time = [b for f in self.TIME_FIELDS for b in utils.flatten(getattr(self._buffer, f)) if b is not None]
    self._static = [b for f in self._buffer._fields if f not in self.TIME_FIELDS
                    for b in utils.flatten(getattr(self._buffer, f)) if b is not None]
    steps = set(int(b.shape[1]) for b in self._time)
    if len(steps) != 1:
      raise ValueError('every per-timestep field must have the same leading T1, got %s' % sorted(steps))
    self.steps = steps.pop()

  def _leaves(self, struct):
    t = [v for f in self.TIME_FIELDS for v in utils.flatten(getattr(struct, f)) if v is not None]
    s = [v for f in struct._fields if f not in self.TIME_FIELDS for v in utils.flatten(getattr(struct, f))
         if v is not None]
    return t, s

  def _rows(self, slots):
    n = slots.numel()
    rr = torch.empty(n * self.steps, dtype=torch.int64, device=self.device)
    br = torch.empty(n * self.steps, dtype=torch.int64, device=self.device)
    ops.replay_time_rows(slots, self.steps, rr, br)
    return rr

  def insert_time_major(self, unroll, priorities):
    """replay_buffer.insert(unrolls, unrolls.priority) (learner.py:436) for unrolls whose time fields are
    [T1, n, ...]; `unroll.priority` itself is stored too when the specs hold it.  Returns the slot indices."""
    tv, sv = self._leaves(unroll)
    n = int(torch.as_tensor(priorities).shape[0])
    if n > self._size:
      raise ValueError('insert of %d unrolls into a replay buffer of size %d' % (n, self._size))
    slots = (torch.arange(self.num_inserted, self.num_inserted + n, dtype=torch.int64, device=self.device)
             % self._size).contiguous()
    rr = self._rows(slots)
    T1 = self.steps
    ops.rows_move_multi(self._time, [v.to(b.dtype).contiguous() for v, b in zip(tv, self._time)],
                        [unroll_store._row_bytes(b, 2) for b in self._time], rr, None, n * T1)
    if self._static:
      ops.rows_move_multi(self._static, [v.to(b.dtype).contiguous() for v, b in zip(sv, self._static)],
                          [unroll_store._row_bytes(b, 1) for b in self._static], slots, None, n)
    self.num_inserted += n
    self._priorities[slots] = torch.as_tensor(priorities, device=self.device).to(torch.float32)
    return slots

  def sample_time_major(self, num_samples, priority_exp, uniforms=None, out=None):
    """replay_buffer.sample + make_time_major (learner.py:451-457): (indices, weights, unroll with [T1, B, ...] time
    fields).  `out`: an Unroll of preallocated tensors to gather into (the static input of a captured train step)."""
    if self.num_inserted <= 0:
      raise ValueError('Cannot sample if replay buffer is empty')
    limit = min(self._size, self.num_inserted)
    if priority_exp == 0:
      indices = torch.randint(0, limit, (num_samples,), dtype=torch.int64, device=self.device)
      weights = torch.ones(num_samples, dtype=torch.float32, device=self.device)
    else:
      if uniforms is None:
        uniforms = torch.rand(num_samples, dtype=torch.float32, device=self.device)
      indices = torch.empty(num_samples, dtype=torch.int64, device=self.device)
      weights = torch.empty(num_samples, dtype=torch.float32, device=self.device)
      ops.replay_sample(self._priorities, limit, priority_exp, self._importance_sampling_exponent,
                        uniforms.contiguous(), indices, weights, self._ws)
    T1, B = self.steps, num_samples
    if out is None:
      mk = lambda b, lead: None if b is None else torch.empty(lead + tuple(b.shape[len(lead):]), dtype=b.dtype,
                                                              device=self.device)
      out = type(self._buffer)(*[
          utils.map_structure((lambda b: mk(b, (T1, B))) if f in self.TIME_FIELDS else (lambda b: mk(b, (B,))),
                              getattr(self._buffer, f)) for f in self._buffer._fields])
    to, so = self._leaves(out)
    rr = self._rows(indices)
    ops.rows_move_multi(to, self._time, [unroll_store._row_bytes(b, 2) for b in self._time], None, rr, B * T1)
    if self._static:
      ops.rows_move_multi(so, self._static, [unroll_store._row_bytes(b, 1) for b in self._static], None, indices, B)
    return indices, weights, out
