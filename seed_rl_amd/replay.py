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
