"""Device-resident trajectory store: UnrollStore / Aggregator of /root/reference/common/utils.py
(:119-257, :461-543) re-designed for HBM residency.

Differences from the reference (by design):
  * every field is stored TIME-MAJOR, [full_length, num_envs, ...], and completed unrolls are returned
    time-major [full_length, num_completed, ...] (or written straight into a pre-allocated training batch
    with `out=` / `out_col=`): the reference's batch-major gather + host-side `make_time_major` transpose
    (utils.py:735-761, learner.py:418-432) never happens;
  * all payload movement is one HIP primitive (csrc/store.hip: rows_move); only the tiny int64 index
    arithmetic uses torch ops.
Semantics (indices, overlap carry-over, reset, duplicate-id check) follow the reference line by line and
are pinned by its known-answer sequences (tests/utils_test.py:70-286 -> tests/test_gpu_store.py).
"""
import collections

import numpy as np
import torch

from seed_rl_amd import ops, utils

Spec = collections.namedtuple('Spec', 'shape dtype')      # stands in for tf.TensorSpec


def _map_specs(fn, specs):
  if isinstance(specs, Spec):
    return fn(specs)
  if isinstance(specs, tuple) and hasattr(specs, '_fields'):
    return type(specs)(*[_map_specs(fn, s) for s in specs])
  if isinstance(specs, (tuple, list)):
    return type(specs)(_map_specs(fn, s) for s in specs)
  if isinstance(specs, dict):
    return {k: _map_specs(fn, v) for k, v in specs.items()}
  raise TypeError('specs must be a structure of Spec, got %r' % (specs,))


def specs_like(struct):
  """Spec structure of a structure of [batch, ...] tensors."""
  return utils.map_structure(lambda t: Spec(tuple(t.shape[1:]), t.dtype), struct)


def _row_bytes(t, lead):
  return int(np.prod(t.shape[lead:], dtype=np.int64)) * t.element_size()


def _check_unique(ids, what):
  if ids.numel() and torch.unique(ids).numel() != ids.numel():
    raise ValueError('Duplicate environment ids in %s' % what)       # utils.py:173-176, 530-540


class UnrollStore(object):

  def __init__(self, num_envs, unroll_length, timestep_specs, num_overlapping_steps=0, device='cuda',
               name='UnrollStore'):
    self._name = name
    self._num_envs = num_envs
    self._full_length = num_overlapping_steps + unroll_length + 1              # utils.py:130
    self._unroll_length, self._overlap = unroll_length, num_overlapping_steps
    self.device = torch.device(device)
    self._specs = timestep_specs
    self._state = _map_specs(
        lambda s: torch.zeros((self._full_length, num_envs) + tuple(s.shape), dtype=s.dtype, device=self.device),
        timestep_specs)
    self._index = torch.full((num_envs,), num_overlapping_steps, dtype=torch.int64, device=self.device)
    self._t_range = torch.arange(self._full_length, dtype=torch.int64, device=self.device)

  @property
  def full_length(self):
    return self._full_length

  @property
  def unroll_specs(self):
    """Specs of one completed unroll, time-major: [full_length, ...]."""
    return _map_specs(lambda s: Spec((self._full_length,) + tuple(s.shape), s.dtype), self._specs)

  def _rows(self, t_idx, ids, width):
    """Flattened row ids (t, column) of a [len(t_idx), len(ids)] block in a [*, width, ...] buffer."""
    return (t_idx[:, None] * width + ids[None, :]).reshape(-1).contiguous()

  def append(self, env_ids, values, out=None, out_col=0, check_duplicates=True):
    """Appends one step for `env_ids` (utils.py:155-196).  Returns (completed_env_ids int64[k], unrolls)
    with unrolls time-major [full_length, k, ...]; with `out` (a structure of [full_length, capacity, ...]
    tensors) the completed unrolls are instead written into columns out_col .. out_col+k-1 of `out`."""
    ids = env_ids.to(device=self.device, dtype=torch.int64).contiguous()
    if check_duplicates:
      _check_unique(ids, 'store ' + self._name)
    n = ids.numel()
    flat_state, flat_vals = utils.flatten(self._state), utils.flatten(values)
    for v in flat_vals:
      if v.shape[0] != n:
        raise ValueError('Batch dimension must equal the number of environments in store %s.' % self._name)
    rows = (self._index[ids] * self._num_envs + ids).contiguous()
    for s, v in zip(flat_state, flat_vals):
      ops.rows_move(s, rows, v.to(s.dtype).contiguous(), None, n, _row_bytes(s, 2))      # :187-190
    self._index.index_add_(0, ids, torch.ones_like(ids))                                  # :194
    return self._complete_unrolls(ids, out, out_col)

  def reset(self, env_ids):
    """utils.py:198-225 (actor restarts only, not episode boundaries)."""
    ids = env_ids.to(device=self.device, dtype=torch.int64).contiguous()
    if ids.numel() == 0:
      return
    self._index[ids] = self._overlap
    if self._overlap:
      rows = self._rows(self._t_range[:self._overlap], ids, self._num_envs)
      for s in utils.flatten(self._state):
        ops.rows_move(s, rows, None, None, rows.numel(), _row_bytes(s, 2))

  def _complete_unrolls(self, ids, out, out_col):
    done = ids[self._index[ids] == self._full_length]                                     # :229-233 (one host sync)
    k = done.numel()
    L, E = self._full_length, self._num_envs
    src_rows = self._rows(self._t_range, done, E)
    if out is None:
      def gather(s):
        o = torch.empty((L, k) + tuple(s.shape[2:]), dtype=s.dtype, device=self.device)
        ops.rows_move(o, None, s, src_rows, L * k, _row_bytes(s, 2))
        return o
      unrolls = utils.map_structure(gather, self._state)
    else:
      cap = utils.flatten(out)[0].shape[1]
      if out_col + k > cap:
        raise ValueError('training batch has %d columns; cannot place %d unrolls at column %d' % (cap, k, out_col))
      cols = torch.arange(out_col, out_col + k, dtype=torch.int64, device=self.device)
      dst_rows = self._rows(self._t_range, cols, cap)
      for s, o in zip(utils.flatten(self._state), utils.flatten(out)):
        ops.rows_move(o, dst_rows, s, src_rows, L * k, _row_bytes(s, 2))
      unrolls = out
    if k:
      # s[:j, ids] = s[-j:, ids] (utils.py:237-252); gather first, then scatter: the ranges may overlap
      j = self._overlap + 1
      tail = self._rows(self._t_range[L - j:], done, E)
      head = self._rows(self._t_range[:j], done, E)
      for s in utils.flatten(self._state):
        tmp = torch.empty((j * k,) + tuple(s.shape[2:]), dtype=s.dtype, device=self.device)
        ops.rows_move(tmp, None, s, tail, j * k, _row_bytes(s, 2))
        ops.rows_move(s, head, tmp, None, j * k, _row_bytes(s, 2))
      self._index[done] = 1 + self._overlap                                               # :254-255
    return done, unrolls


class Aggregator(object):
  """utils.py:461-543: [num_envs, ...] tables with reset / add / read / replace."""

  def __init__(self, num_envs, specs, device='cuda', name='Aggregator'):
    self._name = name
    self.device = torch.device(device)
    self._state = _map_specs(
        lambda s: torch.zeros((num_envs,) + tuple(s.shape), dtype=s.dtype, device=self.device), specs)

  def _ids(self, env_ids):
    return torch.as_tensor(env_ids, device=self.device).to(torch.int64).contiguous()

  def reset(self, env_ids):
    ids = self._ids(env_ids)
    for s in utils.flatten(self._state):
      ops.rows_move(s, ids, None, None, ids.numel(), _row_bytes(s, 1))

  def add(self, env_ids, values):
    """Scatter-add (utils.py:497-511).  A few scalars per env: torch index_add_, not a HIP kernel."""
    ids = self._ids(env_ids)
    for s, v in zip(utils.flatten(self._state), utils.flatten(values)):
      v = torch.as_tensor(v, device=self.device).to(s.dtype)
      s.index_add_(0, ids, v.expand((ids.numel(),) + tuple(s.shape[1:])).contiguous())

  def read(self, env_ids):
    ids = self._ids(env_ids)
    def rd(s):
      o = torch.empty((ids.numel(),) + tuple(s.shape[1:]), dtype=s.dtype, device=self.device)
      ops.rows_move(o, None, s, ids, ids.numel(), _row_bytes(s, 1))
      return o
    return utils.map_structure(rd, self._state)

  def replace(self, env_ids, values, check_duplicates=True):
    ids = self._ids(env_ids)
    if check_duplicates:
      _check_unique(ids, 'aggregator ' + self._name)
    for s, v in zip(utils.flatten(self._state), utils.flatten(values)):
      ops.rows_move(s, ids, torch.as_tensor(v, device=self.device).to(s.dtype).contiguous(), None, ids.numel(),
                    _row_bytes(s, 1))
