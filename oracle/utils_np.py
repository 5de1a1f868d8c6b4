"""NumPy restatement of the learner-side state containers (oracle; test infrastructure only).

Follows /root/reference/common/utils.py: UnrollStore (:119-257), Aggregator (:461-543), batch_apply
(:714-732), make_time_major (:735-761).  Pinned by the reference's known-answer sequences
(tests/utils_test.py:70-301, 585-606) restated in tests/test_oracle_utils.py.
Structures are flat dicts / tuples of arrays (tf.nest replaced by explicit maps).
"""
import collections

import numpy as np

Spec = collections.namedtuple('Spec', 'shape dtype')     # stands in for tf.TensorSpec (a leaf)


def _map(fn, struct):
  if isinstance(struct, Spec):
    return fn(struct)
  if isinstance(struct, dict):
    return {k: _map(fn, v) for k, v in struct.items()}
  if isinstance(struct, tuple) and hasattr(struct, '_fields'):
    return type(struct)(*[_map(fn, v) for v in struct])
  if isinstance(struct, (tuple, list)):
    return type(struct)(_map(fn, v) for v in struct)
  return fn(struct)


def _zip_apply(fn, a, b):
  if isinstance(a, dict):
    return {k: _zip_apply(fn, a[k], b[k]) for k in a}
  if isinstance(a, tuple) and hasattr(a, '_fields'):
    return type(a)(*[_zip_apply(fn, x, y) for x, y in zip(a, b)])
  if isinstance(a, (tuple, list)):
    return type(a)(_zip_apply(fn, x, y) for x, y in zip(a, b))
  return fn(a, b)


class UnrollStore(object):
  """utils.py:119-257.  timestep_specs: structure of Spec(shape, dtype)."""

  def __init__(self, num_envs, unroll_length, timestep_specs, num_overlapping_steps=0):
    self._full_length = num_overlapping_steps + unroll_length + 1            # :130
    self._unroll_length, self._overlap = unroll_length, num_overlapping_steps
    self._state = _map(lambda sd: np.zeros((num_envs, self._full_length) + tuple(sd.shape), sd.dtype),
                       timestep_specs)
    self._index = np.full([num_envs], num_overlapping_steps, np.int32)        # :143-146

  def append(self, env_ids, values):
    env_ids = np.asarray(env_ids)
    if len(np.unique(env_ids)) != len(env_ids):                               # :173-176
      raise ValueError('Duplicate environment ids in store')
    cur = self._index[env_ids]
    def upd(s, v):
      s[env_ids, cur] = v                                                     # :187-190
      return s
    _zip_apply(upd, self._state, values)
    self._index[env_ids] += 1                                                 # :194
    return self._complete_unrolls(env_ids)

  def reset(self, env_ids):
    env_ids = np.asarray(env_ids, np.int64)
    self._index[env_ids] = self._overlap                                      # :207-208
    j = self._overlap
    def z(s):
      s[env_ids, :j] = 0                                                      # :210-225
      return s
    _map(z, self._state)

  def _complete_unrolls(self, env_ids):
    idx = self._index[env_ids]
    done_ids = env_ids[idx == self._full_length].astype(np.int64)             # :229-233
    unrolls = _map(lambda s: s[done_ids].copy(), self._state)
    j = self._overlap + 1
    def carry(s):
      s[done_ids, :j] = s[done_ids, -j:]                                      # :237-252
      return s
    _map(carry, self._state)
    self._index[done_ids] = 1 + self._overlap                                 # :254-255
    return done_ids, unrolls


class Aggregator(object):
  """utils.py:461-543: per-env table with reset / add / read / replace."""

  def __init__(self, num_envs, spec):
    self._state = np.zeros((num_envs,) + tuple(spec.shape), spec.dtype)

  def reset(self, env_ids):
    self._state[np.asarray(env_ids, np.int64)] = 0

  def add(self, env_ids, values):
    np.add.at(self._state, np.asarray(env_ids, np.int64), values)

  def read(self, env_ids):
    return self._state[np.asarray(env_ids, np.int64)].copy()

  def replace(self, env_ids, values):
    env_ids = np.asarray(env_ids, np.int64)
    if len(np.unique(env_ids)) != len(env_ids):                               # :530-540
      raise ValueError('Duplicate environment ids')
    self._state[env_ids] = values


def batch_apply(fn, inputs):
  """utils.py:714-732."""
  t, b = inputs[0].shape[0], inputs[0].shape[1]
  folded = [x.reshape((t * b,) + x.shape[2:]) for x in inputs]
  outs = fn(*folded)
  return tuple(o.reshape((t, b) + o.shape[1:]) for o in outs)


def make_time_major(x):
  """utils.py:735-761: swap the two leading axes of every array of the structure."""
  return _map(lambda a: np.swapaxes(a, 0, 1) if a.ndim >= 2 else a, x)


class PrioritizedReplay(object):
  """utils.py:260-370 (restated; sampling takes explicit uniforms so that it is reproducible: the reference draws
  with tf.random.categorical).  Pinned by tests/utils_test.py:304-405."""

  def __init__(self, size, specs, importance_sampling_exponent):
    self._priorities = np.zeros([size], np.float32)
    self._buffer = _map(lambda sd: np.zeros((size,) + tuple(sd.shape), sd.dtype), specs)
    self.num_inserted = 0
    self._is_exp = importance_sampling_exponent

  def insert(self, values, priorities):
    n = _first_leaf(values).shape[0]
    size = self._priorities.shape[0]
    idx = np.arange(self.num_inserted, self.num_inserted + n) % size                      # :297
    def upd(b, v):
      b[idx] = v
      return b
    _zip_apply(upd, self._buffer, values)
    self.num_inserted += n
    self._priorities[idx] = priorities
    return idx

  def sample(self, num_samples, priority_exp, uniforms=None, rng=None):
    assert self.num_inserted > 0, 'Cannot sample if replay buffer is empty'
    size = self._priorities.shape[0]
    limit = min(size, self.num_inserted)
    rng = rng or np.random.default_rng(0)
    if priority_exp == 0:
      indices = rng.integers(0, limit, num_samples)
      weights = np.ones(num_samples, np.float32)
    else:
      prob = self._priorities[:limit].astype(np.float32) ** np.float32(priority_exp)
      prob = prob / prob.sum(dtype=np.float32)                                           # :342-343
      u = rng.uniform(size=num_samples).astype(np.float32) if uniforms is None else np.asarray(uniforms, np.float32)
      cdf = np.cumsum(prob, dtype=np.float64)
      indices = np.minimum(np.searchsorted(cdf, u.astype(np.float64) * cdf[-1], side='right'), limit - 1)
      weights = ((np.float32(1.) / np.float32(limit)) / prob[indices]) ** np.float32(self._is_exp)   # :350-352
      weights = (weights / weights.max()).astype(np.float32)                              # :353
    return indices.astype(np.int64), weights, _map(lambda b: b[indices].copy(), self._buffer)

  def update_priorities(self, indices, priorities):
    self._priorities[np.asarray(indices, np.int64)] = priorities


def _first_leaf(struct):
  while not isinstance(struct, np.ndarray):
    struct = list(struct.values())[0] if isinstance(struct, dict) else struct[0]
  return struct
