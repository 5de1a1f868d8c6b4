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
