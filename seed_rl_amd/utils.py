"""Learner-side data contract and small helpers.

Mirrors the pieces of /root/reference/common/utils.py that sit on the hot path:
EnvOutput (:41-42), make_time_major (:735-761), batch_apply (:714-732).  State
containers (UnrollStore, Aggregator) live in seed_rl_amd/unroll_store.py.
"""
import collections

import torch

EnvOutput = collections.namedtuple(
    'EnvOutput', 'reward done observation abandoned episode_step')


def map_structure(fn, *structs):
  """Minimal tf.nest.map_structure over namedtuples / tuples / lists / tensors."""
  s0 = structs[0]
  if isinstance(s0, tuple) and hasattr(s0, '_fields'):
    return type(s0)(*[map_structure(fn, *xs) for xs in zip(*structs)])
  if isinstance(s0, (tuple, list)):
    return type(s0)(map_structure(fn, *xs) for xs in zip(*structs))
  if isinstance(s0, dict):
    return {k: map_structure(fn, *[s[k] for s in structs]) for k in s0}
  return fn(*structs)


def flatten(struct):
  out = []
  map_structure(lambda t: out.append(t), struct)
  return out


def batch_apply(fn, inputs):
  """utils.py:714-732: folds time into batch, applies fn, unfolds."""
  flat = flatten(inputs)
  t, b = flat[0].shape[0], flat[0].shape[1]
  folded = map_structure(lambda x: x.reshape((t * b,) + tuple(x.shape[2:])), inputs)
  out = fn(*folded)
  return map_structure(lambda x: x.reshape((t, b) + tuple(x.shape[1:])), out)


def make_time_major(x):
  """utils.py:735-761: [B, T, ...] -> [T, B, ...] for every tensor of a structure."""
  def tr(t):
    if t.dim() < 2:
      return t
    return t.transpose(0, 1).contiguous()
  return map_structure(tr, x)
