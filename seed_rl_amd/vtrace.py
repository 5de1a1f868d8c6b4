"""V-trace on MI355X.  Drop-in for /root/reference/common/vtrace.py.

`from_importance_weights` keeps the reference's Python signature, defaults,
rank checks (vtrace.py:99-107), `None`-threshold semantics (:91-96,:111-114,
:138-142) and return type (:31), and runs the whole op sequence of
vtrace.py:84-148 as ONE HIP kernel (csrc/vtrace.hip) through the C ABI
`seedhip_vtrace_from_importance_weights`.  Outputs carry no gradient
(vtrace.py:147-148).
"""
import collections

import torch

from seed_rl_amd import _lib

VTraceReturns = collections.namedtuple('VTraceReturns', 'vs pg_advantages')


def _f32(x, device):
  t = torch.as_tensor(x, device=device)
  return t.to(torch.float32).contiguous()       # convert_to_tensor(dtype=float32), :86-90


def from_importance_weights(
    target_action_log_probs, behaviour_action_log_probs,
    discounts, rewards, values, bootstrap_value,
    clip_rho_threshold=1.0, clip_pg_rho_threshold=1.0, lambda_=1.0,
    name='vtrace_from_importance_weights'):
  """See common/vtrace.py:34-82 for the argument documentation."""
  del name
  dev = None
  for x in (target_action_log_probs, behaviour_action_log_probs, discounts, rewards, values,
            bootstrap_value):
    if isinstance(x, torch.Tensor) and x.is_cuda:
      dev = x.device
      break
  if dev is None:
    raise _lib.SeedHipError('from_importance_weights needs device (cuda) tensors; '
                            'there is no CPU fallback.')
  with torch.no_grad():
    tgt = _f32(target_action_log_probs, dev)
    beh = _f32(behaviour_action_log_probs, dev)
    discounts = _f32(discounts, dev)
    rewards = _f32(rewards, dev)
    values = _f32(values, dev)
    bootstrap_value = _f32(bootstrap_value, dev)

    # Make sure tensor ranks are consistent (vtrace.py:99-107).
    rho_rank = tgt.dim()
    if beh.shape != tgt.shape:
      raise ValueError('log-prob shapes differ: %s vs %s' % (tuple(tgt.shape), tuple(beh.shape)))
    for nm, t, r in (('values', values, rho_rank), ('bootstrap_value', bootstrap_value, rho_rank - 1),
                     ('discounts', discounts, rho_rank), ('rewards', rewards, rho_rank)):
      if t.dim() != r:
        raise ValueError('Shape %s of %s must have rank %d' % (tuple(t.shape), nm, r))
    for nm, c in (('clip_rho_threshold', clip_rho_threshold),
                  ('clip_pg_rho_threshold', clip_pg_rho_threshold)):
      if c is not None and torch.as_tensor(c).dim() != 0:
        raise ValueError('%s must be a scalar' % nm)
    if rho_rank < 1:
      raise ValueError('log-probs must have rank >= 1 ([T, B, ...])')
    for t in (discounts, rewards, values):
      if t.shape != tgt.shape:
        raise ValueError('shape mismatch: %s vs %s' % (tuple(t.shape), tuple(tgt.shape)))
    if tuple(bootstrap_value.shape) != tuple(tgt.shape[1:]):
      raise ValueError('bootstrap_value shape %s != %s' % (tuple(bootstrap_value.shape), tuple(tgt.shape[1:])))

    T = tgt.shape[0]
    B = 1
    for d in tgt.shape[1:]:
      B *= d                                   # trailing dims are independent columns (:49-51)
    vs = torch.empty_like(values)
    pg = torch.empty_like(values)
    crho = -1.0 if clip_rho_threshold is None else float(clip_rho_threshold)
    cpg = -1.0 if clip_pg_rho_threshold is None else float(clip_pg_rho_threshold)
    if (clip_rho_threshold is not None and crho < 0) or (clip_pg_rho_threshold is not None and cpg < 0):
      raise ValueError('clip thresholds must be >= 0 (or None)')
    with torch.cuda.device(dev):
      rc = _lib.lib().seedhip_vtrace_from_importance_weights(
          _lib.ptr(tgt), _lib.ptr(beh), _lib.ptr(discounts), _lib.ptr(rewards), _lib.ptr(values),
          _lib.ptr(bootstrap_value), crho, cpg, float(lambda_), int(T), int(B),
          _lib.ptr(vs), _lib.ptr(pg), _lib.stream())
    _lib.check(rc, 'seedhip_vtrace_from_importance_weights')
  return VTraceReturns(vs=vs, pg_advantages=pg)
