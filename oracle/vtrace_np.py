"""NumPy restatement of V-trace (oracle; test infrastructure only).

Follows /root/reference/common/vtrace.py:34-148 op for op, in IEEE fp32, with
the same evaluation order as the reference's Python-unrolled reverse scan
(`acc = delta + discount * c * acc`, vtrace.py:124-130).

`ground_truth_calculation` restates the O(T^2) NumPy formula the reference's
own test uses (tests/vtrace_test.py:41-82); it is the second, structurally
different implementation used to pin the fp32 recursion.
"""
import collections

import numpy as np

VTraceReturns = collections.namedtuple('VTraceReturns', 'vs pg_advantages')

f32 = np.float32


def from_importance_weights(target_action_log_probs, behaviour_action_log_probs,
                            discounts, rewards, values, bootstrap_value,
                            clip_rho_threshold=1.0, clip_pg_rho_threshold=1.0,
                            lambda_=1.0, dtype=np.float32):
  """vtrace.py:34-148.  All inputs [T,B,...] / [B,...]; returns fp32 arrays.

  dtype=np.float64 gives a high-precision variant of the *same* recursion
  (used only to report fp32 rounding distance).
  """
  tgt = np.asarray(target_action_log_probs, dtype=dtype)
  beh = np.asarray(behaviour_action_log_probs, dtype=dtype)
  log_rhos = (tgt - beh).astype(dtype)                       # vtrace.py:84
  discounts = np.asarray(discounts, dtype=dtype)
  rewards = np.asarray(rewards, dtype=dtype)
  values = np.asarray(values, dtype=dtype)
  bootstrap_value = np.asarray(bootstrap_value, dtype=dtype)

  # Rank checks, vtrace.py:99-107.
  rho_rank = log_rhos.ndim
  if values.ndim != rho_rank or discounts.ndim != rho_rank or \
     rewards.ndim != rho_rank or bootstrap_value.ndim != rho_rank - 1:
    raise ValueError('inconsistent ranks')

  rhos = np.exp(log_rhos).astype(dtype)                      # :110
  if clip_rho_threshold is not None:
    clipped_rhos = np.minimum(dtype(clip_rho_threshold), rhos)   # :111-114
  else:
    clipped_rhos = rhos
  cs = np.minimum(dtype(1.0), rhos)                          # :116
  cs = (cs * dtype(lambda_)).astype(dtype)                   # :117

  values_t_plus_1 = np.concatenate(
      [values[1:], bootstrap_value[None]], axis=0)           # :120-121
  deltas = (clipped_rhos *
            (rewards + discounts * values_t_plus_1 - values)).astype(dtype)  # :122

  acc = np.zeros_like(bootstrap_value)                       # :124
  out = []
  for i in range(discounts.shape[0] - 1, -1, -1):            # :126-129
    acc = (deltas[i] + discounts[i] * cs[i] * acc).astype(dtype)
    out.append(acc)
  vs_minus_v_xs = np.stack(out[::-1], axis=0) if out else np.zeros_like(values)
  vs = (vs_minus_v_xs + values).astype(dtype)                # :133

  vs_t_plus_1 = np.concatenate([vs[1:], bootstrap_value[None]], axis=0)  # :136-137
  if clip_pg_rho_threshold is not None:
    clipped_pg_rhos = np.minimum(dtype(clip_pg_rho_threshold), rhos)   # :138-142
  else:
    clipped_pg_rhos = rhos
  pg_advantages = (clipped_pg_rhos *
                   (rewards + discounts * vs_t_plus_1 - values)).astype(dtype)  # :143-144
  return VTraceReturns(vs=vs, pg_advantages=pg_advantages)


def ground_truth_calculation(discounts, behaviour_action_log_probs,
                             target_action_log_probs, rewards, values,
                             bootstrap_value, clip_rho_threshold,
                             clip_pg_rho_threshold):
  """O(T^2) formula from the paper, as in tests/vtrace_test.py:41-82 (fp64)."""
  discounts = np.asarray(discounts, np.float64)
  rewards = np.asarray(rewards, np.float64)
  values = np.asarray(values, np.float64)
  bootstrap_value = np.asarray(bootstrap_value, np.float64)
  log_rhos = (np.asarray(target_action_log_probs, np.float64) -
              np.asarray(behaviour_action_log_probs, np.float64))
  seq_len = len(discounts)
  rhos = np.exp(log_rhos)
  cs = np.minimum(rhos, 1.0)
  clipped_rhos = rhos
  if clip_rho_threshold:
    clipped_rhos = np.minimum(rhos, clip_rho_threshold)
  clipped_pg_rhos = rhos
  if clip_pg_rho_threshold:
    clipped_pg_rhos = np.minimum(rhos, clip_pg_rho_threshold)
  values_t_plus_1 = np.concatenate([values, bootstrap_value[None, :]], axis=0)
  vs = []
  for s in range(seq_len):
    v_s = np.copy(values[s])
    for t in range(s, seq_len):
      v_s += (np.prod(discounts[s:t], axis=0) * np.prod(cs[s:t], axis=0) *
              clipped_rhos[t] *
              (rewards[t] + discounts[t] * values_t_plus_1[t + 1] - values[t]))
    vs.append(v_s)
  vs = np.stack(vs, axis=0)
  pg_advantages = clipped_pg_rhos * (
      rewards + discounts *
      np.concatenate([vs[1:], bootstrap_value[None, :]], axis=0) - values)
  return VTraceReturns(vs=vs, pg_advantages=pg_advantages)
