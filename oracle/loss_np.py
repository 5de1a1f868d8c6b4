"""NumPy restatement of the IMPALA loss head (oracle; test infrastructure only).

Follows /root/reference/agents/vtrace/learner.py:82-157 given the learner's
unrolled outputs (policy_logits [T+1,B,A], baseline [T+1,B]); the network
forward (learner.py:75-79) is restated separately in oracle/nets_torch.py.

Gradients wrt the learner outputs are hand-derived and
cross-checked against torch autograd in tests/test_oracle_golden.py.
"""
import collections

import numpy as np

from oracle import categorical_np, vtrace_np

LossOutputs = collections.namedtuple(
    'LossOutputs',
    'total_loss policy_loss v_loss entropy_loss kl_loss entropy kl_mean '
    'value_mean v_l2_error max_action_abs vs pg_advantages '
    'd_policy_logits d_baseline')


def compute_loss_from_outputs(learner_logits, learner_baseline,
                              behaviour_logits, actions, rewards, done,
                              entropy_cost=0.00025, baseline_cost=0.5,
                              kl_cost=0.0, discounting=0.99, lambda_=1.0,
                              max_abs_reward=0.0, dtype=np.float32,
                              mean_denominator=None):
  """learner.py:82-135 + the logged scalars of :138-157.

  Args (all time-major with T+1 steps, exactly what compute_loss receives):
    learner_logits   f32[T+1,B,A]  learner_outputs.policy_logits
    learner_baseline f32[T+1,B]    learner_outputs.baseline
    behaviour_logits f32[T+1,B,A]  agent_outputs.policy_logits
    actions          int[T+1,B]    agent_outputs.action
    rewards          f32[T+1,B]    env_outputs.reward
    done             bool[T+1,B]   env_outputs.done
    mean_denominator: N used by the means (default T*B); data-parallel shards
      pass the *global* T*B so that summed shard losses equal the global mean.
  """
  f = dtype
  learner_logits = np.asarray(learner_logits, f)
  learner_baseline = np.asarray(learner_baseline, f)
  behaviour_logits = np.asarray(behaviour_logits, f)
  actions = np.asarray(actions)
  rewards = np.asarray(rewards, f)
  done = np.asarray(done).astype(bool)
  T1, B, A = learner_logits.shape
  T = T1 - 1
  n = f(mean_denominator if mean_denominator is not None else T * B)

  bootstrap_value = learner_baseline[-1]                        # :82
  tgt_logits = learner_logits[:-1]                              # :88
  values = learner_baseline[:-1]
  beh_logits = behaviour_logits[:-1]                            # :86
  act = actions[:-1]
  rew = rewards[1:]                                             # :87
  dn = done[1:]
  if max_abs_reward:
    rew = np.clip(rew, f(-max_abs_reward), f(max_abs_reward))   # :90-92
  discounts = ((~dn).astype(f) * f(discounting)).astype(f)      # :93

  tgt_lp = categorical_np.log_prob(tgt_logits, act, f)          # :95-96
  beh_lp = categorical_np.log_prob(beh_logits, act, f)          # :97-98
  vt = vtrace_np.from_importance_weights(
      tgt_lp, beh_lp, discounts, rew, values, bootstrap_value,
      lambda_=lambda_, dtype=f)                                 # :101-108

  policy_loss = -np.sum(tgt_lp * vt.pg_advantages, dtype=f) / n     # :111-112
  v_error = (vt.vs - values).astype(f)                              # :115
  v_loss = f(baseline_cost) * f(0.5) * (np.sum(v_error * v_error, dtype=f) / n)
  ent = categorical_np.entropy(tgt_logits, f)
  entropy = np.sum(ent, dtype=f) / n                                # :119-120
  entropy_loss = f(entropy_cost) * -entropy                         # :121
  kl = (beh_lp - tgt_lp).astype(f)                                  # :124
  kl_mean = np.sum(kl, dtype=f) / n
  kl_loss = f(kl_cost) * kl_mean                                    # :125
  total = policy_loss + v_loss + entropy_loss + kl_loss             # :134-135

  # Hand-derived gradients.
  ls = categorical_np.log_softmax(tgt_logits, f)
  p = np.exp(ls).astype(f)
  onehot = (np.arange(A)[None, None, :] == act[..., None]).astype(f)
  coef = (vt.pg_advantages + f(kl_cost)).astype(f)                  # PG + KL
  d_logits = -(coef[..., None] / n) * (onehot - p)
  d_logits = d_logits + (f(entropy_cost) / n) * p * (ls + ent[..., None])
  d_logits_full = np.zeros_like(learner_logits)
  d_logits_full[:-1] = d_logits.astype(f)
  d_baseline = np.zeros_like(learner_baseline)
  d_baseline[:-1] = (f(baseline_cost) * (values - vt.vs) / n).astype(f)

  return LossOutputs(
      total_loss=f(total), policy_loss=f(policy_loss), v_loss=f(v_loss),
      entropy_loss=f(entropy_loss), kl_loss=f(kl_loss), entropy=f(entropy),
      kl_mean=f(kl_mean),
      value_mean=f(np.sum(values, dtype=f) / n),
      v_l2_error=f(np.sqrt(np.sum(v_error * v_error, dtype=f) / n)),
      max_action_abs=np.max(np.abs(act)) if act.size else 0,
      vs=vt.vs, pg_advantages=vt.pg_advantages,
      d_policy_logits=d_logits_full, d_baseline=d_baseline)
