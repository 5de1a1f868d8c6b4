"""R2D2 learner step on MI355X (BASELINE config 5).

Mirrors /root/reference/agents/r2d2/learner.py on the hot path:
  * `Unroll` (with `priority`)                      learner.py:95-96
  * flag defaults (-> R2D2Config)                   learner.py:43-87; atari/r2d2_main.py:36-39
  * `compute_loss_and_priorities(...)`              learner.py:333-384 (burn-in on both networks without
                                                    gradient, suffix unroll on both, n-step double-Q loss)
  * `R2D2Learner.minimize(unroll, importance_weights)`  learner.py:572-636: reduce_mean(loss * weights),
                                                    global-norm clip, Adam, new priorities; target-network
                                                    copy every `update_target_every_n_step` (:856-857).
The loss is the n-step double-Q Bellman loss with value rescaling -- the reference has no Retrace
(SURVEY.md section 0, D2).  Replay sampling / priority updates (utils.PrioritizedReplay) sit outside this
step: `importance_weights` are an input, new priorities an output.
"""
import collections

import torch

from seed_rl_amd import ops
from seed_rl_amd.learner import all_reduce_gradients

Unroll = collections.namedtuple('Unroll', 'agent_state priority prev_actions env_outputs agent_outputs')


class R2D2Config(object):

  def __init__(self, burn_in=40, n_steps=5, discounting=.997, eta=0.9, value_function_rescaling_epsilon=1e-3,
               clip_norm=40., update_target_every_n_step=2500):
    self.burn_in, self.n_steps, self.discounting, self.eta = burn_in, n_steps, discounting, eta
    self.value_function_rescaling_epsilon = value_function_rescaling_epsilon
    self.clip_norm, self.update_target_every_n_step = clip_norm, update_target_every_n_step


def _split(struct, n):
  """utils.split_structure (utils.py:947-986): (struct[:n], struct[n:]) along time for every tensor."""
  from seed_rl_amd import utils
  return (utils.map_structure(lambda t: None if t is None else t[:n], struct),
          utils.map_structure(lambda t: None if t is None else t[n:], struct))


def compute_loss_and_priorities(training_agent, target_agent, agent_state, prev_actions, env_outputs,
                                agent_outputs, gamma, burn_in, config=None, importance_weights=None,
                                mean_denominator=None):
  """learner.py:333-384 + the weighted mean of :604.  Returns (loss [B], priorities [B], total 0-d tensor);
  the gradient wrt the training network's q-values is left in its 'd_q' buffer for `backward`."""
  cfg = config or R2D2Config()
  if burn_in:
    (pa_pre, env_pre), (pa_suf, env_suf) = _split((prev_actions, env_outputs), burn_in)
    _, ao_suf = _split(agent_outputs, burn_in)
    _, training_state = training_agent((pa_pre, env_pre), agent_state, unroll=True)   # no gradient is taken
    _, target_state = target_agent((pa_pre, env_pre), agent_state, unroll=True)
  else:
    pa_suf, env_suf, ao_suf = prev_actions, env_outputs, agent_outputs
    training_state = target_state = agent_state
  target_out, _ = target_agent((pa_suf, env_suf), target_state, unroll=True)
  training_out, _ = training_agent((pa_suf, env_suf), training_state, unroll=True)     # last: keeps activations
  T, B, A = training_out.q_values.shape
  ag = training_agent
  loss_b = ag._buf('r2d2_loss_b', (B,))
  prio_b = ag._buf('r2d2_prio_b', (B,))
  total = ag._buf('r2d2_total', (1,))
  dq = ag._buf('d_q', (T, B, A))
  ws = ag._buf('r2d2_ws', (ops.r2d2_loss_workspace_bytes(T, B, cfg.n_steps) // 4 + 4,))
  iw = None if importance_weights is None else importance_weights.to(torch.float32).contiguous()
  ops.r2d2_loss_fwd_bwd(
      training_out.q_values.contiguous(), target_out.q_values.contiguous(),
      ao_suf.action.to(torch.int32).contiguous(), env_suf.reward.to(torch.float32).contiguous(),
      ops.as_u8(env_suf.done), iw, T, B, A, gamma, cfg.n_steps, cfg.eta,
      cfg.value_function_rescaling_epsilon, B if mean_denominator is None else mean_denominator,
      loss_b, prio_b, dq, total, ws)
  ag._last['dq'] = dq
  return loss_b, prio_b, total[0]


class R2D2Learner(object):
  """One data-parallel R2D2 learner replica."""

  def __init__(self, training_agent, target_agent, optimizer, config=None, reduction='mean', process_group=None):
    assert reduction in ('mean', 'sum')
    self.agent, self.target_agent, self.optimizer = training_agent, target_agent, optimizer
    self.config = config or R2D2Config()
    self.reduction, self.pg = reduction, process_group
    self.world = 1
    if torch.distributed.is_available() and torch.distributed.is_initialized():
      self.world = torch.distributed.get_world_size(process_group)
    self.iterations = 0
    # one sticky abort word for both networks: a timed-out LSTM sequence kernel of the TARGET network invalidates the
    # training network's gradients just the same, and the update kernel is guarded by the training agent's word
    if hasattr(training_agent, '_seq_sticky') and hasattr(target_agent, '_ws'):
      target_agent._ws[('lstm_seq_sticky', (1,), torch.int32)] = training_agent._seq_sticky()   # pylint: disable=protected-access
    self.update_target()

  def update_target(self):
    """learner.py:856-857 / :640-644: target <- training (one flat copy)."""
    self.target_agent.flat.params.copy_(self.agent.flat.params)

  def after_graph_replay(self):
    """Python-side bookkeeping of a step replayed from a HIP graph (learner.GraphedStep)."""
    self.iterations += 1
    cfg = self.config
    if cfg.update_target_every_n_step and self.iterations % cfg.update_target_every_n_step == 0:
      self.update_target()

  def compute_gradients(self, unroll, importance_weights=None):
    cfg = self.config
    B = unroll.env_outputs.done.shape[1]
    n = B * (self.world if self.reduction == 'mean' else 1)
    loss_b, prio, total = compute_loss_and_priorities(
        self.agent, self.target_agent, unroll.agent_state, unroll.prev_actions, unroll.env_outputs,
        unroll.agent_outputs, cfg.discounting, cfg.burn_in, cfg, importance_weights, mean_denominator=n)
    self.agent.backward()
    self._sumsq = self.agent._buf('gnorm_sumsq', (1,))
    return total, prio, self._sumsq

  def reduce_gradients(self):
    all_reduce_gradients(self.agent.flat.grads, self.pg)

  def update(self):
    flat = self.agent.flat
    gws = self.agent._buf('gnorm_ws', (ops.global_norm_workspace_bytes() // 4 + 4,))
    ops.clip_by_global_norm(flat.grads, self.config.clip_norm or 0.0, self._sumsq, gws)   # learner.py:605-609
    self.optimizer.apply_gradients(flat)

  def minimize(self, unroll, importance_weights=None):
    """learner.py:572-636.  Returns (loss, new priorities [B], gradient norm before clipping)."""
    cfg = self.config
    begin = getattr(self.optimizer, 'begin_step', None)
    if begin is not None and not torch.cuda.is_current_stream_capturing():
      begin(self.agent.flat.params.device)
    total, prio, sumsq = self.compute_gradients(unroll, importance_weights)
    self.reduce_gradients()
    self.update()
    self.iterations += 1
    if cfg.update_target_every_n_step and self.iterations % cfg.update_target_every_n_step == 0:
      self.update_target()
    return total, prio, torch.sqrt(sumsq[0])
