"""Optimizer for the flat parameter buffer.

Mirrors what `create_optimizer_fn` returns in the reference
(/root/reference/dmlab/vtrace_main.py:46-51): Keras `Adam` driven by a
`PolynomialDecay` schedule; the update is ONE fused HIP kernel over the flat
buffer (csrc/adam.hip) instead of one Keras update per variable
(agents/vtrace/learner.py:272-275).
"""
import math

import torch

from seed_rl_amd import ops


class PolynomialDecay(object):
  """tf.keras.optimizers.schedules.PolynomialDecay (power, end_learning_rate)."""

  def __init__(self, initial_learning_rate, decay_steps, end_learning_rate=0.0, power=1.0):
    self.lr0, self.steps, self.end, self.power = initial_learning_rate, decay_steps, end_learning_rate, power

  def __call__(self, step):
    s = min(step, self.steps)
    return (self.lr0 - self.end) * (1 - s / self.steps) ** self.power + self.end


class Adam(object):
  """tf.keras.optimizers.Adam semantics on a FlatParams."""

  def __init__(self, learning_rate=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-7, capturable=False):
    """capturable=True keeps the bias-corrected learning rate in a device scalar (refreshed by `begin_step()`
    outside any HIP-graph capture), so that apply_gradients() can be captured and replayed."""
    self.learning_rate = learning_rate
    self.beta_1, self.beta_2, self.epsilon = beta_1, beta_2, epsilon
    self.iterations = 0
    self._m = self._v = None
    self.capturable = capturable
    self._lr_dev = None

  def lr_t(self):
    t = self.iterations + 1
    return self._lr() * math.sqrt(1.0 - self.beta_2 ** t) / (1.0 - self.beta_1 ** t)

  def begin_step(self, device=None):
    """Publishes this step's learning rate to the device scalar (capturable mode; no-op otherwise)."""
    if not self.capturable:
      return
    if self._lr_dev is None:
      self._lr_dev = torch.zeros(1, dtype=torch.float32, device=device or 'cuda')
    self._lr_dev.fill_(float(self.lr_t()))

  def _lr(self):
    return self.learning_rate(self.iterations) if callable(self.learning_rate) else self.learning_rate

  def apply_gradients(self, flat, grad_scale=1.0):
    if self._m is None:
      self._m = torch.zeros_like(flat.params)
      self._v = torch.zeros_like(flat.params)
    t = self.iterations + 1
    # flat.step_guard (int32[1] on the device, set by agents with LSTM sequence kernels): non-zero = this step's gradients
    # are invalid and the update is dropped on the device (csrc/adam.hip)
    # (such a step still advances `iterations` on the host: the learning-rate schedule and the bias correction move on by
    #  one for an update that did not happen -- accepted for a fallback that fires at most once per agent, see
    #  networks._lstm_seq_check)
    guard = getattr(flat, 'step_guard', None)
    if self.capturable:
      if self._lr_dev is None:
        self.begin_step(flat.params.device)
      ops.adam_flat_dev_lr(flat.params, flat.grads, self._m, self._v, self._lr_dev, self.beta_1, self.beta_2,
                           self.epsilon, float(grad_scale), clamp=getattr(flat, 'constraint', None), guard=guard)
    else:
      ops.adam_flat(flat.params, flat.grads, self._m, self._v, float(self.lr_t()), self.beta_1, self.beta_2,
                    self.epsilon, float(grad_scale), clamp=getattr(flat, 'constraint', None), guard=guard)
    self.iterations = t

  def state_dict(self):
    return dict(iterations=self.iterations, m=self._m, v=self._v)

  def load_state_dict(self, sd):
    self.iterations, self._m, self._v = sd['iterations'], sd['m'], sd['v']
