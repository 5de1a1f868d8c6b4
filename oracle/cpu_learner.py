"""Eager PyTorch-CPU fp32 learner step: restatement of the reference learner's
compute_gradients + apply_gradients (/root/reference/agents/vtrace/learner.py:255-280)
for the Atari configs (oracle; also the `cpu_baseline` "port" leg of bench.py).

Same graph as the reference would run (frame stacking -> /255 -> conv torso ->
heads -> unfused loss ops -> per-variable Keras Adam), torch autograd standing in for
tf.GradientTape.  NOT TensorFlow (not installable here): labelled "port".
"""
import time

import numpy as np
import torch

from oracle import nets_torch


class CpuAtariLearner(object):

  def __init__(self, kind, num_actions, seed=0, lr=4.8e-4, decay_steps=10000):
    self.kind, self.A = kind, num_actions
    self.params = nets_torch.to_torch(
        nets_torch.init_params(nets_torch.param_spec(kind, num_actions), seed), requires_grad=True)
    self.opt = nets_torch.KerasAdam(list(self.params.values()), nets_torch.polynomial_decay(lr, decay_steps),
                                    beta_1=0.0, epsilon=3.125e-7)

  def step(self, u, **loss_kw):
    for p in self.params.values():
      p.grad = None
    logits, baseline, new_fs, _ = nets_torch.atari_shallow_unroll(
        self.params, self.kind, self.A, u['prev_actions'], u['reward'], u['done'], u['frames'],
        u['frame_state'])
    total, aux = nets_torch.impala_loss_torch(logits, baseline, u['behaviour_logits'], u['actions'],
                                              u['reward'], u['done'], **loss_kw)
    total.backward()
    self.opt.apply_gradients([p.grad for p in self.params.values()])
    return float(total.detach())


def time_cpu_learner(kind, num_actions, T1, B, steps=3, warmup=1, threads=None, seed=0):
  """Returns (env_frames_per_s, seconds_per_step, threads)."""
  from tests import synth
  if threads:
    torch.set_num_threads(threads)
  u = synth.atari_unroll(seed, T1, B, num_actions)
  u = {k: torch.as_tensor(v) for k, v in u.items()}
  lrn = CpuAtariLearner(kind, num_actions, seed)
  for _ in range(warmup):
    lrn.step(u)
  ts = []
  for _ in range(steps):
    t0 = time.perf_counter()
    lrn.step(u)
    ts.append(time.perf_counter() - t0)
  sec = float(np.median(ts))
  return (T1 - 1) * B / sec, sec, torch.get_num_threads()


class CpuDeepLearner(object):
  """ImpalaDeep (dmlab/networks.py) learner step, eager PyTorch-CPU fp32."""

  def __init__(self, num_actions, obs=(72, 96, 3), seed=0, lr=4.8e-4, decay_steps=10000):
    self.A = num_actions
    self.params = nets_torch.to_torch(
        nets_torch.init_params(nets_torch.param_spec('impala_deep', num_actions, obs), seed), requires_grad=True)
    self.opt = nets_torch.KerasAdam(list(self.params.values()), nets_torch.polynomial_decay(lr, decay_steps),
                                    beta_1=0.0, epsilon=3.125e-7)

  def step(self, u, **loss_kw):
    for p in self.params.values():
      p.grad = None
    logits, baseline, _ = nets_torch.impala_deep_unroll(
        self.params, self.A, u['prev_actions'], u['reward'], u['done'], u['frames'], (u['h0'], u['c0']))
    total, _ = nets_torch.impala_loss_torch(logits, baseline, u['behaviour_logits'], u['actions'], u['reward'],
                                            u['done'], **loss_kw)
    total.backward()
    self.opt.apply_gradients([p.grad for p in self.params.values()])
    return float(total.detach())


def time_cpu_deep_learner(num_actions, T1, B, steps=3, warmup=1, seed=0):
  from tests import synth
  u = {k: torch.as_tensor(v) for k, v in synth.dmlab_unroll(seed, T1, B, num_actions).items()}
  lrn = CpuDeepLearner(num_actions, seed=seed)
  for _ in range(warmup):
    lrn.step(u)
  ts = []
  for _ in range(steps):
    t0 = time.perf_counter()
    lrn.step(u)
    ts.append(time.perf_counter() - t0)
  sec = float(np.median(ts))
  return (T1 - 1) * B / sec, sec, torch.get_num_threads()


class CpuR2D2Learner(object):
  """R2D2 learner step (agents/r2d2/learner.py:572-636), eager PyTorch-CPU fp32."""

  def __init__(self, num_actions, seed=0, burn_in=40, n_steps=5, gamma=0.997, clip_norm=40.0):
    self.A, self.burn_in, self.n_steps, self.gamma, self.clip = num_actions, burn_in, n_steps, gamma, clip_norm
    spec = nets_torch.param_spec('r2d2', num_actions)
    self.params = nets_torch.to_torch(nets_torch.init_params(spec, seed), requires_grad=True)
    self.target = nets_torch.to_torch(nets_torch.init_params(spec, seed))
    self.opt = nets_torch.KerasAdam(list(self.params.values()), lambda step: 4.8e-4, epsilon=1e-3)

  def step(self, u, iw):
    for p in self.params.values():
      p.grad = None
    b = self.burn_in
    def run(pp, lo, hi, fs, core):
      return nets_torch.r2d2_unroll(pp, self.A, u['prev_actions'][lo:hi], u['reward'][lo:hi], u['done'][lo:hi],
                                    u['frames'][lo:hi], fs, core)
    T1 = u['done'].shape[0]
    with torch.no_grad():
      _, fs1, core1 = run(self.params, 0, b, u['frame_state'], (u['h0'], u['c0']))
      _, fs1t, core1t = run(self.target, 0, b, u['frame_state'], (u['h0'], u['c0']))
      out_t, _, _ = run(self.target, b, T1, fs1t, core1t)
    out, _, _ = run(self.params, b, T1, fs1, tuple(x.detach() for x in core1))
    total, _, _ = nets_torch.r2d2_loss_torch(out.q_values, out_t.q_values, u['actions'][b:], u['reward'][b:],
                                             u['done'][b:], iw, self.gamma, self.n_steps)
    total.backward()
    grads = [p.grad for p in self.params.values()]
    gn = float(torch.sqrt(sum((g ** 2).sum() for g in grads)))
    scale = min(1.0, self.clip / max(gn, 1e-30))
    self.opt.apply_gradients([g * scale for g in grads])
    return float(total.detach())


def time_cpu_r2d2_learner(num_actions, T1, B, burn_in=40, steps=2, warmup=1, seed=0):
  from tests import synth
  u = synth.atari_unroll(seed, T1, B, num_actions)
  u = {k: torch.as_tensor(v) for k, v in u.items()}
  u['h0'] = torch.zeros((B, 512)); u['c0'] = torch.zeros((B, 512))
  iw = torch.ones(B)
  lrn = CpuR2D2Learner(num_actions, seed, burn_in=burn_in)
  for _ in range(warmup):
    lrn.step(u, iw)
  ts = []
  for _ in range(steps):
    t0 = time.perf_counter()
    lrn.step(u, iw)
    ts.append(time.perf_counter() - t0)
  sec = float(np.median(ts))
  return (T1 - 1) * B / sec, sec, torch.get_num_threads()
