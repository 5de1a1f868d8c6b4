"""NumPy restatement of the R2D2 loss math (oracle; test infrastructure only).

Follows /root/reference/agents/r2d2/learner.py:180-192 (value rescaling h,
h^-1; eps=1e-3 flag default :77-79), :195-255 (n-step Bellman target) and
:258-330 (double-Q loss + priorities).  Pinned by
agents/r2d2/learner_test.py:114-198.  NOTE: this is n-step double-Q, not
Retrace.
"""
import numpy as np

f32 = np.float32


def value_function_rescaling(x, eps=1e-3, dtype=np.float32):
  x = np.asarray(x, dtype)
  return (np.sign(x) * (np.sqrt(np.abs(x) + dtype(1.)) - dtype(1.)) +
          dtype(eps) * x).astype(dtype)


def inverse_value_function_rescaling(x, eps=1e-3, dtype=np.float32):
  x = np.asarray(x, dtype)
  e = dtype(eps)
  inner = np.sqrt(dtype(1.) + dtype(4.) * e * (np.abs(x) + dtype(1.) + e))
  return (np.sign(x) * (np.square((inner - dtype(1.)) / (dtype(2.) * e)) -
                        dtype(1.))).astype(dtype)


def n_step_bellman_target(rewards, done, q_target, gamma, n_steps,
                          dtype=np.float32):
  """learner.py:195-255."""
  rewards = np.asarray(rewards, dtype)
  done = np.asarray(done).astype(bool)
  q_target = np.asarray(q_target, dtype)
  g = dtype(gamma)
  bellman = np.concatenate(
      [np.zeros_like(q_target[0:1]), q_target] +
      [(q_target[-1:] / dtype(gamma ** k)).astype(dtype)
       for k in range(1, n_steps)], axis=0)
  done = np.concatenate([done] + [np.zeros_like(done[0:1])] * n_steps, axis=0)
  rewards = np.concatenate(
      [rewards] + [np.zeros_like(rewards[0:1])] * n_steps, axis=0)
  for _ in range(n_steps):
    rewards = rewards[:-1]
    done = done[:-1]
    bellman = (rewards + g * (dtype(1.) - done.astype(dtype)) *
               bellman[1:]).astype(dtype)
  return bellman


def loss_and_priorities(training_q, target_q, rewards, done, actions,
                        gamma=0.997, n_steps=5, eta=0.9, eps=1e-3,
                        dtype=np.float32):
  """learner.py:258-330 given q-values.

  training_q, target_q: f32[T,B,A]; rewards f32[T,B]; done bool[T,B];
  actions int[T,B] (replayed actions).  Returns (loss[B], priorities[B],
  d_training_q[T,B,A] for sum(loss*w) with w=1).
  """
  training_q = np.asarray(training_q, dtype)
  target_q = np.asarray(target_q, dtype)
  T, B, A = training_q.shape
  act = np.asarray(actions).astype(np.int64)
  replay_q = np.take_along_axis(training_q, act[..., None], -1)[..., 0]
  best = np.argmax(training_q, axis=-1)                      # agent _head argmax
  qt = np.take_along_axis(target_q, best[..., None], -1)[..., 0]
  qtarget_max = inverse_value_function_rescaling(qt, eps, dtype)
  bt = n_step_bellman_target(rewards, done, qtarget_max, gamma, n_steps, dtype)
  bt = bt[1:]
  rq = replay_q[:-1]
  bt = value_function_rescaling(bt, eps, dtype)
  td = (bt - rq).astype(dtype)
  abs_td = np.abs(td)
  prio = (dtype(eta) * abs_td.max(axis=0) +
          dtype(1 - eta) * abs_td.mean(axis=0, dtype=dtype)).astype(dtype)
  loss = (dtype(0.5) * np.sum(abs_td * abs_td, axis=0, dtype=dtype)).astype(dtype)
  d_q = np.zeros_like(training_q)
  onehot = (np.arange(A)[None, None] == act[:-1][..., None])
  d_q[:-1] = np.where(onehot, (-td)[..., None], dtype(0))
  return loss, prio, d_q
