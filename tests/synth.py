"""Seeded synthetic inputs; numpy.random.default_rng(seed)."""
import numpy as np


def vtrace_inputs(seed, T=20, B=32, A=6, stress=False):
  """cfg1: distribution of agents/policy_gradient/modules/advantages_test.py:129-137."""
  rng = np.random.default_rng(seed)
  beh_logits = rng.normal(size=(T, B, A)).astype(np.float32)
  tgt_logits = (beh_logits + 0.5 * rng.normal(size=(T, B, A))).astype(np.float32)
  p = np.exp(beh_logits - beh_logits.max(-1, keepdims=True))
  p /= p.sum(-1, keepdims=True)
  u = rng.uniform(size=(T, B, 1))
  actions = np.minimum((p.cumsum(-1) < u).sum(-1), A - 1).astype(np.int64)
  def lp(l):
    z = l - l.max(-1, keepdims=True)
    ls = z - np.log(np.exp(z).sum(-1, keepdims=True))
    return np.take_along_axis(ls, actions[..., None], -1)[..., 0].astype(np.float32)
  tgt_lp, beh_lp = lp(tgt_logits), lp(beh_logits)
  if stress:   # log-rho in [-2.5, 2.5) as tests/vtrace_test.py:128-129
    beh_lp = np.zeros((T, B), np.float32)
    tgt_lp = rng.uniform(-2.5, 2.5, (T, B)).astype(np.float32)
  done = rng.uniform(size=(T, B)) < 0.05
  return dict(
      target_action_log_probs=tgt_lp, behaviour_action_log_probs=beh_lp,
      discounts=(0.99 * (~done)).astype(np.float32),
      rewards=rng.uniform(0, 3, (T, B)).astype(np.float32),
      values=rng.uniform(0, 3, (T, B)).astype(np.float32),
      bootstrap_value=rng.uniform(0, 3, (B,)).astype(np.float32))


def loss_inputs(seed, T=20, B=32, A=6, action_dtype=np.int64):
  """T+1-step learner/behaviour outputs for compute_loss (learner.py:73-74)."""
  rng = np.random.default_rng(seed)
  beh = rng.normal(size=(T + 1, B, A)).astype(np.float32)
  tgt = (beh + 0.5 * rng.normal(size=beh.shape)).astype(np.float32)
  p = np.exp(beh - beh.max(-1, keepdims=True))
  p /= p.sum(-1, keepdims=True)
  u = rng.uniform(size=(T + 1, B, 1))
  act = np.minimum((p.cumsum(-1) < u).sum(-1), A - 1).astype(action_dtype)
  rew = rng.uniform(0, 3, (T + 1, B)).astype(np.float32)
  base = rng.uniform(0, 3, (T + 1, B)).astype(np.float32)
  done = rng.uniform(size=(T + 1, B)) < 0.05
  return tgt, base, beh, act, rew, done


def atari_unroll(seed, T1=21, B=4, A=18, H=84, W=84, done_p=0.01, zero_state=True):
  """cfg2-shaped unroll: uint8 frames, packed stacking state, etc."""
  rng = np.random.default_rng(seed)
  frames = rng.integers(0, 256, (T1, B, H, W, 1)).astype(np.uint8)
  state = np.zeros((B, H * W), np.int32) if zero_state else \
      rng.integers(0, 2 ** 24, (B, H * W)).astype(np.int32)
  done = rng.uniform(size=(T1, B)) < done_p
  prev_actions = rng.integers(0, A, (T1, B)).astype(np.int64)
  actions = rng.integers(0, A, (T1, B)).astype(np.int64)
  reward = rng.normal(size=(T1, B)).astype(np.float32)
  beh_logits = rng.normal(size=(T1, B, A)).astype(np.float32)
  beh_baseline = rng.normal(size=(T1, B)).astype(np.float32)
  return dict(frames=frames, frame_state=state, done=done, prev_actions=prev_actions, actions=actions,
              reward=reward, behaviour_logits=beh_logits, behaviour_baseline=beh_baseline)


def dmlab_unroll(seed, T1=21, B=4, A=9, H=72, W=96, done_p=0.05, H_lstm=256):
  """cfg3-shaped unroll: uint8 RGB frames, LSTM initial state ~N(0,0.1)."""
  rng = np.random.default_rng(seed)
  return dict(
      frames=rng.integers(0, 256, (T1, B, H, W, 3)).astype(np.uint8),
      done=rng.uniform(size=(T1, B)) < done_p,
      prev_actions=rng.integers(0, A, (T1, B)).astype(np.int64),
      actions=rng.integers(0, A, (T1, B)).astype(np.int64),
      reward=(2 * rng.normal(size=(T1, B))).astype(np.float32),
      behaviour_logits=rng.normal(size=(T1, B, A)).astype(np.float32),
      behaviour_baseline=rng.normal(size=(T1, B)).astype(np.float32),
      h0=(0.1 * rng.normal(size=(B, H_lstm))).astype(np.float32),
      c0=(0.1 * rng.normal(size=(B, H_lstm))).astype(np.float32))
