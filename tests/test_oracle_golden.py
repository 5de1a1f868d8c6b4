"""Pins the CPU oracle against the reference's own known-answer tests.

Every case cites the reference test it restates.  CPU only (no GPU, no HIP).
"""
import numpy as np
import pytest
import torch

from oracle import categorical_np, frames_np, loss_np, nets_torch, r2d2_np, vtrace_np


def _shaped_arange(*shape):
  return np.arange(np.prod(shape), dtype=np.float32).reshape(*shape)


def _ref_vtrace_inputs():
  # /root/reference/tests/vtrace_test.py:120-141
  batch_size, seq_len = 5, 5
  log_rhos = _shaped_arange(seq_len, batch_size) / (batch_size * seq_len)
  log_rhos = 5 * (log_rhos - 0.5)
  return dict(
      behaviour_action_log_probs=np.zeros_like(log_rhos),
      target_action_log_probs=log_rhos,
      discounts=np.array([[0.9 / (b + 1) for b in range(batch_size)]
                          for _ in range(seq_len)]),
      rewards=_shaped_arange(seq_len, batch_size),
      values=_shaped_arange(seq_len, batch_size) / batch_size,
      bootstrap_value=_shaped_arange(batch_size) + 1.0,
      clip_rho_threshold=3.7, clip_pg_rho_threshold=2.2)


def test_vtrace_matches_reference_ground_truth():
  """tests/vtrace_test.py:120-145 (assertAllClose rtol=atol=1e-6)."""
  v = _ref_vtrace_inputs()
  out = vtrace_np.from_importance_weights(**v)
  gt = vtrace_np.ground_truth_calculation(**v)
  np.testing.assert_allclose(out.vs, gt.vs, rtol=1e-6, atol=1e-6)
  np.testing.assert_allclose(out.pg_advantages, gt.pg_advantages,
                             rtol=1e-6, atol=1e-6)
  assert out.vs.dtype == np.float32
  # Values recorded in SURVEY.md section 8(c) item 1 (fp64 ground truth).
  np.testing.assert_allclose(
      gt.vs[0], [0.929816, 0.556437, 0.816895, 1.189767, 1.654996], atol=2e-6)
  np.testing.assert_allclose(gt.vs[4], [66.53, 69.69, 72.85, 76.01, 79.17], atol=6e-3)
  np.testing.assert_allclose(
      gt.pg_advantages[0], [0.929816, 0.356437, 0.416895, 0.589767, 0.854995],
      atol=2e-6)
  assert abs(gt.vs.sum() - 930.41754) < 1e-3
  assert abs(gt.pg_advantages.sum() - 824.18212) < 1e-3
  # fp32 same-order recursion vs fp64 truth: ~7.6e-6.
  assert np.max(np.abs(out.vs - gt.vs)) < 1e-5


def test_vtrace_none_thresholds_and_rank_checks():
  """vtrace.py:91-96,99-107,111-114,138-142."""
  v = _ref_vtrace_inputs()
  v['clip_rho_threshold'] = None
  v['clip_pg_rho_threshold'] = None
  out = vtrace_np.from_importance_weights(**v)
  gt = vtrace_np.ground_truth_calculation(**v)
  np.testing.assert_allclose(out.vs, gt.vs, rtol=2e-6, atol=1e-4)
  with pytest.raises(ValueError):
    bad = dict(v); bad['bootstrap_value'] = v['values']
    vtrace_np.from_importance_weights(**bad)


def test_vtrace_extra_dims():
  """vtrace.py:49-51: [T,B,C] inputs behave as independent columns."""
  rng = np.random.default_rng(0)
  T, B, C = 7, 3, 2
  a = {k: rng.uniform(-1, 1, (T, B, C)).astype(np.float32)
       for k in ['tgt', 'beh', 'rew', 'val']}
  disc = (rng.uniform(size=(T, B, C)) > 0.1).astype(np.float32) * 0.99
  boot = rng.uniform(size=(B, C)).astype(np.float32)
  o = vtrace_np.from_importance_weights(a['tgt'], a['beh'], disc, a['rew'], a['val'], boot)
  o2 = vtrace_np.from_importance_weights(
      a['tgt'].reshape(T, -1), a['beh'].reshape(T, -1), disc.reshape(T, -1),
      a['rew'].reshape(T, -1), a['val'].reshape(T, -1), boot.reshape(-1))
  np.testing.assert_array_equal(o.vs.reshape(T, -1), o2.vs)


def test_log_probs_from_logits_and_actions():
  """tests/vtrace_test.py:88-115."""
  batch_size, seq_len, num_actions = 2, 7, 3
  logits = _shaped_arange(seq_len, batch_size, num_actions) + 10
  actions = np.random.default_rng(0).integers(
      0, num_actions - 1, size=(seq_len, batch_size)).astype(np.int32)
  lp = categorical_np.log_prob(logits, actions)
  sm = np.exp(logits) / np.sum(np.exp(logits), axis=-1, keepdims=True)
  mask = actions[..., None] == np.arange(num_actions)
  gt = np.log(sm)[mask].reshape(seq_len, batch_size)
  np.testing.assert_allclose(lp, gt, rtol=1e-6, atol=1e-6)


def test_entropy_matches_torch():
  rng = np.random.default_rng(1)
  logits = rng.normal(size=(5, 4, 6)).astype(np.float32) * 3
  ent = categorical_np.entropy(logits)
  ref = torch.distributions.Categorical(logits=torch.tensor(logits)).entropy().numpy()
  np.testing.assert_allclose(ent, ref, rtol=1e-5, atol=1e-6)


def test_policy_gradient_vtrace_cross_check_distribution():
  """agents/policy_gradient/modules/advantages_test.py:128-150 input
  distribution: fp32 recursion vs O(T^2) truth on random data, <=1e-5."""
  rng = np.random.default_rng(0)
  T, B = 20, 10
  values = rng.uniform(0, 3, (T, B)).astype(np.float32)
  rewards = rng.uniform(0, 3, (T, B)).astype(np.float32)
  tgt = rng.uniform(-2, 2, (T, B)).astype(np.float32)
  beh = rng.uniform(-2, 2, (T, B)).astype(np.float32)
  done = rng.uniform(size=(T, B)) < 0.05
  disc = (0.99 * (~done)).astype(np.float32)
  boot = rng.uniform(0, 3, B).astype(np.float32)
  o = vtrace_np.from_importance_weights(tgt, beh, disc, rewards, values, boot,
                                        lambda_=0.95)
  o64 = vtrace_np.from_importance_weights(tgt, beh, disc, rewards, values, boot,
                                          lambda_=0.95, dtype=np.float64)
  assert np.max(np.abs(o.vs - o64.vs)) < 1e-5
  assert np.max(np.abs(o.pg_advantages - o64.pg_advantages)) < 1e-5


# ---- frame stacking / unroll cell: atari/networks_test.py:119-247 ---------- #
def _stack_fn(input_t, state):
  # atari/networks_test.py:30-52 (stack_fn): keeps last 3 inputs as state.
  new_state = tuple(state[1:]) + (input_t,)
  return np.concatenate(list(state) + [input_t], axis=-1), new_state


def test_unroll_cell():
  zero = (np.array([[0]]),) * 3
  out, st = frames_np.unroll_cell([[[1]]], [[False]], zero, zero, _stack_fn)
  np.testing.assert_array_equal(out, [[[0, 0, 0, 1]]])
  out, st = frames_np.unroll_cell([[[2]]], [[False]], st, zero, _stack_fn)
  np.testing.assert_array_equal(out, [[[0, 0, 1, 2]]])
  out, st = frames_np.unroll_cell(
      [[[3]], [[4]], [[5]], [[6]], [[7]], [[8]]], [[False]] * 6, st, zero, _stack_fn)
  np.testing.assert_array_equal(out[0], [[0, 1, 2, 3]])
  np.testing.assert_array_equal(out[5], [[5, 6, 7, 8]])


def test_unroll_cell_done():
  zero = (np.array([[0]]),) * 3
  out, st = frames_np.unroll_cell([[[1]]], [[False]], zero, zero, _stack_fn)
  out, st = frames_np.unroll_cell([[[2]]], [[True]], st, zero, _stack_fn)
  np.testing.assert_array_equal(out, [[[0, 0, 0, 2]]])
  out, st = frames_np.unroll_cell(
      [[[3]], [[4]], [[5]], [[6]], [[7]], [[8]]],
      [[False], [False], [False], [False], [True], [False]], st, zero, _stack_fn)
  np.testing.assert_array_equal(out[0], [[0, 0, 2, 3]])
  np.testing.assert_array_equal(out[5], [[0, 0, 7, 8]])


def test_stack_frames():
  st = frames_np.initial_frame_stacking_state(4, 1, [1])
  out, st = frames_np.stack_frames([[[1]]], st, [[False]], 4)
  np.testing.assert_array_equal(out, [[[1, 0, 0, 0]]])
  out, st = frames_np.stack_frames([[[2]]], st, [[False]], 4)
  np.testing.assert_array_equal(out, [[[2, 1, 0, 0]]])
  out, st = frames_np.stack_frames(
      [[[3]], [[4]], [[5]], [[6]], [[7]], [[8]]], st, [[False]] * 6, 4)
  assert out.shape[0] == 6
  np.testing.assert_array_equal(out[0], [[3, 2, 1, 0]])
  np.testing.assert_array_equal(out[5], [[8, 7, 6, 5]])


def test_stack_frames_done():
  st = frames_np.initial_frame_stacking_state(4, 1, [1])
  out, st = frames_np.stack_frames([[[1]]], st, [[False]], 4)
  out, st = frames_np.stack_frames([[[2]]], st, [[True]], 4)
  np.testing.assert_array_equal(out, [[[2, 0, 0, 0]]])
  out, st = frames_np.stack_frames(
      [[[3]], [[4]], [[5]], [[6]], [[7]], [[8]]], st,
      [[False], [False], [False], [False], [True], [False]], 4)
  np.testing.assert_array_equal(out[0], [[3, 2, 0, 0]])
  np.testing.assert_array_equal(out[5], [[8, 7, 0, 0]])


def test_stack_frames_torch_equals_numpy():
  rng = np.random.default_rng(3)
  T, B, H, W = 6, 3, 5, 4
  fr = rng.integers(0, 256, (T, B, H, W, 1)).astype(np.uint8)
  dn = rng.uniform(size=(T, B)) < 0.3
  st = rng.integers(0, 2 ** 24, (B, H * W)).astype(np.int32)
  o, s = frames_np.stack_frames(fr, st, dn, 4)
  o2, s2 = nets_torch.stack_frames_torch(torch.tensor(fr), torch.tensor(st),
                                         torch.tensor(dn), 4)
  np.testing.assert_array_equal(o, o2.numpy())
  np.testing.assert_array_equal(s, s2.numpy())


# ---- R2D2 math: agents/r2d2/learner_test.py:114-198 ------------------------ #
def test_value_function_rescaling():
  for x in np.linspace(-100., 100.):
    np.testing.assert_allclose(
        r2d2_np.inverse_value_function_rescaling(
            r2d2_np.value_function_rescaling(x, dtype=np.float64), dtype=np.float64),
        x, rtol=1e-6, atol=1e-6)
  assert r2d2_np.value_function_rescaling(0.) == 0
  assert r2d2_np.value_function_rescaling(1000.) > 10.
  assert r2d2_np.value_function_rescaling(-1000.) < -10.
  assert r2d2_np.inverse_value_function_rescaling(0.) == 0
  np.testing.assert_allclose(
      r2d2_np.value_function_rescaling([0., 3., -3.]), [0., 1 + 3e-3, -1 - 3e-3],
      rtol=1e-6)
  np.testing.assert_allclose(
      r2d2_np.inverse_value_function_rescaling([0., 1 + 3e-3, -1 - 3e-3]),
      [0., 3, -3], atol=2e-4)


@pytest.mark.parametrize('rewards,done,q,n,expected', [
    ([1., 2., 3.], [False] * 3, [100, 200, 300], 1,
     [1 + 0.9 * 100, 2 + 0.9 * 200, 3 + 0.9 * 300]),
    ([1., 2., 3.], [False, True, False], [100, 200, 300], 1,
     [1 + 0.9 * 100, 2, 3 + 0.9 * 300]),
    ([1., 2., 3.], [False] * 3, [100, 200, 300], 2,
     [1 + 0.9 * 2 + 0.9 ** 2 * 200, 2 + 0.9 * 3 + 0.9 ** 2 * 300, 3 + 0.9 * 300]),
    ([1., 2., 3., 4., 5., 6., 7.], [False, False, False, True, False, False, False],
     [100, 200, 300, 400, 500, 600, 700], 3,
     [1 + 0.9 * 2 + 0.9 ** 2 * 3 + 0.9 ** 3 * 300, 2 + 0.9 * 3 + 0.9 ** 2 * 4,
      3 + 0.9 * 4, 4, 5 + 0.9 * 6 + 0.9 ** 2 * 7 + 0.9 ** 3 * 700,
      6 + 0.9 * 7 + 0.9 ** 2 * 700, 7 + 0.9 * 700]),
])
def test_n_step_bellman_target(rewards, done, q, n, expected):
  t = r2d2_np.n_step_bellman_target(
      np.array([rewards], np.float32).T, np.array([done]).T,
      np.array([q], np.float32).T, 0.9, n)
  np.testing.assert_allclose(t, np.array([expected]).T, rtol=1e-6)


# ---- loss head: hand-derived gradients vs torch autograd -------------------- #
def _loss_inputs(seed, T=20, B=32, A=6):
  rng = np.random.default_rng(seed)
  beh = rng.normal(size=(T + 1, B, A)).astype(np.float32)
  tgt = (beh + 0.5 * rng.normal(size=beh.shape)).astype(np.float32)
  p = np.exp(beh) / np.exp(beh).sum(-1, keepdims=True)
  act = np.array([[rng.choice(A, p=p[t, b] / p[t, b].sum()) for b in range(B)]
                  for t in range(T + 1)]).astype(np.int64)
  rew = rng.uniform(0, 3, (T + 1, B)).astype(np.float32)
  base = rng.uniform(0, 3, (T + 1, B)).astype(np.float32)
  done = rng.uniform(size=(T + 1, B)) < 0.05
  return tgt, base, beh, act, rew, done


@pytest.mark.parametrize('kw', [
    dict(), dict(lambda_=0.95, kl_cost=0.1, max_abs_reward=1.0, entropy_cost=0.01)])
def test_loss_head_matches_torch_autograd(kw):
  tgt, base, beh, act, rew, done = _loss_inputs(0)
  out = loss_np.compute_loss_from_outputs(tgt, base, beh, act, rew, done, **kw)
  tl = torch.tensor(tgt, requires_grad=True)
  tb = torch.tensor(base, requires_grad=True)
  total, aux = nets_torch.impala_loss_torch(
      tl, tb, torch.tensor(beh), torch.tensor(act), torch.tensor(rew),
      torch.tensor(done), **kw)
  total.backward()
  assert abs(float(total) - float(out.total_loss)) < 1e-5 * max(1, abs(float(total)))
  np.testing.assert_allclose(out.vs, aux['vs'].numpy(), atol=1e-5)
  np.testing.assert_allclose(out.pg_advantages, aux['pg_advantages'].numpy(), atol=1e-5)
  np.testing.assert_allclose(out.d_policy_logits, tl.grad.numpy(), atol=2e-7, rtol=1e-4)
  np.testing.assert_allclose(out.d_baseline, tb.grad.numpy(), atol=2e-7, rtol=1e-4)
  assert np.all(out.d_baseline[-1] == 0) and np.all(out.d_policy_logits[-1] == 0)


# ---- network structure pins -------------------------------------------------- #
def test_impala_deep_structure():
  """tests/agents_test.py:45 (39 trainable tensors) + SURVEY a4 param count."""
  spec = nets_torch.param_spec('impala_deep', 9)
  assert len(spec) == 39
  assert sum(int(np.prod(s)) for _, s, _ in spec) == 1520714
  assert sum(int(np.prod(s)) for _, s, _ in
             nets_torch.param_spec('atari_shallow', 18)) == 681027


def test_impala_deep_forward_runs_and_tf_same_pool():
  p = nets_torch.to_torch(nets_torch.init_params(nets_torch.param_spec('impala_deep', 9)))
  T1, B = 2, 2
  rng = np.random.default_rng(0)
  fr = torch.tensor(rng.integers(0, 256, (T1, B, 72, 96, 3)).astype(np.uint8))
  lg, bl, st = nets_torch.impala_deep_unroll(
      p, 9, torch.zeros(T1, B, dtype=torch.int64), torch.zeros(T1, B),
      torch.zeros(T1, B, dtype=torch.bool), fr,
      (torch.zeros(B, 256), torch.zeros(B, 256)))
  assert lg.shape == (T1, B, 9) and bl.shape == (T1, B)
  # TF SAME 3/2 pool on even size: window i covers rows [2i, 2i+2] (pad after).
  x = torch.arange(16.).reshape(1, 4, 4, 1)
  y = nets_torch.max_pool_3x3_s2_same(x)
  np.testing.assert_array_equal(y[0, :, :, 0].numpy(), [[10, 11], [14, 15]])


def test_r2d2_core_input_width():
  """atari/networks_test.py:105-117: LSTM input = 512 + num_actions + 1."""
  spec = dict((n, s) for n, s, _ in nets_torch.param_spec('r2d2', 37))
  assert spec['core/kernel'][0] == 512 + 37 + 1
