"""SURVEY 8(f) rank 4 agents: GFootball (football/networks.py:68-150) and MLPandLSTM (agents/vtrace/networks.py:25-121)
-- train-step parity (unroll -> fused loss -> backward -> Adam) against the torch-CPU oracle of the same graphs, the
bit-plane unpacking against football/observation.py's patterns, single-step == first unroll step."""
import numpy as np
import pytest
import torch

from oracle import nets_torch

pytestmark = pytest.mark.gpu


def _to(device, a):
  return torch.as_tensor(np.ascontiguousarray(a)).to(device)


def test_unpackbits_matches_observation_py(device):
  from seed_rl_amd import ops
  rng = np.random.default_rng(0)
  w = rng.integers(0, 65536, (3, 5, 7, 2)).astype(np.uint16)
  w[0, 0, 0, 0] = 0x8001
  out = torch.empty((3, 5, 7, 32), dtype=torch.uint8, device=device)
  ops.unpackbits_u16(_to(device, w.view(np.int16)), out)
  ref = nets_torch.unpackbits(torch.tensor(w.astype(np.int32)))
  np.testing.assert_array_equal(out.cpu().numpy(), ref.numpy().astype(np.uint8))
  # np.packbits of 16 binary planes viewed as one little-endian uint16 -- how the environment packs (observation.py:36-46)
  planes = rng.integers(0, 2, (4, 6, 16)).astype(np.uint8)
  packed = np.packbits(planes, axis=-1).view(np.uint16)
  out = torch.empty((4, 6, 16), dtype=torch.uint8, device=device)
  ops.unpackbits_u16(_to(device, packed.view(np.int16)), out)
  np.testing.assert_array_equal(out.cpu().numpy(), planes * 255)


def _loss_inputs(rng, T1, B, A):
  return dict(actions=rng.integers(0, A, (T1, B)).astype(np.int64), beh=rng.normal(size=(T1, B, A)).astype(np.float32),
              reward=rng.normal(size=(T1, B)).astype(np.float32), done=rng.uniform(size=(T1, B)) < 0.15,
              prev=rng.integers(0, A, (T1, B)).astype(np.int64))


def _check(agent, lrn, p, total, loss, lr, tol_grad):
  assert abs(float(loss) - float(total.detach())) <= 2e-4 * max(1.0, abs(float(total.detach())))
  grads = agent.reference_gradients()
  for n, t in p.items():
    g, r = grads[n].cpu().numpy(), t.grad.numpy()
    assert np.max(np.abs(g - r)) <= tol_grad(n) * max(np.abs(r).max(), 1e-3), n
  lrn.apply_gradients()
  kopt = nets_torch.KerasAdam(list(p.values()), lambda step: lr, beta_1=0.9, epsilon=1e-7)
  kopt.apply_gradients([t.grad for t in p.values()])
  for (n, v), t in zip(agent.trainable_variables, p.values()):
    assert np.max(np.abs(v.cpu().numpy() - t.detach().numpy())) < 5e-5 + 2 * lr * n.startswith('stack'), n


def test_gfootball_train_step_parity(device):
  from seed_rl_amd import learner, networks, optimizers, utils, parametric_distribution as pd
  T1, B, A, obs = 4, 3, 19, (24, 32, 1)
  rng = np.random.default_rng(2)
  u = _loss_inputs(rng, T1, B, A)
  frames = rng.integers(0, 65536, (T1, B) + obs).astype(np.uint16)
  agent = networks.GFootball(A, observation_shape=obs, device=device, seed=4)
  spec = nets_torch.param_spec('gfootball', A, obs)
  ref = nets_torch.init_params(spec, seed=4)
  assert len(agent.trainable_variables) == 4 * 10 + 6                                # 4 stacks x 5 convs x (kernel, bias) + 3 Dense
  for (n, v) in agent.trainable_variables:
    np.testing.assert_array_equal(v.cpu().numpy(), ref[n])
  assert abs(float(np.std(ref['stack0/conv/kernel'])) - (1.0 / (9 * 16)) ** 0.5) < 0.01   # lecun_normal: std = sqrt(1 / fan_in)
  lr = 1e-3
  lrn = learner.Learner(agent, optimizers.Adam(lr), pd.categorical_distribution(A))
  env = utils.EnvOutput(_to(device, u['reward']), _to(device, u['done']), _to(device, frames.view(np.int16)), None, None)
  ao = networks.AgentOutput(_to(device, u['actions']), _to(device, u['beh']), _to(device, u['reward'] * 0.5))
  unroll = learner.Unroll((), _to(device, u['prev']), env, ao)
  loss, _ = lrn.compute_gradients(unroll)
  p = nets_torch.to_torch(ref, requires_grad=True)
  t = lambda a: torch.tensor(a)
  logits, baseline = nets_torch.gfootball_unroll(p, A, t(frames.astype(np.int32)))
  total, _ = nets_torch.impala_loss_torch(logits, baseline, t(u['beh']), t(u['actions']), t(u['reward']), t(u['done']))
  total.backward()
  head, _, ldh = agent.head_buffers()
  head = head.cpu().numpy().reshape(T1, B, ldh)
  assert np.max(np.abs(head[..., :A] - logits.detach().numpy())) < 3e-4
  _check(agent, lrn, p, total, loss, lr, lambda n: 1e-2 if n.startswith('stack') and not n.startswith('stack3/res') else 1e-3)
  # single step == first step of the unroll; no recurrent state
  env1 = utils.EnvOutput(env.reward[0], env.done[0], env.observation[0], None, None)
  out1, st = agent(unroll.prev_actions[0], env1, ())
  assert st == () and out1.action.shape == (B,) and int(out1.action.max()) < A


@pytest.mark.parametrize('mlp,lstm,obs_dim', [((64, 32), (64,), 17), ((), (32, 64), 12), ((128,), (64, 64), 11)])
def test_mlp_and_lstm_train_step_parity(device, mlp, lstm, obs_dim):
  from seed_rl_amd import learner, networks, optimizers, utils, parametric_distribution as pd
  T1, B, A = 7, 5, 6
  rng = np.random.default_rng(5)
  u = _loss_inputs(rng, T1, B, A)
  obs = rng.normal(size=(T1, B, obs_dim)).astype(np.float32)
  state = [((0.1 * rng.normal(size=(B, h))).astype(np.float32), (0.1 * rng.normal(size=(B, h))).astype(np.float32)) for h in lstm]
  agent = networks.MLPandLSTM(A, obs_dim, mlp, lstm, device=device, seed=6)
  ref = nets_torch.init_params(nets_torch.param_spec('mlp_lstm', A, core=(obs_dim, mlp, lstm)), seed=6)
  for (n, v) in agent.trainable_variables:
    np.testing.assert_array_equal(v.cpu().numpy(), ref[n])
  lr = 1e-3
  lrn = learner.Learner(agent, optimizers.Adam(lr), pd.categorical_distribution(A))
  env = utils.EnvOutput(_to(device, u['reward']), _to(device, u['done']), _to(device, obs), None, None)
  ao = networks.AgentOutput(_to(device, u['actions']), _to(device, u['beh']), _to(device, u['reward'] * 0.5))
  dstate = tuple((_to(device, h), _to(device, c)) for h, c in state)
  unroll = learner.Unroll(dstate, _to(device, u['prev']), env, ao)
  loss, _ = lrn.compute_gradients(unroll)
  p = nets_torch.to_torch(ref, requires_grad=True)
  t = lambda a: torch.tensor(a)
  logits, baseline, new_state = nets_torch.mlp_lstm_unroll(p, len(mlp), len(lstm), t(obs), t(u['done']),
                                                           [(t(h), t(c)) for h, c in state])
  total, _ = nets_torch.impala_loss_torch(logits, baseline, t(u['beh']), t(u['actions']), t(u['reward']), t(u['done']))
  total.backward()
  head, _, ldh = agent.head_buffers()
  head = head.cpu().numpy().reshape(T1, B, ldh)
  assert np.max(np.abs(head[..., :A] - logits.detach().numpy())) < 2e-4
  # carried state of every cell
  _, st = agent(unroll.prev_actions, env, dstate, unroll=True, is_training=True)
  for (h, c), (hr, cr) in zip(st, new_state):
    assert np.max(np.abs(h.cpu().numpy() - hr.detach().numpy())) < 2e-4
    assert np.max(np.abs(c.cpu().numpy() - cr.detach().numpy())) < 2e-4
  lrn.compute_gradients(unroll)
  _check(agent, lrn, p, total, loss, lr, lambda n: 1e-3)
  # single step from the initial state == first step of an unroll from the initial state
  init = agent.initial_state(B)
  env1 = utils.EnvOutput(env.reward[0], env.done[0], env.observation[0], None, None)
  out1, st1 = agent(unroll.prev_actions[0], env1, init)
  l1 = out1.policy_logits.clone()
  outT, _ = agent(unroll.prev_actions, env, init, unroll=True, is_training=True)
  assert torch.allclose(l1, outT.policy_logits[0], atol=1e-5) and len(st1) == len(lstm)


def test_tf_checkpoint_round_trip_through_agent(device, tmp_path):
  """save_agent writes an ImpalaDeep + Adam state under the reference's tf.train.Checkpoint keys
  (agents/vtrace/learner.py:286-296); restore_agent into a fresh agent + optimizer continues bit-identically."""
  from seed_rl_amd import learner, networks, optimizers, smoke_step, tf_checkpoint as tc, parametric_distribution as pd
  A, obs = 9, (24, 32, 3)

  def make(seed):
    agent = networks.ImpalaDeep(A, observation_shape=obs, device=device, seed=seed)
    opt = optimizers.Adam(optimizers.PolynomialDecay(1e-3, 100), beta_1=0.0, epsilon=3.125e-7)
    return agent, opt, learner.Learner(agent, opt, pd.categorical_distribution(A))
  agent, opt, lrn = make(0)
  unroll = smoke_step.make_deep_unroll(agent, 5, 4, A, device, seed=3)
  lrn.minimize(unroll); lrn.minimize(unroll)
  prefix = str(tmp_path / 'ckpt-2')
  keys = tc.save_agent(prefix, agent, optimizer=opt)
  assert 'agent/_stacks/0/_conv/kernel/.ATTRIBUTES/VARIABLE_VALUE' in keys
  assert 'agent/_core/recurrent_kernel/.OPTIMIZER_SLOT/optimizer/v/.ATTRIBUTES/VARIABLE_VALUE' in keys
  assert 'optimizer/iter/.ATTRIBUTES/VARIABLE_VALUE' in keys and 'agent/entropy_cost_param/.ATTRIBUTES/VARIABLE_VALUE' in keys
  agent2, opt2, lrn2 = make(123)
  tc.restore_agent(prefix, agent2, optimizer=opt2)
  assert opt2.iterations == 2
  l1, _ = lrn.minimize(unroll)
  l2, _ = lrn2.minimize(unroll)
  assert float(l1) == float(l2) and torch.equal(agent.flat.params, agent2.flat.params)
