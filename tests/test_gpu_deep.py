"""GPU parity of the ImpalaDeep path (dmlab/networks.py) -- max-pool, LSTM unroll with done-reset, and the
whole train step -- vs the torch-CPU fp32 oracle (oracle/nets_torch.py) on identical seeded inputs."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import nets_torch
from tests import synth

pytestmark = pytest.mark.gpu


def _to(device, a):
  return torch.as_tensor(np.ascontiguousarray(a)).to(device)


@pytest.mark.parametrize('n,ih,iw,c', [(2, 72, 96, 16), (3, 36, 48, 32), (2, 9, 12, 32), (2, 7, 5, 4), (1, 1, 1, 8), (4, 18, 24, 32),
                                        (1, 2, 2, 4), (3, 4, 6, 8), (2, 5, 8, 4)])
def test_maxpool_same_parity(device, n, ih, iw, c):
  """TF 'SAME' 3x3/2 max-pool (asymmetric padding for even sizes) forward: exact; backward: exact
  (gradient routed to the window argmax)."""
  from seed_rl_amd import ops
  rng = np.random.default_rng(n * 100 + ih)
  x = rng.normal(size=(n, ih, iw, c)).astype(np.float32)
  xt = torch.tensor(x, requires_grad=True)
  y = nets_torch.max_pool_3x3_s2_same(xt)
  dy = rng.normal(size=tuple(y.shape)).astype(np.float32)
  y.backward(torch.tensor(dy))
  oh, ow = (ih + 1) // 2, (iw + 1) // 2
  yd = torch.empty((n, oh, ow, c), device=device)
  arg = torch.empty((n, oh, ow, c), dtype=torch.uint8, device=device)
  ops.maxpool_fwd(_to(device, x), yd, arg)
  np.testing.assert_array_equal(yd.cpu().numpy(), y.detach().numpy())
  dx = torch.full((n, ih, iw, c), 7.0, device=device)
  ops.maxpool_bwd(_to(device, dy), arg, dx)
  np.testing.assert_allclose(dx.cpu().numpy(), xt.grad.numpy(), rtol=0, atol=1e-6)


def _deep_unroll(device, u):
  from seed_rl_amd import learner, networks, utils
  T1, B = u['done'].shape
  env = utils.EnvOutput(reward=_to(device, u['reward']), done=_to(device, u['done']),
                        observation=_to(device, u['frames']),
                        abandoned=torch.zeros((T1, B), dtype=torch.bool, device=device),
                        episode_step=torch.ones((T1, B), dtype=torch.int32, device=device))
  ao = networks.AgentOutput(_to(device, u['actions']), _to(device, u['behaviour_logits']),
                            _to(device, u['behaviour_baseline']))
  return learner.Unroll((_to(device, u['h0']), _to(device, u['c0'])), _to(device, u['prev_actions']), env, ao)


@pytest.mark.parametrize('T1,B,A,obs', [(5, 3, 9, (72, 96, 3)), (21, 2, 9, (72, 96, 3)), (4, 5, 6, (24, 32, 3))])
def test_impala_deep_train_step_parity(device, T1, B, A, obs):
  """ImpalaDeep unroll -> fused loss -> backward (BPTT through the LSTM with done-reset, residual stacks,
  max-pool) -> Adam vs the oracle graph.  Tolerances: logits/baseline 3e-4 abs; gradients 1e-3 of each tensor's
  max (15 conv layers + 21 recurrent steps of fp32 re-association) for the layers after the last max-pool and
  1e-2 for the layers upstream of a max-pool: a pool window whose two largest inputs differ by less than the
  fp32 re-association noise (~1e-7; O(1) such windows among the ~5e5 of a step) routes its gradient to the other
  pixel on one side of the comparison -- a discrete, legitimate difference; parameters after Adam 5e-5."""
  from seed_rl_amd import learner, networks, optimizers, parametric_distribution as pd
  u = synth.dmlab_unroll(7, T1, B, A, H=obs[0], W=obs[1], done_p=0.15)
  agent = networks.ImpalaDeep(A, observation_shape=obs, device=device, seed=3)
  assert len(agent.trainable_variables) == 39                       # tests/agents_test.py:45
  if obs == (72, 96, 3) and A == 9:
    assert agent.flat.num_params() - (agent._ldh - A - 1) * (agent._H + 1) - 1 == 1520714   # SURVEY 8(a) a4 (+1: entropy-cost slot)
  ref_params = nets_torch.init_params(nets_torch.param_spec('impala_deep', A, obs), seed=3)
  for (n, v) in agent.trainable_variables:
    np.testing.assert_array_equal(v.cpu().numpy(), ref_params[n])
  cfg = learner.LossConfig(lambda_=0.95, max_abs_reward=1.0)
  opt = optimizers.Adam(optimizers.PolynomialDecay(4.8e-4, 100), beta_1=0.0, epsilon=3.125e-7)
  lrn = learner.Learner(agent, opt, pd.categorical_distribution(A), config=cfg)
  unroll = _deep_unroll(device, u)
  loss, session = learner.compute_loss(None, lrn.dist, agent, *unroll, config=cfg, want_vtrace=True)
  agent.backward()
  head, _, ldh = agent.head_buffers()
  head = head.cpu().numpy().reshape(T1, B, ldh)

  p = nets_torch.to_torch(ref_params, requires_grad=True)
  logits, baseline, state = nets_torch.impala_deep_unroll(
      p, A, torch.tensor(u['prev_actions']), torch.tensor(u['reward']), torch.tensor(u['done']),
      torch.tensor(u['frames']), (torch.tensor(u['h0']), torch.tensor(u['c0'])))
  total, aux = nets_torch.impala_loss_torch(
      logits, baseline, torch.tensor(u['behaviour_logits']), torch.tensor(u['actions']),
      torch.tensor(u['reward']), torch.tensor(u['done']), entropy_cost=0.00025, lambda_=0.95, max_abs_reward=1.0)
  total.backward()
  assert np.max(np.abs(head[..., :A] - logits.detach().numpy())) < 3e-4
  assert np.max(np.abs(head[..., A] - baseline.detach().numpy())) < 3e-4
  assert abs(float(loss) - float(total.detach())) < 2e-4 * max(1.0, abs(float(total.detach())))
  # final LSTM state handed to the next unroll
  _, st = agent(unroll.prev_actions, unroll.env_outputs, unroll.agent_state, unroll=True, is_training=True)
  assert np.max(np.abs(st[0].cpu().numpy() - state[0].detach().numpy())) < 2e-4
  assert np.max(np.abs(st[1].cpu().numpy() - state[1].detach().numpy())) < 2e-4
  grads = agent.reference_gradients()
  for n, t in p.items():
    g, r = grads[n].cpu().numpy(), t.grad.numpy()
    upstream_of_pool = n.startswith('stack0/') or n.startswith('stack1/') or n.startswith('stack2/conv/')
    tol = 1e-2 if upstream_of_pool else 1e-3
    assert np.max(np.abs(g - r)) <= tol * max(np.abs(r).max(), 1e-3), n
  lrn.apply_gradients()
  kopt = nets_torch.KerasAdam(list(p.values()), nets_torch.polynomial_decay(4.8e-4, 100), beta_1=0.0,
                              epsilon=3.125e-7)
  kopt.apply_gradients([t.grad for t in p.values()])
  for (n, v), t in zip(agent.trainable_variables, p.values()):
    # Adam(beta_1=0) normalises each gradient element to ~lr: an argmax flip moves single elements by <= 2 lr
    assert np.max(np.abs(v.cpu().numpy() - t.detach().numpy())) < 5e-5 + 2 * 4.8e-4 * (n.startswith('stack')), n


def test_lstm_input_projection_on_padded_kernel(device, monkeypatch):
  """The LSTM input projection of a training-sized batch (>= 4096 rows) runs as [N, ldx] x [ldx, 4H] on a zero-padded
  copy of its kernel (in_dim = 256 + 1 + 9 = 266 is not a multiple of 4, which would keep it off the bf16x6 GEMMs):
  outputs, final state and every gradient equal the un-padded path's (SEEDHIP_LSTM_PADK=0) to fp32 rounding, and the
  layer reports the bf16 pipe."""
  from seed_rl_amd import learner, networks, ops, parametric_distribution as pd
  T1, B, A, obs = 16, 256, 9, (24, 32, 3)
  u = synth.dmlab_unroll(11, T1, B, A, H=obs[0], W=obs[1], done_p=0.1)
  unroll = _deep_unroll(device, u)
  cfg = learner.LossConfig(lambda_=0.95, max_abs_reward=1.0)
  res = {}
  for mode in ('1', '0'):
    monkeypatch.setenv('SEEDHIP_LSTM_PADK', mode)
    agent = networks.ImpalaDeep(A, observation_shape=obs, device=device, seed=3)
    loss, _ = learner.compute_loss(None, pd.categorical_distribution(A), agent, *unroll, config=cfg, want_vtrace=True)
    agent.backward()
    head, _, ldh = agent.head_buffers()
    grads = {n: t.cpu().numpy().copy() for n, t in agent.reference_gradients().items()}
    res[mode] = (float(loss), head.cpu().numpy().copy(), grads, agent._last_lstm['padded'], agent._last_lstm['gx'])
  assert res['1'][3] and not res['0'][3]
  assert ops.conv2d_pipe(res['1'][4], 0) == 6
  assert abs(res['1'][0] - res['0'][0]) <= 1e-5 * max(1.0, abs(res['0'][0]))
  assert np.max(np.abs(res['1'][1] - res['0'][1])) <= 2e-5
  for n, g0 in res['0'][2].items():
    g1 = res['1'][2][n]
    assert np.max(np.abs(g1 - g0)) <= 1e-4 * max(np.abs(g0).max(), 1e-3), n


def test_impala_deep_byte_masks_and_fused_pool_backward_change_no_bit(device, monkeypatch):
  """At 512 images every ImpalaDeep layer runs on its training-size kernel.  The ReLU masks as bytes (written by the
  producing epilogues, read by the masked data gradients) and the max-pool backward inside the entry convolution's data
  gradient are re-arrangements of the same arithmetic: with either switched off (networks._RELU_BITS,
  SEEDHIP_POOL_DGRAD=0) the loss and EVERY gradient are bit-identical."""
  from seed_rl_amd import learner, networks, ops, parametric_distribution as pd
  T1, B, A, obs = 2, 256, 9, (72, 96, 3)
  u = synth.dmlab_unroll(5, T1, B, A, H=obs[0], W=obs[1], done_p=0.1)
  unroll = _deep_unroll(device, u)
  cfg = learner.LossConfig(lambda_=0.95, max_abs_reward=1.0)
  g = ops.conv_geom(T1 * B, 36, 48, 16, 3, 3, 1, 'same', 16)
  assert ops.conv2d_fwd_outbits_supported(g) and ops.conv2d_bwd_data_pool_supported(ops.conv_geom(T1 * B, 36, 48, 16, 3, 3, 1, 'same', 32))

  def run():
    agent = networks.ImpalaDeep(A, observation_shape=obs, device=device, seed=3)
    loss, _ = learner.compute_loss(None, pd.categorical_distribution(A), agent, *unroll, config=cfg, want_vtrace=True)
    agent.backward()
    used_bits = agent._last['saved'][1]['blocks'][0][2] is not None
    return float(loss), {n: t.clone() for n, t in agent.reference_gradients().items()}, used_bits

  base = run()
  assert base[2]
  monkeypatch.setattr(networks, '_RELU_BITS', False)
  plain_masks = run()
  assert not plain_masks[2]
  monkeypatch.setattr(networks, '_RELU_BITS', True)
  monkeypatch.setenv('SEEDHIP_POOL_DGRAD', '0')
  two_kernels = run()
  for other in (plain_masks, two_kernels):
    assert other[0] == base[0]
    for n, t in base[1].items():
      assert torch.equal(t, other[1][n]), n


def test_impala_deep_single_step_inference(device):
  """unroll=False (central inference, learner.py:386-390) == first step of the unroll; state carried."""
  from seed_rl_amd import networks, utils
  A = 9
  u = synth.dmlab_unroll(2, 3, 4, A, H=24, W=32)
  agent = networks.ImpalaDeep(A, observation_shape=(24, 32, 3), device=device, seed=1)
  st = (_to(device, u['h0']), _to(device, u['c0']))
  env1 = utils.EnvOutput(_to(device, u['reward'][0]), _to(device, u['done'][0]), _to(device, u['frames'][0]), None, None)
  out1, st1 = agent(_to(device, u['prev_actions'][0]), env1, st)
  l1, b1 = out1.policy_logits.clone(), out1.baseline.clone()
  assert out1.action.shape == (4,) and int(out1.action.max()) < A
  envT = utils.EnvOutput(_to(device, u['reward']), _to(device, u['done']), _to(device, u['frames']), None, None)
  outT, _ = agent(_to(device, u['prev_actions']), envT, st, unroll=True, is_training=True)
  assert torch.allclose(l1, outT.policy_logits[0], atol=1e-5) and torch.allclose(b1, outT.baseline[0], atol=1e-5)
  # second single step from the carried state == second unroll step
  env2 = utils.EnvOutput(_to(device, u['reward'][1]), _to(device, u['done'][1]), _to(device, u['frames'][1]), None, None)
  out2, _ = agent(_to(device, u['prev_actions'][1]), env2, st1)
  assert torch.allclose(out2.policy_logits, outT.policy_logits[1], atol=1e-5)
