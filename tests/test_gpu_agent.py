"""GPU parity of the whole learner step (agent unroll -> loss -> backward -> Adam) vs the
torch-CPU fp32 oracle of the same graph, on identical seeded synthetic unrolls."""
import numpy as np
import pytest
import torch

from oracle import nets_torch
from tests import synth

pytestmark = pytest.mark.gpu


def _to(device, a):
  return torch.as_tensor(np.ascontiguousarray(a)).to(device)


def _unroll(device, agent, u):
  from seed_rl_amd import learner, networks, utils
  T1, B = u['done'].shape
  env = utils.EnvOutput(reward=_to(device, u['reward']), done=_to(device, u['done']),
                        observation=_to(device, u['frames']),
                        abandoned=torch.zeros((T1, B), dtype=torch.bool, device=device),
                        episode_step=torch.ones((T1, B), dtype=torch.int32, device=device))
  ao = networks.AgentOutput(_to(device, u['actions']), _to(device, u['behaviour_logits']),
                            _to(device, u['behaviour_baseline']))
  st = networks.AgentState((), _to(device, u['frame_state']))
  return learner.Unroll(st, _to(device, u['prev_actions']), env, ao)


@pytest.mark.parametrize('torso,T1,B,A', [('shallow', 6, 5, 6), ('shallow', 21, 8, 18), ('dqn', 4, 3, 18)])
def test_atari_shallow_train_step_parity(device, torso, T1, B, A):
  """Tolerances (fp32, different accumulation order on MFMA vs oneDNN):
  logits/baseline 2e-4 abs; gradients 3e-4 of each tensor's max; vs/pg_adv 1e-4
  (network outputs feed V-trace here; the <=1e-5 V-trace bar on identical inputs is
  tests/test_gpu_kernels.py); parameters after one Adam step 1e-6 abs."""
  from seed_rl_amd import learner, networks, optimizers, parametric_distribution as pd
  kind = 'atari_shallow' if torso == 'shallow' else 'atari_dqn_body'
  u = synth.atari_unroll(3, T1, B, A, done_p=0.1, zero_state=False)
  agent = networks.AtariShallow(A, torso=torso, device=device, seed=5)
  ref_params = nets_torch.init_params(nets_torch.param_spec(kind, A), seed=5)
  # same initialiser on both sides
  for (n, v) in agent.trainable_variables:
    np.testing.assert_array_equal(v.cpu().numpy(), ref_params[n])

  cfg = learner.LossConfig(lambda_=0.95, kl_cost=0.05, max_abs_reward=1.0)
  opt = optimizers.Adam(optimizers.PolynomialDecay(4.8e-4, 100), beta_1=0.0, epsilon=3.125e-7)
  lrn = learner.Learner(agent, opt, pd.categorical_distribution(A), config=cfg)
  unroll = _unroll(device, agent, u)
  loss, session = learner.compute_loss(None, lrn.dist, agent, *unroll, config=cfg, want_vtrace=True)
  agent.backward()
  head, _, ldh = agent.head_buffers()
  head = head.cpu().numpy().reshape(T1, B, ldh)

  # ---- oracle ----
  p = nets_torch.to_torch(ref_params, requires_grad=True)
  logits, baseline, new_fs, _ = nets_torch.atari_shallow_unroll(
      p, kind, A, torch.tensor(u['prev_actions']), torch.tensor(u['reward']), torch.tensor(u['done']),
      torch.tensor(u['frames']), torch.tensor(u['frame_state']))
  total, aux = nets_torch.impala_loss_torch(
      logits, baseline, torch.tensor(u['behaviour_logits']), torch.tensor(u['actions']),
      torch.tensor(u['reward']), torch.tensor(u['done']), entropy_cost=0.00025,
      lambda_=0.95, kl_cost=0.05, max_abs_reward=1.0)
  total.backward()

  assert np.max(np.abs(head[..., :A] - logits.detach().numpy())) < 2e-4
  assert np.max(np.abs(head[..., A] - baseline.detach().numpy())) < 2e-4
  assert abs(float(loss) - float(total.detach())) < 1e-4 * max(1.0, abs(float(total.detach())))
  assert np.max(np.abs(session['vtrace/vs'].cpu().numpy() - aux['vs'].numpy())) < 1e-4
  assert np.max(np.abs(session['vtrace/pg_advantages'].cpu().numpy() - aux['pg_advantages'].numpy())) < 1e-4
  grads = agent.reference_gradients()
  for n, t in p.items():
    g, r = grads[n].cpu().numpy(), t.grad.numpy()
    assert np.max(np.abs(g - r)) <= 3e-4 * max(np.abs(r).max(), 1e-3), n

  # frame-stacking state handed to the next unroll: bit exact
  _, st = agent(unroll.prev_actions, unroll.env_outputs, unroll.agent_state, unroll=True, is_training=True)
  np.testing.assert_array_equal(st.frame_stacking_state.cpu().numpy(), new_fs.numpy())

  # ---- one optimizer step ----
  lrn.apply_gradients()
  kopt = nets_torch.KerasAdam(list(p.values()), nets_torch.polynomial_decay(4.8e-4, 100), beta_1=0.0,
                              epsilon=3.125e-7)
  kopt.apply_gradients([t.grad for t in p.values()])
  for (n, v), t in zip(agent.trainable_variables, p.values()):
    assert np.max(np.abs(v.cpu().numpy() - t.detach().numpy())) < 5e-5, n


def test_atari_shallow_single_step_inference(device):
  """unroll=False path (central inference, learner.py:386-390): one step == first step of an unroll."""
  from seed_rl_amd import networks, utils
  A = 6
  u = synth.atari_unroll(4, 3, 4, A, zero_state=False)
  agent = networks.AtariShallow(A, device=device, seed=1)
  st = networks.AgentState((), _to(device, u['frame_state']))
  env1 = utils.EnvOutput(_to(device, u['reward'][0]), _to(device, u['done'][0]), _to(device, u['frames'][0]),
                         None, None)
  out1, st1 = agent(_to(device, u['prev_actions'][0]), env1, st)
  l1 = out1.policy_logits.clone(); b1 = out1.baseline.clone()
  assert out1.action.shape == (4,) and int(out1.action.max()) < A
  envT = utils.EnvOutput(_to(device, u['reward']), _to(device, u['done']), _to(device, u['frames']), None, None)
  outT, _ = agent(_to(device, u['prev_actions']), envT, st, unroll=True, is_training=True)
  assert torch.allclose(l1, outT.policy_logits[0], atol=1e-6) and torch.allclose(b1, outT.baseline[0], atol=1e-6)
  # state after one step == oracle stack_frames state
  from oracle import frames_np
  _, ns = frames_np.stack_frames(u['frames'][:1], u['frame_state'], u['done'][:1], 4)
  np.testing.assert_array_equal(st1.frame_stacking_state.cpu().numpy(), ns)


def test_checkpoint_roundtrip(device, tmp_path):
  """seed_rl_amd.checkpoint: save after two train steps, restore into a FRESH agent + optimizer, and the third step
  of both runs is bitwise identical (parameters, Adam moments and step counter all carried; arrays are stored under
  the reference's variable names in Keras layouts)."""
  from seed_rl_amd import checkpoint, learner, networks, optimizers, smoke_step
  from seed_rl_amd import parametric_distribution as pd
  A, T, B = 6, 5, 8

  def make(seed):
    agent = networks.AtariShallow(A, device=device, seed=seed)
    opt = optimizers.Adam(optimizers.PolynomialDecay(1e-3, 100), beta_1=0.0, epsilon=3.125e-7)
    return agent, opt, learner.Learner(agent, opt, pd.categorical_distribution(A))

  agent, opt, lrn = make(0)
  unroll = smoke_step.make_unroll(agent, T + 1, B, A, device, seed=3)
  lrn.minimize(unroll); lrn.minimize(unroll)
  path = str(tmp_path / 'ckpt')          # no extension: save and restore agree on '.npz' 
  names = checkpoint.save(path, agent, opt)
  assert 'agent/policy_logits/kernel' in names and 'adam_v/baseline/bias' in names and 'iterations' in names
  agent2, opt2, lrn2 = make(123)                          # different init: everything must come from the file
  checkpoint.restore(path, agent2, opt2)
  assert opt2.iterations == 2
  l1, _ = lrn.minimize(unroll)
  l2, _ = lrn2.minimize(unroll)
  assert float(l1) == float(l2)
  assert torch.equal(agent.flat.params, agent2.flat.params)
