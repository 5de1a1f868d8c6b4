"""GPU parity of the R2D2 path (config 5): dueling head, n-step double-Q loss kernel (incl. the reference's
known-answer n-step cases), DuelingLSTMDQNNet train step with burn-in vs the torch-CPU oracle."""
import numpy as np
import pytest
import torch

from oracle import nets_torch, r2d2_np
from tests import synth

pytestmark = pytest.mark.gpu


def _to(device, a):
  return torch.as_tensor(np.ascontiguousarray(a)).to(device)


def _run_loss(device, tq, gq, act, rew, done, iw=None, gamma=0.997, n=5):
  from seed_rl_amd import ops
  T, B, A = tq.shape
  loss_b = torch.empty(B, device=device); prio = torch.empty(B, device=device)
  total = torch.empty(1, device=device); dq = torch.full((T, B, A), 7.0, device=device)
  ws = torch.empty(ops.r2d2_loss_workspace_bytes(T, B, n) // 4 + 4, device=device)
  ops.r2d2_loss_fwd_bwd(_to(device, tq), _to(device, gq), _to(device, act.astype(np.int32)), _to(device, rew),
                        _to(device, done.astype(np.uint8)), None if iw is None else _to(device, iw), T, B, A, gamma, n,
                        0.9, 1e-3, B, loss_b, prio, dq, total, ws)
  return loss_b.cpu().numpy(), prio.cpu().numpy(), float(total[0]), dq.cpu().numpy()


@pytest.mark.parametrize('T,B,A,n', [(10, 32, 6, 5), (81, 37, 18, 5), (3, 2, 4, 1), (6, 5, 3, 3)])
def test_r2d2_loss_kernel_parity(device, T, B, A, n):
  """vs oracle/r2d2_np.py (pinned by agents/r2d2/learner_test.py:114-198).  fp32, same op order: loss /
  priorities rtol 2e-5 (h^-1 is numerically touchy: the reference's own test uses atol 2e-4), gradient 2e-5."""
  rng = np.random.default_rng(T * 100 + B)
  tq = rng.uniform(0, 1, (T, B, A)).astype(np.float32)
  gq = (rng.uniform(0, 1, (T, B, A)) * 3).astype(np.float32)
  act = rng.integers(0, A, (T, B))
  rew = rng.normal(size=(T, B)).astype(np.float32)
  done = rng.uniform(size=(T, B)) < 0.1
  iw = rng.uniform(0.1, 1, B).astype(np.float32)
  loss, prio, d_q = r2d2_np.loss_and_priorities(tq, gq, rew, done, act, 0.997, n)
  gl, gp, gt, gdq = _run_loss(device, tq, gq, act, rew, done, iw, n=n)
  np.testing.assert_allclose(gl, loss, rtol=2e-5, atol=1e-6)
  np.testing.assert_allclose(gp, prio, rtol=2e-5, atol=1e-6)
  assert abs(gt - float(np.mean(loss * iw))) <= 2e-5 * max(1.0, abs(float(np.mean(loss * iw))))
  np.testing.assert_allclose(gdq, d_q * (iw / B)[None, :, None], rtol=2e-5, atol=1e-7)


def test_r2d2_nstep_reference_known_answers(device):
  """agents/r2d2/learner_test.py:142-198 through the kernel: with q-values chosen so that
  h^-1(Q_target(s, argmax)) equals the test's q_target, the per-sequence loss reproduces the expected targets."""
  gamma, eps = 0.9, 1e-3
  # learner_test.py:165-183 (n_steps = 2, one done): rewards 1..6 (T=6... shifted), expected bellman targets
  rewards = np.array([[1.], [2.], [3.], [4.], [5.], [6.]], np.float32)
  done = np.array([[False], [False], [True], [False], [False], [False]])
  q_target = np.array([[10.], [20.], [30.], [40.], [50.], [60.]], np.float32)
  expect = r2d2_np.n_step_bellman_target(rewards, done, q_target, gamma, 2)
  # single action => replay_q = training_q, argmax = 0; target_q = h(q_target) so that h^-1 gives q_target back
  tq = np.zeros((6, 1, 1), np.float32)
  gq = r2d2_np.value_function_rescaling(q_target, eps)[..., None]
  loss, prio, total, dq = _run_loss(device, tq, gq, np.zeros((6, 1), np.int64), rewards, done, gamma=gamma, n=2)
  td = r2d2_np.value_function_rescaling(expect[1:], eps)[:, 0]          # replay_q = 0
  assert abs(loss[0] - 0.5 * float(np.sum(td * td))) <= 2e-4 * 0.5 * float(np.sum(td * td))


def test_dueling_head(device):
  from seed_rl_amd import ops
  rng = np.random.default_rng(0)
  N, A, ld = 50, 18, 20
  va = rng.normal(size=(N, ld)).astype(np.float32)
  adv, v = va[:, :A], va[:, A:A + 1]
  q_ref = v + (adv - adv.mean(-1, keepdims=True, dtype=np.float32))
  q = torch.empty((N, A), device=device); act = torch.empty(N, dtype=torch.int32, device=device)
  ops.dueling_fwd(_to(device, va), ld, N, A, q, act)
  np.testing.assert_allclose(q.cpu().numpy(), q_ref, rtol=0, atol=1e-6)
  np.testing.assert_array_equal(act.cpu().numpy(), q_ref.argmax(-1))
  dq = rng.normal(size=(N, A)).astype(np.float32)
  d_va = torch.full((N, ld), 7.0, device=device)
  ops.dueling_bwd(_to(device, dq), N, A, d_va, ld)
  ref = np.zeros((N, ld), np.float32)
  ref[:, :A] = dq - dq.mean(-1, keepdims=True)
  ref[:, A] = dq.sum(-1)
  np.testing.assert_allclose(d_va.cpu().numpy(), ref, rtol=0, atol=2e-6)


@pytest.mark.parametrize('T1,B,A,burn_in', [(9, 3, 6, 3), (13, 2, 18, 5)])
def test_r2d2_train_step_parity(device, T1, B, A, burn_in):
  """DuelingLSTMDQNNet x2 (training / target), burn-in without gradient, suffix unroll, n-step double-Q loss,
  global-norm clip 40, Adam(eps 1e-3) vs the oracle graph.  Tolerances: q-values 3e-4 abs, loss 2e-4 rel,
  priorities 1e-3 rel, gradients 1e-3 of each tensor's max."""
  from seed_rl_amd import networks, optimizers, r2d2_learner, utils
  u = synth.atari_unroll(5, T1, B, A, done_p=0.1, zero_state=False)
  rng = np.random.default_rng(1)
  h0 = (0.1 * rng.normal(size=(B, 512))).astype(np.float32); c0 = (0.1 * rng.normal(size=(B, 512))).astype(np.float32)
  iw = rng.uniform(0.2, 1.0, B).astype(np.float32)
  agent = networks.DuelingLSTMDQNNet(A, device=device, seed=2)
  target = networks.DuelingLSTMDQNNet(A, device=device, seed=9)
  ref = nets_torch.init_params(nets_torch.param_spec('r2d2', A), seed=2)
  ref_t = nets_torch.init_params(nets_torch.param_spec('r2d2', A), seed=9)
  assert set(n for n, _ in agent.trainable_variables) == set(ref)
  agent.load_reference_params(ref); target.load_reference_params(ref_t)
  cfg = r2d2_learner.R2D2Config(burn_in=burn_in, n_steps=3, update_target_every_n_step=0)
  opt = optimizers.Adam(4.8e-4, epsilon=1e-3)
  lrn = r2d2_learner.R2D2Learner(agent, target, opt, cfg)
  target.load_reference_params(ref_t)                      # undo the constructor's target <- training copy
  env = utils.EnvOutput(_to(device, u['reward']), _to(device, u['done']), _to(device, u['frames']), None, None)
  ao = networks.R2D2AgentOutput(_to(device, u['actions'].astype(np.int32)), None)
  st = networks.AgentState((_to(device, h0), _to(device, c0)), _to(device, u['frame_state']))
  unroll = r2d2_learner.Unroll(st, None, _to(device, u['prev_actions']), env, ao)
  loss_b, prio, total = r2d2_learner.compute_loss_and_priorities(
      agent, target, unroll.agent_state, unroll.prev_actions, unroll.env_outputs, unroll.agent_outputs,
      cfg.discounting, cfg.burn_in, cfg, _to(device, iw))
  q_gpu = agent._buf('q', ((T1 - burn_in) * B, A)).cpu().numpy().reshape(T1 - burn_in, B, A)
  agent.backward()

  # ---- oracle ----
  p = nets_torch.to_torch(ref, requires_grad=True)
  pt = nets_torch.to_torch(ref_t)
  t = lambda a: torch.tensor(a)
  def run(pp, lo, hi, fs, core):
    return nets_torch.r2d2_unroll(pp, A, t(u['prev_actions'][lo:hi]), t(u['reward'][lo:hi]), t(u['done'][lo:hi]),
                                  t(u['frames'][lo:hi]), fs, core)
  with torch.no_grad():
    _, fs1, core1 = run(p, 0, burn_in, t(u['frame_state']), (t(h0), t(c0)))
    _, fs1t, core1t = run(pt, 0, burn_in, t(u['frame_state']), (t(h0), t(c0)))
    out_t, _, _ = run(pt, burn_in, T1, fs1t, core1t)
  out, _, _ = run(p, burn_in, T1, fs1, tuple(x.detach() for x in core1))
  total_ref, loss_ref, prio_ref = nets_torch.r2d2_loss_torch(
      out.q_values, out_t.q_values, t(u['actions'][burn_in:]), t(u['reward'][burn_in:]), t(u['done'][burn_in:]),
      t(iw), cfg.discounting, cfg.n_steps)
  total_ref.backward()
  assert np.max(np.abs(q_gpu - out.q_values.detach().numpy())) < 3e-4
  assert abs(float(total) - float(total_ref)) <= 2e-4 * max(1.0, abs(float(total_ref)))
  np.testing.assert_allclose(prio.cpu().numpy(), prio_ref.detach().numpy(), rtol=1e-3, atol=1e-4)
  grads = agent.reference_gradients()
  for n, tt in p.items():
    g, r = grads[n].cpu().numpy(), tt.grad.numpy()
    assert np.max(np.abs(g - r)) <= 1e-3 * max(np.abs(r).max(), 1e-4), n
  # ---- minimize(): clip + Adam + priorities ----
  agent.load_reference_params(ref)
  tot2, prio2, gnorm = lrn.minimize(unroll, _to(device, iw))
  gn_ref = float(torch.sqrt(sum((tt.grad ** 2).sum() for tt in p.values())))
  assert abs(float(gnorm) - gn_ref) <= 1e-3 * gn_ref
  scale = min(1.0, cfg.clip_norm / gn_ref)
  kopt = nets_torch.KerasAdam(list(p.values()), lambda step: 4.8e-4, epsilon=1e-3)
  kopt.apply_gradients([tt.grad * scale for tt in p.values()])
  for (n, v), tt in zip(agent.trainable_variables, p.values()):
    assert np.max(np.abs(v.cpu().numpy() - tt.detach().numpy())) < 5e-5, n
