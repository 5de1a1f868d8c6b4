"""One tiny learner step (forward + loss + backward + Adam) used by __graft_entry__.smoke()."""
import numpy as np
import torch


def make_unroll(agent, T1, B, A, device, seed=0, done_p=0.01):
  """Synthetic cfg2-shaped unroll resident in HBM."""
  from seed_rl_amd import learner, utils
  g = torch.Generator(device='cpu').manual_seed(seed)
  H, W = agent._obs
  frames = torch.randint(0, 256, (T1, B, H, W, 1), dtype=torch.uint8, generator=g).to(device)
  done = (torch.rand((T1, B), generator=g) < done_p).to(device)
  env = utils.EnvOutput(
      reward=torch.randn((T1, B), generator=g).to(device), done=done, observation=frames,
      abandoned=torch.zeros_like(done), episode_step=torch.ones((T1, B), dtype=torch.int32, device=device))
  from seed_rl_amd.networks import AgentOutput
  ao = AgentOutput(action=torch.randint(0, A, (T1, B), generator=g).to(device),
                   policy_logits=torch.randn((T1, B, A), generator=g).to(device),
                   baseline=torch.randn((T1, B), generator=g).to(device))
  prev = torch.randint(0, A, (T1, B), generator=g).to(device)
  return learner.Unroll(agent.initial_state(B), prev, env, ao)


def make_deep_unroll(agent, T1, B, A, device, seed=0, done_p=0.01):
  """Synthetic cfg3-shaped unroll (DMLab 72x96x3 uint8 frames, LSTM state) resident in HBM."""
  from seed_rl_amd import learner, utils
  from seed_rl_amd.networks import AgentOutput
  g = torch.Generator(device='cpu').manual_seed(seed)
  h, w, c = agent._obs
  frames = torch.randint(0, 256, (T1, B, h, w, c), dtype=torch.uint8, generator=g).to(device)
  done = (torch.rand((T1, B), generator=g) < done_p).to(device)
  env = utils.EnvOutput(
      reward=torch.randn((T1, B), generator=g).to(device), done=done, observation=frames,
      abandoned=torch.zeros_like(done), episode_step=torch.ones((T1, B), dtype=torch.int32, device=device))
  ao = AgentOutput(action=torch.randint(0, A, (T1, B), generator=g).to(device),
                   policy_logits=torch.randn((T1, B, A), generator=g).to(device),
                   baseline=torch.randn((T1, B), generator=g).to(device))
  prev = torch.randint(0, A, (T1, B), generator=g).to(device)
  st = tuple((0.1 * torch.randn((B, agent._H), generator=g)).to(device) for _ in range(2))
  return learner.Unroll(st, prev, env, ao)


def run(device):
  from seed_rl_amd import learner, networks, optimizers, parametric_distribution as pd
  A = 6
  agent = networks.AtariShallow(A, device=device, seed=0)
  opt = optimizers.Adam(optimizers.PolynomialDecay(4.8e-4, 1000), beta_1=0.0, epsilon=3.125e-7)
  lr = learner.Learner(agent, opt, pd.categorical_distribution(A))
  unroll = make_unroll(agent, 5, 4, A, device)
  p0 = agent.flat.params.clone()
  loss, _ = lr.minimize(unroll)
  torch.cuda.synchronize()
  assert np.isfinite(float(loss)), loss
  assert not torch.equal(p0, agent.flat.params)
  print('smoke: learner step ok, loss %.5f' % float(loss))
