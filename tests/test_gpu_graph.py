"""HIP-graph replay of the train step (learner.GraphedStep): bitwise the same parameters as eager launches."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _mk(device, kind):
  from seed_rl_amd import learner, networks, optimizers, parametric_distribution as pd, smoke_step
  A = 6
  opt = optimizers.Adam(optimizers.PolynomialDecay(4.8e-4, 50), beta_1=0.0, epsilon=3.125e-7, capturable=True)
  if kind == 'atari':
    agent = networks.AtariShallow(A, device=device, seed=0)
    unroll = smoke_step.make_unroll(agent, 5, 4, A, device, seed=3, done_p=0.2)
  else:
    # 'deep_long': an unroll long / wide enough that the whole-unroll LSTM kernels' exchange buffers are megabytes
    # (replayed hipMemsetAsync nodes of that size corrupted memory on ROCm 7.2; they are armed by a kernel now)
    T1, B = (21, 32) if kind == 'deep_long' else (5, 4)
    agent = networks.ImpalaDeep(A, observation_shape=(24, 32, 3), device=device, seed=0)
    unroll = smoke_step.make_deep_unroll(agent, T1, B, A, device, seed=3, done_p=0.2)
  return learner.Learner(agent, opt, pd.categorical_distribution(A)), unroll


@pytest.mark.parametrize('kind', ['atari', 'deep', 'deep_long'])
def test_graphed_step_matches_eager(device, kind):
  """The warm-up steps GraphedStep runs before capturing leave NO trace: parameters, Adam moments and the
  step counter are put back, so replay k is bitwise eager step k."""
  from seed_rl_amd import learner
  eager, unroll = _mk(device, kind)
  losses = []
  for _ in range(3):
    l, _ = eager.minimize(unroll)
    losses.append(float(l))
  graphed, unroll2 = _mk(device, kind)
  p0 = graphed.agent.flat.params.clone()
  step = learner.GraphedStep(graphed, unroll2, warmup=2)
  assert graphed.optimizer.iterations == 0
  assert torch.equal(graphed.agent.flat.params, p0)
  sd = graphed.optimizer.state_dict()
  assert float(sd['m'].abs().max()) == 0.0 and float(sd['v'].abs().max()) == 0.0
  glosses = []
  for _ in range(3):
    out = step()
    torch.cuda.synchronize()
    glosses.append(float(out[0]))
  assert graphed.optimizer.iterations == 3
  assert glosses == losses
  assert torch.equal(graphed.agent.flat.params, eager.agent.flat.params)
  step.check_errors()
