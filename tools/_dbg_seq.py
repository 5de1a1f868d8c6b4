import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from seed_rl_amd import learner, networks, ops, optimizers, parametric_distribution as pd, smoke_step
dev = torch.device('cuda:0')
T1, B, A = 21, 32, 9
opt = optimizers.Adam(4.8e-4, beta_1=0.0, epsilon=3.125e-7, capturable=True)
agent = networks.ImpalaDeep(A, device=dev, seed=0)
unroll = smoke_step.make_deep_unroll(agent, T1, B, A, dev, seed=1000)
lrn = learner.Learner(agent, opt, pd.categorical_distribution(A))
for _ in range(2): lrn.minimize(unroll)
torch.cuda.synchronize()
use_graph = int(os.environ.get('USE_GRAPH', '1'))
if use_graph:
  gs = learner.GraphedStep(lrn, unroll, warmup=1)
  step = lambda: gs()
else:
  step = lambda: lrn.minimize(unroll)
H = 256
for i in range(int(os.environ.get('N', '40'))):
  t0 = time.perf_counter()
  out = step()
  torch.cuda.synchronize()
  dt = (time.perf_counter() - t0) * 1e3
  dz = agent._buf('lstm_dz', (T1, B, 4 * H))
  fb = int(agent._buf('lstm_seq_sync_bwd', (2,), torch.int32)[1]); ff = int(agent._buf('lstm_seq_sync', (2,), torch.int32)[1])
  nan_dz = int(torch.isnan(dz).sum()); nan_p = int(torch.isnan(agent.flat.params).sum())
  bad_t = [t for t in range(T1) if bool(torch.isnan(dz[t]).any())]
  print('iter %d %.2f ms loss %s flags fwd %d bwd %d nan(dz) %d steps %s nan(params) %d' % (i, dt, float(out[0]), ff, fb, nan_dz, bad_t[:6], nan_p))
  if nan_dz or nan_p or fb or ff: break
