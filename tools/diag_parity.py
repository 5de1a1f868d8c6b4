#!/usr/bin/env python
"""Per-tensor gradient distances: HIP vs fp64 oracle and fp32 oracle vs fp64 oracle (diagnostic; GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from oracle import nets_torch
from tests import synth, parity


def main():
  B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
  T1, A = 21, 18
  dev = torch.device('cuda:0')
  from seed_rl_amd import learner, networks, optimizers, utils, parametric_distribution as pd
  u = synth.atari_unroll(3, T1, B, A)
  agent = networks.AtariShallow(A, device=dev, seed=5)
  ref_params = nets_torch.init_params(nets_torch.param_spec('atari_shallow', A), seed=5)
  opt = optimizers.Adam(1e-3)
  lrn = learner.Learner(agent, opt, pd.categorical_distribution(A))
  t_ = lambda a: torch.as_tensor(np.ascontiguousarray(a)).to(dev)
  env = utils.EnvOutput(t_(u['reward']), t_(u['done']), t_(u['frames']), None, None)
  ao = networks.AgentOutput(t_(u['actions']), t_(u['behaviour_logits']), t_(u['behaviour_baseline']))
  unroll = learner.Unroll(networks.AgentState((), t_(u['frame_state'])), t_(u['prev_actions']), env, ao)
  lrn.compute_gradients(unroll)
  grads = {n: g.cpu().numpy().astype(np.float64) for n, g in agent.reference_gradients().items()}
  t = lambda a: torch.tensor(a)
  res = {}
  import contextlib
  for name, dt in (('f32', torch.float32), ('f64', torch.float64)):
    ctx = nets_torch.float64_truth() if dt == torch.float64 else contextlib.nullcontext()
    with ctx:
      p = nets_torch.to_torch(ref_params, requires_grad=True, dtype=dt)
      logits, baseline, _, _ = nets_torch.atari_shallow_unroll(p, 'atari_shallow', A, t(u['prev_actions']), t(u['reward']),
                                                               t(u['done']), t(u['frames']), t(u['frame_state']))
      total, _ = nets_torch.impala_loss_torch(logits, baseline, t(u['behaviour_logits']), t(u['actions']), t(u['reward']),
                                              t(u['done']), entropy_cost=0.00025)
      total.backward()
      res[name] = {k: v.grad.double().numpy() for k, v in p.items()}
  for k in res['f64']:
    r = res['f64'][k]; den = max(np.abs(r).max(), 1e-3)
    eh = np.abs(grads[k] - r); eo = np.abs(res['f32'][k] - r)
    ih = np.unravel_index(eh.argmax(), eh.shape)
    print('%-24s max|g64| %.3e  HIP-vs-64 %.3e at %s (g64 there %.3e, hip %.3e)  oracle32-vs-64 %.3e   rms hip %.2e oracle %.2e' % (
        k, np.abs(r).max(), eh.max() / den, ih, r[ih], grads[k][ih], eo.max() / den,
        np.sqrt((eh ** 2).mean()) / den, np.sqrt((eo ** 2).mean()) / den))
  k = 'fc/kernel'
  e = np.abs(grads[k] - res['f64'][k])
  print('fc/kernel err by row-block (81 pixels x 32 ch): top rows', np.argsort(-e.max(1))[:10], 'top cols', np.argsort(-e.max(0))[:10])
  print('err quantiles', np.quantile(e, [0.5, 0.9, 0.99, 0.999, 1.0]))


if __name__ == '__main__' and not (len(sys.argv) > 2 and sys.argv[2] == 'flips'):
  main()


def relu_flips():
  """Hypothesis check: the sparse fc/kernel differences are ReLU-mask flips of fc pre-activations within fp32 rounding
  of zero (a discrete, legitimate difference like a max-pool argmax flip)."""
  B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
  T1, A = 21, 18
  dev = torch.device('cuda:0')
  from seed_rl_amd import networks, utils
  u = synth.atari_unroll(3, T1, B, A)
  agent = networks.AtariShallow(A, device=dev, seed=5)
  ref_params = nets_torch.init_params(nets_torch.param_spec('atari_shallow', A), seed=5)
  t_ = lambda a: torch.as_tensor(np.ascontiguousarray(a)).to(dev)
  env = utils.EnvOutput(t_(u['reward']), t_(u['done']), t_(u['frames']), None, None)
  agent(t_(u['prev_actions']), env, networks.AgentState((), t_(u['frame_state'])), unroll=True, is_training=True)
  hfc = agent._last['hfc'].cpu().numpy()
  t = lambda a: torch.tensor(a)
  out = {}
  import contextlib
  for name, dt in (('f32', torch.float32), ('f64', torch.float64)):
    ctx = nets_torch.float64_truth() if dt == torch.float64 else contextlib.nullcontext()
    with ctx, torch.no_grad():
      p = nets_torch.to_torch(ref_params, dtype=dt)
      stacked, _ = nets_torch.stack_frames_torch(t(u['frames']), t(u['frame_state']), t(u['done']), 4)
      x = (stacked / 255).reshape((T1 * B,) + stacked.shape[2:])
      for i, s in enumerate((4, 2)):
        x = torch.relu(nets_torch.conv2d(x, p['conv%d/kernel' % i], p['conv%d/bias' % i], s, 'valid'))
      z = x.reshape(x.shape[0], -1) @ p['fc/kernel'] + p['fc/bias']
      out[name] = z.double().numpy()
  z64 = out['f64']
  for name, act in (('HIP', hfc > 0), ('oracle32', out['f32'] > 0)):
    flips = np.argwhere(act != (z64 > 0))
    print(name, 'relu-mask flips vs fp64:', len(flips), 'cols', sorted(set(flips[:, 1].tolist()))[:20],
          '|z64| at flips max %.2e' % (np.abs(z64[act != (z64 > 0)]).max() if len(flips) else 0))
  print('max |hfc - relu(z64)| %.3e ; max |relu(z32) - relu(z64)| %.3e' % (
      np.abs(hfc - np.maximum(z64, 0)).max(), np.abs(np.maximum(out['f32'], 0) - np.maximum(z64, 0)).max()))


if __name__ == '__main__' and len(sys.argv) > 2 and sys.argv[2] == 'flips':
  relu_flips()
