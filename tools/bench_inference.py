#!/usr/bin/env python
"""Central-inference micro-benchmark (learner.py:350-405 on the device store): env steps per second through
`InferenceState.inference` for one inference batch size, including run-id bookkeeping, the single-step agent
forward, the store append and the hand-over of completed unrolls.

  python tools/bench_inference.py [--agent atari|deep] [--n 64] [--envs 512] [--unroll 20] [--calls 200]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from seed_rl_amd import inference, networks, utils
from seed_rl_amd.unroll_store import Spec


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--agent', default='atari', choices=['atari', 'deep'])
  ap.add_argument('--n', type=int, default=64, help='inference batch size')
  ap.add_argument('--envs', type=int, default=512)
  ap.add_argument('--unroll', type=int, default=20)
  ap.add_argument('--calls', type=int, default=200)
  ap.add_argument('--mode', default='all', choices=['all', 'reference', 'fused', 'graph', 'replay', 'packed'])
  a = ap.parse_args()
  dev = torch.device('cuda')
  A = 18 if a.agent == 'atari' else 9
  obs = (84, 84, 1) if a.agent == 'atari' else (72, 96, 3)
  agent = networks.AtariShallow(A, device=dev) if a.agent == 'atari' else networks.ImpalaDeep(A, device=dev)
  env_specs = utils.EnvOutput(Spec((), torch.float32), Spec((), torch.bool), Spec(obs, torch.uint8),
                              Spec((), torch.bool), Spec((), torch.int32))
  ao_specs = networks.AgentOutput(Spec((), torch.int64), Spec((A,), torch.float32), Spec((), torch.float32))
  g = torch.Generator(device='cpu').manual_seed(0)
  groups = [torch.arange(i, i + a.n, dtype=torch.int32) for i in range(0, a.envs, a.n)]
  envs = []
  for ids in groups:
    envs.append(utils.EnvOutput(
        reward=torch.randn(a.n, generator=g).to(dev), done=(torch.rand(a.n, generator=g) < 0.01).to(dev),
        observation=torch.randint(0, 256, (a.n,) + obs, dtype=torch.uint8, generator=g).to(dev),
        abandoned=torch.zeros(a.n, dtype=torch.bool, device=dev), episode_step=torch.zeros(a.n, dtype=torch.int32, device=dev)))
  run_ids = [torch.full((a.n,), 7, dtype=torch.int64, device=dev) for _ in groups]
  groups = [x.to(dev) for x in groups]

  def bench(name, fn, between=None):
    def call(i):
      k = i % len(groups)
      return fn(groups[k], run_ids[k], envs[k], envs[k].reward)
    for i in range(3 * len(groups)):
      call(i)
      if between:
        between(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.calls):
      call(i)
      if between:
        between(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print('%-10s %s agent, inference batch %d over %d envs: %8.1f us per call, %9.0f env-steps/s'
          % (name, a.agent, a.n, a.envs, dt / a.calls * 1e6, a.calls * a.n / dt))

  if a.mode in ('all', 'reference'):
    st = inference.InferenceState(agent, a.envs, a.unroll, env_specs, ao_specs, Spec((), torch.int64), device=dev)
    bench('reference', st.inference)
  if a.mode in ('all', 'fused', 'graph', 'replay', 'packed'):
    cap = 2 * a.envs
    fused = inference.FusedInferenceState(agent, a.envs, a.unroll, env_specs, ao_specs, batch_capacity=cap, device=dev)
    calls_per_round = len(groups) * (a.unroll + 1)

    def drain(i):                      # hand the filled training batch over before it can overflow (no host read)
      if (i + 1) % calls_per_round == 0:
        fused.batch_count.zero_()
    if a.mode in ('all', 'fused'):
      bench('fused', fused.inference, drain)
    if a.mode in ('all', 'graph'):
      bench('hip-graph', fused.graphed(a.n, obs), drain)
    if a.mode in ('all', 'replay'):
      # the transport layer writes requests straight into the graph's static input buffers: replay only
      gfn = fused.graphed(a.n, obs)
      gfn.static_inputs['ids'].copy_(groups[0]); gfn.static_inputs['runs'].copy_(run_ids[0])
      state = {'i': 0}

      def replay_only(ids, runs, env, raw):
        gfn.static_inputs['ids'].copy_(ids)            # 256 B: which envs this batch holds
        gfn.graph.replay()
      bench('replay', replay_only, drain)
    if a.mode in ('all', 'packed'):
      # what bench.py's inference record times: two copies (packed request scalars, frames) + one graph replay
      import numpy as np
      gfn = fused.graphed(a.n, obs)
      packed = [torch.from_numpy(inference.pack_request(
          a.n, g_.cpu().numpy(), np.full((a.n,), 7, np.int64), e.reward.cpu().numpy(), e.reward.cpu().numpy(),
          e.done.cpu().numpy())).to(dev) for g_, e in zip(groups, envs)]
      idx = {id(g_): k for k, g_ in enumerate(groups)}
      bench('packed', lambda ids, runs, env, raw: gfn.replay_packed(packed[idx[id(ids)]], env.observation), drain)
    fused.check_errors()


if __name__ == '__main__':
  main()
