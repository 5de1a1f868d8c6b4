#!/usr/bin/env python
"""End-to-end serving throughput: synthetic actor processes -> gRPC (reference wire format) -> dynamic batching ->
central inference on the GPU -> completed unrolls -> train steps (seed_rl_amd/learner_server.py).

  python tools/bench_serving.py [--procs 8] [--envs-per-proc 16] [--n 64] [--batch 64] [--seconds 10]
Each actor process steps `envs-per-proc` synthetic Atari environments and sends them as ONE client-side batch per call
(the reference's env_batch_size: common/actor.py), so a server-side inference batch of n is filled by n / envs-per-proc
calls.  Prints env-steps/s served and train steps taken.  The transport is host Python (asyncio + protobuf): this
measures IT, not the GPU path (bench.py `inference`: 2.2 M env-steps/s at n = 256).
"""
import argparse
import multiprocessing as mp
import os
import sys
import tempfile
import time
import uuid

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def actor_proc(address, first_env, k, seconds, out):
  import collections
  from seed_rl_amd import grpc_service as gs                 # (no torch in the actor processes)
  EnvOutput = collections.namedtuple('EnvOutput', 'reward done observation abandoned episode_step')
  rng = np.random.default_rng(first_env)
  client = gs.Client(address)
  ids = np.arange(first_env, first_env + k, dtype=np.int32)
  runs = np.full(k, 7, np.int64)
  frames = rng.integers(0, 256, (k, 84, 84, 1)).astype(np.uint8)
  zeros_b, step = np.zeros(k, np.bool_), 0
  t_end = time.time() + seconds
  calls = 0
  try:
    while time.time() < t_end:
      env = EnvOutput(rng.normal(size=k).astype(np.float32), rng.uniform(size=k) < 0.01, frames, zeros_b,
                            np.full(k, step, np.int32))
      client.inference(ids, runs, env, env.reward)
      step += 1; calls += 1
  except gs.OpError:
    pass
  out.put(calls * k)


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--procs', type=int, default=8)
  ap.add_argument('--envs-per-proc', type=int, default=16)
  ap.add_argument('--n', type=int, default=64, help='server-side inference batch')
  ap.add_argument('--batch', type=int, default=64, help='train batch (unrolls)')
  ap.add_argument('--unroll', type=int, default=20)
  ap.add_argument('--seconds', type=float, default=10.0)
  a = ap.parse_args()
  import torch
  from seed_rl_amd import learner, learner_server, networks, optimizers, parametric_distribution as pd
  dev = torch.device('cuda:0')
  A, E = 18, a.procs * a.envs_per_proc
  agent = networks.AtariShallow(A, device=dev, seed=0)
  opt = optimizers.Adam(optimizers.PolynomialDecay(4.8e-4, 10 ** 6), beta_1=0.0, epsilon=3.125e-7)
  lrn = learner.Learner(agent, opt, pd.categorical_distribution(A))
  path = os.path.join(tempfile.gettempdir(), 'seedrl_' + uuid.uuid4().hex[:12])
  srv = learner_server.LearnerServer(agent, lrn, a.unroll, a.batch, a.n, E, (84, 84, 1), ['unix:' + path], device=dev)
  srv.start()
  ctx = mp.get_context('spawn')
  q = ctx.Queue()
  procs = [ctx.Process(target=actor_proc, args=('unix:' + path, i * a.envs_per_proc, a.envs_per_proc, a.seconds, q))
           for i in range(a.procs)]
  t0 = time.time()
  for p in procs:
    p.start()
  steps = 0
  while time.time() - t0 < a.seconds + 1:
    if srv.train_step(timeout=0.5) is not None:
      steps += 1
  dt = time.time() - t0
  srv.shutdown()                                      # unblocks the actors whose last batch can never fill
  total = sum(q.get(timeout=60) for _ in procs)
  for p in procs:
    p.join(timeout=30)
  if os.path.exists(path):
    os.remove(path)
  print('served %d env steps in %.1f s = %.0f env-steps/s through the Python transport (%d actor processes x %d envs, '
        'inference batch %d); %d train steps of %d unrolls x %d' % (total, dt, total / dt, a.procs, a.envs_per_proc, a.n,
                                                                   steps, a.batch, a.unroll))


if __name__ == '__main__':
  main()
