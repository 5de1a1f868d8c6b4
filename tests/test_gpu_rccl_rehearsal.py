"""RCCL rehearsal on ONE GPU (task 5): a process group `nccl` (= RCCL on ROCm) of world size 1, the
learner forced through the N-replica launch mode -- ready-range hooks, asynchronous range all-reduces on RCCL's stream,
`work.wait()` as a STREAM wait (gloo's is a host block: the difference `GraphedStep.__call__` leans on), segmented HIP
graphs, the remainder exchange, the update graph -- and held BIT-EQUAL to the unsplit single-graph step: a one-rank SUM
leaves the bucket as it is, so any difference is an ordering bug between the graph segments and the collective stream.
What the reference does implicitly through tf.distribute (agents/vtrace/learner.py:249-275).  Runs in a child process:
the process group must not leak into the other tests of this session."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np
import torch
os.environ['MASTER_ADDR'] = '127.0.0.1'
os.environ['MASTER_PORT'] = %(port)r
dev = torch.device('cuda:0')
torch.cuda.set_device(dev)
torch.distributed.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
assert torch.distributed.get_backend() == 'nccl'
from seed_rl_amd import learner, networks, optimizers, parametric_distribution as pd, smoke_step

A, T1, B, STEPS = 6, 6, 16, 3


def make(force):
  agent = networks.AtariShallow(A, device=dev, seed=0)
  opt = optimizers.Adam(optimizers.PolynomialDecay(4.8e-4, 50), beta_1=0.0, epsilon=3.125e-7, capturable=True)
  lrn = learner.Learner(agent, opt, pd.categorical_distribution(A), force_exchange=force)
  unroll = smoke_step.make_unroll(agent, T1, B, A, dev, seed=3, done_p=0.2)
  return lrn, unroll

# 1. the unsplit graph (what a single replica replays)
plain, u0 = make(False)
g0 = learner.GraphedStep(plain, u0, warmup=2)
assert not g0.split
l0 = []
for _ in range(STEPS):
  l0.append(float(g0()[0]))
torch.cuda.synchronize()

# 2. the N-replica launch mode on RCCL, one rank
forced, u1 = make(True)
assert forced.world == 1 and forced.exchanging
g1 = learner.GraphedStep(forced, u1, warmup=2)
assert g1.split and len(g1.segments) == 2, len(getattr(g1, 'segments', []))
rng = g1.segments[0][1]
n = forced.agent.flat.grads.numel()
assert rng is not None and 0 <= rng[0] < rng[1] <= n
assert (rng[1] - rng[0]) > 0.9 * n                       # Dense + heads: the overlapped part of the bucket
l1 = []
for _ in range(STEPS):
  l1.append(float(g1()[0]))
torch.cuda.synchronize()
assert l1 == l0, (l1, l0)
assert torch.equal(forced.agent.flat.params, plain.agent.flat.params)
sd0, sd1 = plain.optimizer.state_dict(), forced.optimizer.state_dict()
assert torch.equal(sd0['v'], sd1['v'])

# 3. the eager path with the hook-driven asynchronous exchange: same numbers again
eager, u2 = make(True)
seen = []
orig = eager._on_grads_ready
def spy(lo, hi):
  seen.append((lo, hi)); orig(lo, hi)
eager._on_grads_ready = spy
l2 = [float(eager.minimize(u2)[0]) for _ in range(STEPS)]
torch.cuda.synchronize()
assert seen and l2 == l0, (seen, l2, l0)
assert torch.equal(eager.agent.flat.params, plain.agent.flat.params)
torch.distributed.barrier()
torch.distributed.destroy_process_group()
print('RCCL_REHEARSAL_OK segments=%%d range=%%s' %% (len(g1.segments), (rng,)))
'''


def _free_port():
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  p = s.getsockname()[1]
  s.close()
  return p


def test_rccl_one_rank_split_graph_bit_equal():
  env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
  code = CHILD % dict(root=ROOT, port=str(_free_port()))
  r = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=600)
  assert r.returncode == 0 and 'RCCL_REHEARSAL_OK' in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


def test_bench_force_exchange_on_rccl():
  """`bench.py --gpus 1 --backend nccl --force-exchange`: the driver's N > 1 line with one rank -- `exchange.backend`
  must read nccl and the launch mode must be the segmented one."""
  import json
  env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', MASTER_PORT=str(_free_port()))
  r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--backend', 'nccl',
                      '--force-exchange', '--quick', '--steps', '5', '--warmup', '3', '--batch', '64'],
                     env=env, capture_output=True, text=True, timeout=900)
  assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
  line = json.loads([l for l in r.stdout.strip().splitlines() if l.startswith('{')][-1])
  assert line['exchange']['backend'] == 'nccl' and line['exchange']['ranks'] == 1
  assert 'segments' in line['config']['launch']
  assert line['exchange']['overlapped_ranges']
