"""Data-parallel path on CPU: 2 processes, gloo backend (the GPU path uses the same code with RCCL).

Covers SURVEY.md 8(e): contiguous batch-column shards, ONE all-reduce(SUM) of the flat gradient bucket,
'mean' (global denominator => equals the single-replica gradient of the global batch) and 'sum' (the
reference's cross-replica semantics, tests/utils_test.py:609-650) reductions.  Per-shard gradients come from
the torch-CPU oracle of the loss head (the HIP kernels need a GPU); what is exercised here is the sharding,
the denominators and the exchange."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  p = s.getsockname()[1]
  s.close()
  return p


def _shard_grad(tgt, base, beh, act, rew, done, denom):
  """d(sum-of-losses / denom) / d(logits, baseline) on a shard, via the torch oracle."""
  from oracle import nets_torch
  t = torch.tensor(tgt, requires_grad=True)
  b = torch.tensor(base, requires_grad=True)
  total, _ = nets_torch.impala_loss_torch(t, b, torch.tensor(beh), torch.tensor(act), torch.tensor(rew),
                                          torch.tensor(done))
  n_local = (tgt.shape[0] - 1) * tgt.shape[1]
  (total * n_local / denom).backward()            # oracle takes a local mean; re-normalise to `denom`
  return torch.cat([t.grad.reshape(-1), b.grad.reshape(-1)])


def _worker(rank, world, port, reduction, out):
  sys.path.insert(0, ROOT)
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  torch.distributed.init_process_group('gloo', rank=rank, world_size=world)
  try:
    from seed_rl_amd import learner
    from seed_rl_amd.flat import FlatParams
    from tests import synth
    T, B, A = 6, 8, 5
    tgt, base, beh, act, rew, done = synth.loss_inputs(0, T, B, A)
    cols = learner.shard_columns(B, rank, world)
    n_global = T * B
    denom = n_global if reduction == 'mean' else T * (B // world)
    g = _shard_grad(tgt[:, cols], base[:, cols], beh[:, cols], act[:, cols], rew[:, cols], done[:, cols], denom)
    # the exchange runs on a flat bucket exactly like the GPU path (here: CPU tensors, gloo)
    flat = FlatParams([('x', (g.numel(),))], torch.device('cpu'))
    flat.grads.copy_(g)
    learner.all_reduce_gradients(flat.grads)
    # place the shard gradient into global column order (zero elsewhere) and sum over ranks: what a
    # single replica would have computed on the global batch
    full_l = torch.zeros((T + 1, B, A)); full_b = torch.zeros((T + 1, B))
    per = B // world
    full_l[:, cols] = g[:(T + 1) * per * A].reshape(T + 1, per, A)
    full_b[:, cols] = g[(T + 1) * per * A:].reshape(T + 1, per)
    padded = torch.cat([full_l.reshape(-1), full_b.reshape(-1)])
    torch.distributed.all_reduce(padded)
    if rank == 0:
      torch.save(dict(padded=padded, reduced_shard_sum=flat.grads.clone()), out)
  finally:
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize('reduction', ['mean', 'sum'])
def test_data_parallel_gradient_exchange_gloo(tmp_path, reduction):
  world, port = 2, _free_port()
  out = str(tmp_path / 'r0.pt')
  mp.spawn(_worker, args=(world, port, reduction, out), nprocs=world, join=True)
  got = torch.load(out)
  sys.path.insert(0, ROOT)
  from tests import synth
  T, B, A = 6, 8, 5
  tgt, base, beh, act, rew, done = synth.loss_inputs(0, T, B, A)
  # single-replica gradient of the GLOBAL batch
  scale = 1.0 if reduction == 'mean' else float(world)     # 'sum': each replica's mean is over B/world columns
  ref = _shard_grad(tgt, base, beh, act, rew, done, T * B) * scale
  np.testing.assert_allclose(got['padded'].numpy(), ref.numpy(), rtol=1e-5, atol=1e-7)


def _sgd_worker(rank, world, port, out):
  """tests/utils_test.py:609-650 (MinimizeTest): loss = 2a per replica, gradients SUMMED across replicas,
  SGD(0.1): a = 1 - world * 0.2 on every replica."""
  sys.path.insert(0, ROOT)
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  torch.distributed.init_process_group('gloo', rank=rank, world_size=world)
  try:
    from seed_rl_amd import learner
    a = torch.tensor([1.0])
    grad = torch.tensor([2.0])                       # d(2a)/da on this replica
    learner.all_reduce_gradients(grad)               # reference semantics: SUM
    a -= 0.1 * grad
    torch.save(a, out + str(rank))
  finally:
    torch.distributed.destroy_process_group()


def test_minimize_sum_semantics_like_reference(tmp_path):
  world, port = 2, _free_port()
  out = str(tmp_path / 'a')
  mp.spawn(_sgd_worker, args=(world, port, out), nprocs=world, join=True)
  for r in range(world):
    assert abs(float(torch.load(out + str(r))[0]) - (1.0 - world * 0.2)) < 1e-6


def _overlap_worker(rank, world, port, mode, out):
  sys.path.insert(0, ROOT)
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  torch.distributed.init_process_group('gloo', rank=rank, world_size=world)
  try:
    from seed_rl_amd import learner
    from seed_rl_amd.flat import FlatParams

    class _Agent(object):
      """Stand-in with the agents' protocol: backward() fills flat.grads and reports final ranges through
      grad_ready_hook(lo, hi) -- 'two': tail first then head (as the torsos do), 'tail': only the tail (the rest
      must be exchanged by reduce_gradients), 'none': no reports at all."""
      grad_ready_hook = None

      def __init__(self):
        self.flat = FlatParams([('conv', (37,)), ('fc', (101,)), ('heads', (9,))], torch.device('cpu'))

      def backward(self):
        g = torch.arange(self.flat.size, dtype=torch.float32) * (rank + 1) + rank
        self.flat.grads.copy_(g)
        hook, split = self.grad_ready_hook, self.flat.offsets['fc']
        if hook is not None and mode in ('two', 'tail'):
          hook(split, self.flat.size)
        if hook is not None and mode == 'two':
          hook(0, split)

    agent = _Agent()
    lrn = learner.Learner(agent, None, None)
    assert lrn.world == world
    agent.grad_ready_hook = lrn._on_grads_ready
    agent.backward()
    agent.grad_ready_hook = None
    assert len(lrn._pending) == {'two': 2, 'tail': 1, 'none': 0}[mode]
    lrn.reduce_gradients()
    base = torch.arange(agent.flat.size, dtype=torch.float32)
    want = sum(base * (r + 1) + r for r in range(world))
    assert torch.equal(agent.flat.grads, want)
    assert lrn._pending == []
    if rank == 0:
      open(out, 'w').write('ok')
  finally:
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize('mode', ['two', 'tail', 'none'])
def test_overlapped_gradient_exchange_gloo(tmp_path, mode):
  """The gradient exchange overlapped with the backward pass (Learner._on_grads_ready / reduce_gradients): ranges the
  agent reports are all-reduced asynchronously, unreported ranges by reduce_gradients; the result is the SUM of the
  whole flat bucket in every case."""
  out = str(tmp_path / 'ok.txt')
  mp.spawn(_overlap_worker, args=(2, _free_port(), mode, out), nprocs=2, join=True)
  assert open(out).read() == 'ok'


def test_shard_columns():
  from seed_rl_amd import learner
  assert learner.shard_columns(4096, 3, 8) == slice(1536, 2048)
  with pytest.raises(ValueError):
    learner.shard_columns(10, 0, 4)


def _force_worker(rank, world, port, out):
  sys.path.insert(0, ROOT)
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  torch.distributed.init_process_group('gloo', rank=rank, world_size=world)
  try:
    from seed_rl_amd import learner
    from seed_rl_amd.flat import FlatParams

    class _Agent(object):
      grad_ready_hook = None

      def __init__(self):
        self.flat = FlatParams([('conv', (37,)), ('fc', (101,))], torch.device('cpu'))

      def backward(self):
        self.flat.grads.copy_(torch.arange(self.flat.size, dtype=torch.float32) + 1.0)
        if self.grad_ready_hook is not None:
          self.grad_ready_hook(self.flat.offsets['fc'], self.flat.size)

    plain = learner.Learner(_Agent(), None, None)
    assert plain.world == 1 and not plain.exchanging
    forced = learner.Learner(_Agent(), None, None, force_exchange=True)
    assert forced.world == 1 and forced.exchanging
    forced.agent.grad_ready_hook = forced._on_grads_ready
    forced.agent.backward()
    forced.agent.grad_ready_hook = None
    assert len(forced._pending) == 1                 # the reported range flew as an asynchronous all-reduce of ONE rank
    forced.reduce_gradients()                        # ... and the remainder is exchanged here
    assert forced._pending == []
    assert torch.equal(forced.agent.flat.grads, torch.arange(forced.agent.flat.size, dtype=torch.float32) + 1.0)   # a one-rank SUM is the identity
    open(out, 'w').write('ok')
  finally:
    torch.distributed.destroy_process_group()


def test_force_exchange_with_one_rank(tmp_path):
  """Learner(force_exchange=True): the N-replica exchange path (ready-range hook, asynchronous range all-reduce, remainder
  in reduce_gradients) in a process group of ONE rank -- the CPU twin of tests/test_gpu_rccl_rehearsal.py."""
  out = str(tmp_path / 'ok.txt')
  mp.spawn(_force_worker, args=(1, _free_port(), out), nprocs=1, join=True)
  assert open(out).read() == 'ok'
  from seed_rl_amd import learner
  with pytest.raises(ValueError):
    learner.Learner(object(), None, None, force_exchange=True)      # no process group in THIS process


def _guard_worker(rank, world, port, out):
  """Lock-step abort: rank 1's LSTM sequence kernel "times out" in step 2 (its sticky word is set, its
  gradients are garbage).  The word travels with the gradient exchange (MAX), so BOTH ranks drop that step -- the stand-in
  optimizer is guarded by flat.step_guard exactly like csrc/adam.hip -- both see the word afterwards (the host's demotion
  signal), and the replicas stay bit-equal through the steps that follow."""
  sys.path.insert(0, ROOT)
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  torch.distributed.init_process_group('gloo', rank=rank, world_size=world)
  try:
    from seed_rl_amd import learner
    from seed_rl_amd.flat import FlatParams

    class _Agent(object):
      grad_ready_hook = None

      def __init__(self):
        self.flat = FlatParams([('conv', (37,)), ('core', (101,))], torch.device('cpu'))
        self.flat.params.fill_(1.0)
        self.flat.step_guard = torch.zeros(1, dtype=torch.int32)
        self.step, self.fault_at = 0, None

      def backward(self):
        g = torch.arange(self.flat.size, dtype=torch.float32) * 1e-3 * (rank + 1) + self.step
        if self.step == self.fault_at:                  # the aborted kernel: garbage gradients + the sticky word
          g = torch.full_like(g, float('nan'))
          self.flat.step_guard.fill_(1)
        self.flat.grads.copy_(g)
        if self.grad_ready_hook is not None:
          self.grad_ready_hook(self.flat.offsets['core'], self.flat.size)
        self.step += 1

    class _GuardedSGD(object):
      applied = 0

      def apply_gradients(self, flat):
        if int(flat.step_guard[0]) == 0:                # csrc/adam.hip: the update is dropped while the word is set
          flat.params -= 0.1 * flat.grads
          self.applied += 1

    agent, opt = _Agent(), _GuardedSGD()
    agent.fault_at = 2 if rank == 1 else None
    lrn = learner.Learner(agent, opt, None)
    seen = []
    for step in range(5):
      agent.grad_ready_hook = lrn._on_grads_ready
      agent.backward()
      agent.grad_ready_hook = None
      lrn.apply_gradients()
      seen.append(int(agent.flat.step_guard[0]))
      if seen[-1]:                                       # what _lstm_seq_check does one step later: demote, clear the word
        agent.flat.step_guard.zero_()
    assert seen == [0, 0, 1, 0, 0], seen                 # BOTH ranks saw the word in step 2
    assert opt.applied == 4
    assert bool(torch.isfinite(agent.flat.params).all())
    torch.save(agent.flat.params.clone(), out + str(rank))
  finally:
    torch.distributed.destroy_process_group()


def test_lstm_abort_word_travels_with_the_gradient_exchange(tmp_path):
  world, port = 2, _free_port()
  out = str(tmp_path / 'p')
  mp.spawn(_guard_worker, args=(world, port, out), nprocs=world, join=True)
  p0, p1 = torch.load(out + '0'), torch.load(out + '1')
  assert torch.equal(p0, p1)
