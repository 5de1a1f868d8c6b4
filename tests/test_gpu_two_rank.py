"""A REAL two-replica learner on one GPU: two processes, both on cuda:0, process group `gloo`
(it moves device tensors; RCCL refuses two ranks on one device).  Each replica runs the real AtariShallow
`Learner.minimize` on its `shard_columns` half of the batch -- real backward pass, real `_on_grads_ready` overlap
(asynchronous range all-reduces issued from inside the backward), real Adam -- and the parameters after two steps
must equal a single-process run:
  reduction='mean': the single-replica step on the GLOBAL batch (<= 1e-6);
  reduction='sum' : the reference's cross-replica semantics (per-replica mean losses, gradients SUMMED,
                    /root/reference/tests/utils_test.py:609-650): one process that adds the two shards' gradients.
`graphed=True` runs the same through learner.GraphedStep's SPLIT mode (compute_gradients as a chain of graph segments
cut where a gradient range becomes final, that range's exchange launched between them, the rest after the last segment,
then the update graph).  What stays unmeasured: RCCL/xGMI itself with more than one rank (no multi-GPU box here)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
T1, BG, A, STEPS = 6, 16, 6, 2


def _free_port():
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  p = s.getsockname()[1]
  s.close()
  return p


def _make(device, reduction, cols, capturable=False, seed=3):
  from seed_rl_amd import learner, networks, optimizers, utils, parametric_distribution as pd
  from tests import synth
  u = synth.atari_unroll(seed, T1, BG, A, done_p=0.1, zero_state=False)
  t = lambda a: torch.as_tensor(np.ascontiguousarray(a)).to(device)
  agent = networks.AtariShallow(A, device=device, seed=5)
  opt = optimizers.Adam(optimizers.PolynomialDecay(4.8e-4, 100), beta_1=0.0, epsilon=3.125e-7, capturable=capturable)
  lrn = learner.Learner(agent, opt, pd.categorical_distribution(A), reduction=reduction)
  env = utils.EnvOutput(t(u['reward'][:, cols]), t(u['done'][:, cols]), t(u['frames'][:, cols]), None, None)
  ao = networks.AgentOutput(t(u['actions'][:, cols]), t(u['behaviour_logits'][:, cols]),
                            t(u['behaviour_baseline'][:, cols]))
  unroll = learner.Unroll(networks.AgentState((), t(u['frame_state'][cols])), t(u['prev_actions'][:, cols]), env, ao)
  return agent, lrn, unroll


def _worker(rank, world, port, reduction, graphed, out):
  sys.path.insert(0, ROOT)
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  torch.distributed.init_process_group('gloo', rank=rank, world_size=world)
  try:
    from seed_rl_amd import learner
    dev = torch.device('cuda:0')
    torch.cuda.set_device(dev)
    cols = learner.shard_columns(BG, rank, world)
    agent, lrn, unroll = _make(dev, reduction, cols, capturable=graphed)
    assert lrn.world == world
    ranges = []
    if graphed:
      step = learner.GraphedStep(lrn, unroll, warmup=1)
      assert step.split and step.graph2 is not None
      # compute_gradients is cut where the agent reports the Dense + heads range as final: that range's exchange is
      # launched between the two segments and flies under the conv backward
      fl = agent.flat
      assert [r for _, r in step.segments] == [(fl.offsets['fc/kernel'], fl.size), None], step.segments
      for _ in range(STEPS):
        step()
    else:
      orig = lrn._on_grads_ready

      def spy(lo, hi):
        ranges.append((lo, hi))
        orig(lo, hi)
      lrn._on_grads_ready = spy
      for _ in range(STEPS):
        lrn.minimize(unroll)
    torch.cuda.synchronize()
    torch.save(dict(params=agent.flat.params.cpu(), ranges=ranges, size=agent.flat.size,
                    split=agent.flat.offsets['fc/kernel']), out + str(rank))
  finally:
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize('graphed', [False, True])
@pytest.mark.parametrize('reduction', ['mean', 'sum'])
def test_two_rank_learner_matches_single_process(device, tmp_path, reduction, graphed):
  world, port = 2, _free_port()
  out = str(tmp_path / 'rank')
  mp.spawn(_worker, args=(world, port, reduction, graphed, out), nprocs=world, join=True)
  got = [torch.load(out + str(r)) for r in range(world)]
  # replicas stay identical (same summed gradient, same Adam state)
  assert torch.equal(got[0]['params'], got[1]['params'])
  if not graphed:
    # the overlap really ran from inside the backward pass: tail range (Dense + heads) first, then the conv range
    size, split = got[0]['size'], got[0]['split']
    assert got[0]['ranges'] == [(split, size), (0, split)] * STEPS, got[0]['ranges']

  # ---- single-process reference on the same GPU ----
  from seed_rl_amd import learner
  if reduction == 'mean':
    agent, lrn, unroll = _make(device, 'mean', slice(0, BG))
    for _ in range(STEPS):
      lrn.minimize(unroll)
    want = agent.flat.params.cpu()
    tol = 1e-6
  else:
    agent, lrn, _ = _make(device, 'sum', slice(0, BG))
    shards = [_make(device, 'sum', learner.shard_columns(BG, r, world))[2] for r in range(world)]
    for _ in range(STEPS):
      total = torch.zeros_like(agent.flat.grads)
      for u in shards:                                      # per-replica mean loss, gradients SUMMED
        lrn.compute_gradients(u)
        total += agent.flat.grads
      agent.flat.grads.copy_(total)
      lrn.apply_gradients()
    want = agent.flat.params.cpu()
    tol = 1e-6
  err = float((got[0]['params'] - want).abs().max())
  # Adam(beta_1 = 0) turns a gradient element at the fp32 noise floor into a +-lr step: allow isolated sign flips
  # (different summation order: shard sums vs one batch) but not a systematic difference
  frac = float(((got[0]['params'] - want).abs() > tol).float().mean())
  assert err <= 2.2 * 4.8e-4 and frac <= 1e-3, (err, frac)


def _serving_worker(rank, world, port, graphed, out, sock_prefix):
  """One data-parallel learner replica WITH its own LearnerServer: actors of this rank's env shard connect to this
  rank's address (the reference assigns actors to inference devices, learner.py:406-414), the gradient exchange
  happens inside Learner.minimize, nothing else crosses ranks."""
  import concurrent.futures as futures
  import threading
  sys.path.insert(0, ROOT)
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  torch.distributed.init_process_group('gloo', rank=rank, world_size=world)
  try:
    from seed_rl_amd import grpc_service as gs, learner, learner_server, networks, optimizers, utils
    from seed_rl_amd import parametric_distribution as pd
    dev = torch.device('cuda:0')
    torch.cuda.set_device(dev)
    T, B, n, E = 3, 4, 4, 8
    agent = networks.AtariShallow(A, device=dev, seed=5)              # identical initial weights on every rank
    opt = optimizers.Adam(optimizers.PolynomialDecay(1e-3, 100), beta_1=0.0, epsilon=3.125e-7, capturable=graphed)
    lrn = learner.Learner(agent, opt, pd.categorical_distribution(A), reduction='mean')
    assert lrn.world == world
    address = 'unix:%s%d' % (sock_prefix, rank)
    srv = learner_server.LearnerServer(agent, lrn, T, B, n, E, (84, 84, 1), [address], device=dev, graphed=graphed)
    srv.start()
    stop = threading.Event()

    def actor(env_id):
      rng = np.random.default_rng(100 * rank + env_id)             # every rank's actors see different data
      step = 0
      try:
        client = gs.Client(address, timeout=120)
        while not stop.is_set():
          env = utils.EnvOutput(np.float32(rng.normal()), np.bool_(False),
                                rng.integers(0, 256, (84, 84, 1)).astype(np.uint8), np.bool_(False), np.int32(step))
          client.inference(np.int32(env_id), np.int64(9), env, np.float32(0.0))
          step += 1
      except gs.OpError:
        pass
    losses = []
    with futures.ThreadPoolExecutor(max_workers=E) as ex:
      fs = [ex.submit(actor, e) for e in range(E)]
      try:
        for _ in range(STEPS):
          o = srv.train_step(timeout=120)
          assert o is not None
          losses.append(float(o[0]))
      finally:
        stop.set()
        srv.synchronize()
        srv.shutdown()
      for f in fs:
        f.result(timeout=60)
    torch.save(dict(params=agent.flat.params.cpu(), losses=losses), out + str(rank))
  finally:
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize('graphed', [False, True])
def test_two_rank_learner_servers(device, tmp_path, graphed):
  """Serving under data parallelism: two replicas, each behind its own native gRPC front-end with its own actors,
  different trajectories per rank, ONE gradient all-reduce per step -- after every step both hold the same weights
  (so no weight broadcast to the inference side is ever needed), and they moved away from the initial ones."""
  import tempfile
  import uuid
  world, port = 2, _free_port()
  out = str(tmp_path / 'srv')
  prefix = os.path.join(tempfile.gettempdir(), 'seedrl_dp_' + uuid.uuid4().hex[:8] + '_')
  mp.spawn(_serving_worker, args=(world, port, graphed, out, prefix), nprocs=world, join=True)
  got = [torch.load(out + str(r)) for r in range(world)]
  assert torch.equal(got[0]['params'], got[1]['params'])
  from seed_rl_amd import networks
  init = networks.AtariShallow(A, device=device, seed=5).flat.params.cpu()
  assert not torch.equal(got[0]['params'], init)
  assert all(np.isfinite(l) for g in got for l in g['losses'])
  assert got[0]['losses'] != got[1]['losses']                         # different data on the two ranks


def _abort_worker(rank, world, port, out):
  """Two data-parallel replicas of an LSTM agent through GraphedStep's split mode; rank 1's sequence kernel "times out"
  (its sticky word is armed the way the kernel arms it, as in tests/test_gpu_lstm_demotion.py).  The word travels with the
  gradient exchange (MAX): both ranks drop the step, both see the word one step later, both demote and capture their
  graphs again AT THE SAME STEP (the new capture's warm-up all-reduces meet their partners: no hang), and the replicas stay
  bit-equal."""
  sys.path.insert(0, ROOT)
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  torch.distributed.init_process_group('gloo', rank=rank, world_size=world)
  try:
    from seed_rl_amd import learner, networks, ops, optimizers, parametric_distribution as pd, smoke_step
    dev = torch.device('cuda:0')
    torch.cuda.set_device(dev)
    Ad, T1d, Bd = 6, 21, 32
    if not ops.lstm_seq_supported(T1d, Bd, 256):
      torch.save(dict(skipped=True), out + str(rank))
      return
    agent = networks.ImpalaDeep(Ad, observation_shape=(24, 32, 3), device=dev, seed=0)
    opt = optimizers.Adam(optimizers.PolynomialDecay(4.8e-4, 50), beta_1=0.0, epsilon=3.125e-7, capturable=True)
    unroll = smoke_step.make_deep_unroll(agent, T1d, Bd, Ad, dev, seed=3 + rank, done_p=0.2)
    lrn = learner.Learner(agent, opt, pd.categorical_distribution(Ad))
    step = learner.GraphedStep(lrn, unroll, warmup=1)
    assert step.split and step._captured_seq
    step(); torch.cuda.synchronize()
    p1 = agent.flat.params.clone()
    if rank == 1:
      agent._seq_sticky().fill_(1)                              # rank 1 only: its sequence kernel "timed out"
    step(); torch.cuda.synchronize()
    dropped = torch.equal(agent.flat.params, p1)                # BOTH ranks must have dropped this step
    recaptured = []
    for _ in range(3):
      step(); torch.cuda.synchronize()
      recaptured.append(bool(step.recaptured))
    torch.save(dict(skipped=False, dropped=dropped, recaptured=recaptured, demoted=bool(agent._seq_demoted),
                    sticky=int(agent._seq_sticky()[0]), params=agent.flat.params.cpu(),
                    moved=not torch.equal(agent.flat.params, p1)), out + str(rank))
  finally:
    torch.distributed.destroy_process_group()


def test_two_rank_lstm_abort_is_lock_step(device, tmp_path):
  world, port = 2, _free_port()
  out = str(tmp_path / 'abort')
  mp.spawn(_abort_worker, args=(world, port, out), nprocs=world, join=True)
  got = [torch.load(out + str(r)) for r in range(world)]
  if got[0]['skipped']:
    pytest.skip('sequence kernels not available for this shape')
  for g in got:
    assert g['dropped'] and g['demoted'] and g['sticky'] == 0 and g['moved'], {k: v for k, v in g.items() if k != 'params'}
  assert got[0]['recaptured'] == got[1]['recaptured'] and sum(got[0]['recaptured']) == 1, got[0]['recaptured']
  assert torch.equal(got[0]['params'], got[1]['params'])
