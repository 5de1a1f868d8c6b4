"""Actors -> gRPC -> central inference -> completed unrolls -> dequeue -> Learner.minimize, all live
(seed_rl_amd/learner_server.py; the data path of agents/vtrace/learner.py:300-483): 12 actor threads keep stepping
synthetic Atari environments while the main thread takes 3 train steps of 8 unrolls each."""
import concurrent.futures as futures
import os
import tempfile
import threading
import uuid

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('transport,graphed', [('native', True), ('native', False), ('python', False)])
def test_actors_train_the_learner_end_to_end(device, transport, graphed):
  """transport: the native front-end (libseedserve.so) or the asyncio one; graphed: the train step replayed from HIP
  graphs (one per static unroll) on its own stream while inference batches run on the high-priority stream."""
  from seed_rl_amd import grpc_service as gs, learner, learner_server, networks, optimizers, utils
  from seed_rl_amd import parametric_distribution as pd
  T, B, A, n, E = 4, 8, 6, 4, 12
  obs_shape = (84, 84, 1)
  agent = networks.AtariShallow(A, device=device, seed=0)
  opt = optimizers.Adam(optimizers.PolynomialDecay(1e-3, 1000), beta_1=0.0, epsilon=3.125e-7, capturable=graphed)
  lrn = learner.Learner(agent, opt, pd.categorical_distribution(A))
  path = os.path.join(tempfile.gettempdir(), 'seedrl_' + uuid.uuid4().hex[:12])
  srv = learner_server.LearnerServer(agent, lrn, T, B, n, E, obs_shape, ['unix:' + path], device=device,
                                     transport=transport, graphed=graphed)
  assert srv.infer_agent is not agent and srv.infer_agent.flat is agent.flat      # same parameters, own workspaces
  srv.start()
  stop = threading.Event()
  steps_done = [0] * E

  def actor(env_id):
    rng = np.random.default_rng(env_id)
    client = gs.Client('unix:' + path)
    step = 0
    try:
      while not stop.is_set():
        env = utils.EnvOutput(np.float32(rng.normal()), np.bool_(step > 0 and rng.uniform() < 0.1),
                              rng.integers(0, 256, obs_shape).astype(np.uint8), np.bool_(False), np.int32(step))
        a = client.inference(np.int32(env_id), np.int64(77), env, np.float32(0.0))
        assert 0 <= int(a) < A
        step += 1
        steps_done[env_id] = step
    except gs.UnavailableError:
      pass                                            # server shut down under us: expected at the end
    finally:
      client.close()
  p0 = agent.flat.params.clone()
  losses = []
  with futures.ThreadPoolExecutor(max_workers=E) as ex:
    fs = [ex.submit(actor, e) for e in range(E)]
    try:
      for _ in range(3):
        out = srv.train_step(timeout=120)
        assert out is not None, 'no full batch of unrolls within the timeout'
        losses.append(float(out[0]))
    finally:
      stop.set()
      srv.shutdown()
      for f in fs:
        f.result(timeout=60)
  if os.path.exists(path):
    os.remove(path)
  srv.synchronize()
  srv.state.check_errors()
  assert all(np.isfinite(l) for l in losses) and not torch.equal(p0, agent.flat.params)
  assert srv.steps == 3 and min(steps_done) >= 2 * T + 1        # every actor kept being served while the learner trained
  # the unroll the last step trained on is on-policy data of SOME recent parameters: shapes / dtypes as the reference's
  u = srv.unroll
  assert tuple(u.env_outputs.observation.shape) == (T + 1, B) + obs_shape and u.agent_outputs.action.dtype == torch.int64
  assert bool((u.env_outputs.episode_step[1:] - u.env_outputs.episode_step[:-1] == 1)[~u.env_outputs.done[1:]].all())


def test_learner_loop_entry_point(device):
  """learner_loop(create_env_fn, create_agent_fn, create_optimizer_fn) with the reference's signature
  (agents/vtrace/learner.py:170-187): reads the specs from one environment, builds agent / optimizer through the
  factories, serves actors and trains until max_steps."""
  import collections
  from seed_rl_amd import grpc_service as gs, learner_server, networks, optimizers, utils
  Space = collections.namedtuple('Space', 'shape dtype n')
  A, E, T = 6, 8, 3
  created = {}

  class Env(object):
    observation_space = Space((84, 84, 1), np.uint8, None)
    action_space = Space((), torch.int64, A)

    def close(self):
      created['closed'] = True

  def create_agent(action_space, observation_space, dist):
    created['agent'] = (action_space.n, observation_space.shape, dist.param_size)
    return networks.AtariShallow(action_space.n, observation_shape=observation_space.shape, device=device, seed=0)

  def create_optimizer(final_iteration):
    created['final_iteration'] = final_iteration
    lr = optimizers.PolynomialDecay(1e-3, final_iteration)
    return optimizers.Adam(lr, beta_1=0.0, epsilon=3.125e-7, capturable=True), lr
  path = os.path.join(tempfile.gettempdir(), 'seedrl_' + uuid.uuid4().hex[:12])
  cfg = learner_server.LoopConfig(server_address='unix:' + path, batch_size=4, unroll_length=T, num_envs=E,
                                  inference_batch_size=4, total_environment_frames=10 ** 6, max_steps=3, device=device,
                                  step_timeout=120)
  stop = threading.Event()

  def actor(env_id):
    rng = np.random.default_rng(env_id)
    step = 0
    try:
      client = gs.Client('unix:' + path, timeout=120)
      while not stop.is_set():
        env = utils.EnvOutput(np.float32(rng.normal()), np.bool_(False), rng.integers(0, 256, (84, 84, 1)).astype(np.uint8),
                              np.bool_(False), np.int32(step))
        client.inference(np.int32(env_id), np.int64(5), env, np.float32(0.0))
        step += 1
    except gs.OpError:
      pass
  seen = []
  with futures.ThreadPoolExecutor(max_workers=E) as ex:
    fs = [ex.submit(actor, e) for e in range(E)]
    try:
      res = learner_server.learner_loop(lambda task: Env(), create_agent, create_optimizer, config=cfg,
                                        on_step=lambda it, loss, session: seen.append((it, float(loss))))
    finally:
      stop.set()
    for f in fs:
      f.result(timeout=60)
  assert created['closed'] and created['agent'] == (A, (84, 84, 1), A)
  assert created['final_iteration'] == int(np.ceil(10 ** 6 / (4 * T)))                  # learner.py:236-239
  assert res.iterations == 3 and res.num_env_frames == 3 * 4 * T and [it for it, _ in seen] == [1, 2, 3]
  assert all(np.isfinite(l) for _, l in seen) and 'losses/total' in res.last_session


def test_deep_lstm_agent_behind_the_native_server(device):
  """ImpalaDeep + LSTM (dmlab/networks.py:63-171) served and trained through LearnerServer: RGB observations through the
  native front-end, recurrent agent state in the per-env tables, the inference twin sharing the parameters, two train
  steps from HIP-graph replays while the actors keep stepping."""
  from seed_rl_amd import grpc_service as gs, learner, learner_server, networks, optimizers, utils
  from seed_rl_amd import parametric_distribution as pd
  T, B, A, n, E = 3, 4, 5, 4, 8
  obs_shape = (24, 32, 3)
  agent = networks.ImpalaDeep(A, observation_shape=obs_shape, device=device, seed=0)
  opt = optimizers.Adam(optimizers.PolynomialDecay(1e-3, 1000), beta_1=0.0, epsilon=3.125e-7, capturable=True)
  lrn = learner.Learner(agent, opt, pd.categorical_distribution(A))
  path = os.path.join(tempfile.gettempdir(), 'seedrl_' + uuid.uuid4().hex[:12])
  srv = learner_server.LearnerServer(agent, lrn, T, B, n, E, obs_shape, ['unix:' + path], device=device, graphed=True)
  srv.start()
  stop = threading.Event()

  def actor(env_id):
    rng = np.random.default_rng(env_id)
    step = 0
    try:
      client = gs.Client('unix:' + path, timeout=120)
      while not stop.is_set():
        env = utils.EnvOutput(np.float32(rng.normal()), np.bool_(step > 0 and rng.uniform() < 0.2),
                              rng.integers(0, 256, obs_shape).astype(np.uint8), np.bool_(False), np.int32(step))
        a = client.inference(np.int32(env_id), np.int64(3), env, np.float32(0.0))
        assert 0 <= int(a) < A
        step += 1
    except gs.OpError:
      pass
  p0 = agent.flat.params.clone()
  losses = []
  with futures.ThreadPoolExecutor(max_workers=E) as ex:
    fs = [ex.submit(actor, e) for e in range(E)]
    try:
      for _ in range(2):
        out = srv.train_step(timeout=120)
        assert out is not None
        losses.append(float(out[0]))
    finally:
      stop.set()
      srv.synchronize()
      srv.shutdown()
    for f in fs:
      f.result(timeout=60)
  srv.state.check_errors()
  agent.check_errors()
  assert all(np.isfinite(l) for l in losses) and not torch.equal(p0, agent.flat.params)
  u = srv.unroll
  assert tuple(u.env_outputs.observation.shape) == (T + 1, B) + obs_shape and len(u.agent_state) == 2
