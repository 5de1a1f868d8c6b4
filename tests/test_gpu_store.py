"""Device-resident UnrollStore / Aggregator / inference step:
the reference's known-answer sequences (tests/utils_test.py:70-286) replayed through the HIP row mover,
plus an end-to-end central-inference -> unroll -> learner consistency check."""
import collections

import numpy as np
import pytest
import torch

from tests import test_oracle_utils as seqs

pytestmark = pytest.mark.gpu


def _replay(device, store, seq, batch):
  out = []
  for i in range(0, len(seq) - len(seq) % batch, batch):
    chunk = seq[i:i + batch]
    ids = torch.tensor([c[1] for c in chunk], dtype=torch.int32, device=device)
    vals = torch.tensor([c[2] for c in chunk], dtype=torch.int32, device=device)
    rs = ids[torch.tensor([c[0] for c in chunk], device=device)]
    store.reset(rs)
    done, un = store.append(ids, vals)
    out.append((done.cpu().numpy(), un.cpu().numpy().T))          # time-major -> the reference's batch-major
  return out


def test_unroll_store_full_and_overlap(device):
  from seed_rl_amd.unroll_store import Spec, UnrollStore
  got = _replay(device, UnrollStore(4, 3, Spec((), torch.int32), device=device), seqs.FULL_SEQ, 3)
  for (ids, un), (eids, eun) in zip(got, seqs.FULL_EXPECT):
    np.testing.assert_array_equal(ids, np.array(eids, np.int64))
    np.testing.assert_array_equal(un.reshape(-1, 4), np.array(eun, np.int32).reshape(-1, 4))
  got = _replay(device, UnrollStore(2, 2, Spec((), torch.int32), num_overlapping_steps=2, device=device),
                seqs.OVERLAP_SEQ, 2)
  for (ids, un), (eids, eun) in zip(got, seqs.OVERLAP_EXPECT):
    np.testing.assert_array_equal(ids, np.array(eids, np.int64))
    np.testing.assert_array_equal(un.reshape(-1, 5), np.array(eun, np.int32).reshape(-1, 5))


def test_unroll_store_duplicates_structure_and_batch_assembly(device):
  from seed_rl_amd.unroll_store import Spec, UnrollStore
  store = UnrollStore(2, 3, Spec((), torch.int32), device=device)
  with pytest.raises(ValueError):
    store.append(torch.tensor([1, 1], device=device), torch.tensor([42, 43], dtype=torch.int32, device=device))
  nt = collections.namedtuple('named_tuple', 'x y')
  # frames-like rows (odd byte count -> byte path; 16-B multiple -> uint4 path)
  store = UnrollStore(3, 2, nt(Spec((5,), torch.uint8), Spec((4, 4), torch.float32)), device=device)
  batch = nt(torch.zeros((3, 6, 5), dtype=torch.uint8, device=device), torch.zeros((3, 6, 4, 4), device=device))
  rng = np.random.default_rng(0)
  hist = []
  for step in range(3):
    x = rng.integers(0, 255, (3, 5)).astype(np.uint8); y = rng.normal(size=(3, 4, 4)).astype(np.float32)
    hist.append((x, y))
    ids = torch.tensor([2, 0, 1], device=device)
    done, un = store.append(ids, nt(torch.tensor(x, device=device), torch.tensor(y, device=device)), out=batch, out_col=2)
  np.testing.assert_array_equal(done.cpu().numpy(), [2, 0, 1])
  # env e's data was given at position p of ids=[2,0,1]; completed unrolls land time-major in columns 2,3,4
  for col, env_pos in zip((2, 3, 4), (0, 1, 2)):
    for t in range(3):
      np.testing.assert_array_equal(batch.x[t, col].cpu().numpy(), hist[t][0][env_pos])
      np.testing.assert_array_equal(batch.y[t, col].cpu().numpy(), hist[t][1][env_pos])
  assert float(batch.y[:, :2].abs().sum()) == 0 and float(batch.y[:, 5:].abs().sum()) == 0


def test_aggregator(device):
  from seed_rl_amd.unroll_store import Aggregator, Spec
  agg = Aggregator(4, Spec((), torch.int32), device=device)          # tests/utils_test.py:276-286
  rd = lambda ids: agg.read(torch.tensor(ids, device=device)).cpu().numpy().tolist()
  assert rd([0, 1, 2, 3]) == [0, 0, 0, 0]
  agg.add([0, 1], torch.tensor([42, 43], dtype=torch.int32, device=device))
  assert rd([0, 1]) == [42, 43] and rd([0, 1, 2, 3]) == [42, 43, 0, 0]
  agg.reset([0])
  assert rd([0, 1, 2, 3]) == [0, 43, 0, 0]
  agg.replace([0, 2], torch.tensor([1, 2], dtype=torch.int32, device=device))
  assert rd([0, 1, 2, 3]) == [1, 43, 2, 0]
  with pytest.raises(ValueError):
    agg.replace([1, 1], torch.tensor([1, 2], dtype=torch.int32, device=device))


@pytest.mark.parametrize('kind', ['atari', 'deep'])
def test_central_inference_produces_on_policy_unrolls(device, kind):
  """learner.py:350-405 end to end on the device store: drive `inference` for 2*T+1 steps over 4 envs (two
  inference batches of 2, as two actors would), then check that re-running the learner's training unroll on
  each completed Unroll reproduces the behaviour logits / baseline stored at inference time (same weights):
  validates the single-step path, recurrent / frame-stacking state hand-over (first_agent_states), the
  one-step overlap carry-over and the time-major emission."""
  from seed_rl_amd import inference, networks, utils
  from seed_rl_amd.unroll_store import Spec
  T, E, A = 4, 4, 6
  rng = np.random.default_rng(0)
  if kind == 'atari':
    agent = networks.AtariShallow(A, device=device, seed=0)
    obs_shape = (84, 84, 1)
  else:
    agent = networks.ImpalaDeep(A, observation_shape=(24, 32, 3), device=device, seed=0)
    obs_shape = (24, 32, 3)
  env_specs = utils.EnvOutput(Spec((), torch.float32), Spec((), torch.bool), Spec(obs_shape, torch.uint8),
                              Spec((), torch.bool), Spec((), torch.int32))
  ao_specs = networks.AgentOutput(Spec((), torch.int64), Spec((A,), torch.float32), Spec((), torch.float32))
  unrolls, infos = [], []
  st = inference.InferenceState(agent, E, T, env_specs, ao_specs, Spec((), torch.int64), device=device,
                                unroll_sink=unrolls.append, info_sink=infos.append)
  run_ids = {e: 1000 + e for e in range(E)}
  for step in range(2 * T + 1):
    for ids in ([0, 2], [3, 1]):
      n = len(ids)
      done = rng.uniform(size=n) < (0.0 if step == 0 else 0.2)
      env = utils.EnvOutput(
          reward=torch.tensor(rng.normal(size=n).astype(np.float32), device=device),
          done=torch.tensor(done, device=device),
          observation=torch.tensor(rng.integers(0, 256, (n,) + obs_shape).astype(np.uint8), device=device),
          abandoned=torch.zeros(n, dtype=torch.bool, device=device),
          episode_step=torch.full((n,), step, dtype=torch.int32, device=device))
      act = st.inference(torch.tensor(ids, dtype=torch.int32), torch.tensor([run_ids[e] for e in ids]), env,
                         env.reward)
      assert act.shape == (n,) and int(act.max()) < A
  # T+1 steps complete the first unrolls, T more the second ones (one step of overlap)
  assert sum(int(u.env_outputs.done.shape[1]) for u in unrolls) == 2 * E
  for u in unrolls:
    assert u.env_outputs.done.shape[0] == T + 1
    out, _ = agent(u.prev_actions, u.env_outputs, u.agent_state, unroll=True, is_training=True)
    assert torch.allclose(out.policy_logits, u.agent_outputs.policy_logits, atol=2e-5)
    assert torch.allclose(out.baseline, u.agent_outputs.baseline, atol=2e-5)
  # overlap: the last step of an env's first unroll is the first step of its second unroll
  first, second = unrolls[0], unrolls[2]
  assert torch.equal(first.env_outputs.episode_step[-1], second.env_outputs.episode_step[0])
  assert torch.equal(first.agent_outputs.action[-1], second.agent_outputs.action[0])
  # an actor restart (new run id) resets that env's partial unroll and state
  env = utils.EnvOutput(torch.zeros(1, device=device), torch.zeros(1, dtype=torch.bool, device=device),
                        torch.zeros((1,) + obs_shape, dtype=torch.uint8, device=device),
                        torch.zeros(1, dtype=torch.bool, device=device), torch.zeros(1, dtype=torch.int32, device=device))
  st.inference(torch.tensor([2], dtype=torch.int32), torch.tensor([777]), env, env.reward)
  assert int(st.store._index[2]) == 1


def _drive(device, st_call, E, T, A, obs_shape, steps, seed=0):
  """Feeds the same synthetic actor traffic (two inference batches per env step, one actor restart, random
  episode ends) to an inference implementation; returns the list of actions."""
  from seed_rl_amd import utils
  rng = np.random.default_rng(seed)
  torch.manual_seed(123)
  acts = []
  run_ids = {e: 1000 + e for e in range(E)}
  for step in range(steps):
    if step == 3:
      run_ids[1] = 555                                           # actor of env 1 restarted
    for ids in ([0, 2], [3, 1]):
      n = len(ids)
      done = rng.uniform(size=n) < (0.0 if step == 0 else 0.25)
      env = utils.EnvOutput(
          reward=torch.tensor(rng.normal(size=n).astype(np.float32), device=device),
          done=torch.tensor(done, device=device),
          observation=torch.tensor(rng.integers(0, 256, (n,) + obs_shape).astype(np.uint8), device=device),
          abandoned=torch.zeros(n, dtype=torch.bool, device=device),
          episode_step=torch.full((n,), step, dtype=torch.int32, device=device))
      raw = torch.tensor(rng.normal(size=n).astype(np.float32), device=device)
      acts.append(st_call(torch.tensor(ids, dtype=torch.int32), torch.tensor([run_ids[e] for e in ids]), env, raw).clone())
  return acts


@pytest.mark.parametrize('kind,graphed', [('atari', False), ('deep', False), ('atari', True)])
def test_fused_inference_matches_reference_structured_inference(device, kind, graphed):
  """FusedInferenceState (no host syncs, masks + device scans, optional HIP-graph replay) vs InferenceState (the
  op-by-op mirror of learner.py:350-405) on identical actor traffic incl. an actor restart and episode ends:
  same actions, same completed unrolls (bit-exact, in completion order), same episode statistics."""
  from seed_rl_amd import inference, networks, utils
  from seed_rl_amd.unroll_store import Spec
  T, E, A = 3, 4, 6
  obs_shape = (84, 84, 1) if kind == 'atari' else (24, 32, 3)
  mk = (lambda: networks.AtariShallow(A, device=device, seed=0)) if kind == 'atari' else \
       (lambda: networks.ImpalaDeep(A, observation_shape=obs_shape, device=device, seed=0))
  env_specs = utils.EnvOutput(Spec((), torch.float32), Spec((), torch.bool), Spec(obs_shape, torch.uint8),
                              Spec((), torch.bool), Spec((), torch.int32))
  ao_specs = networks.AgentOutput(Spec((), torch.int64), Spec((A,), torch.float32), Spec((), torch.float32))
  unrolls, infos = [], []
  ref = inference.InferenceState(mk(), E, T, env_specs, ao_specs, Spec((), torch.int64), device=device,
                                 unroll_sink=unrolls.append, info_sink=infos.append)
  steps = 3 * T + 2
  acts_ref = _drive(device, ref.inference, E, T, A, obs_shape, steps)
  fused = inference.FusedInferenceState(mk(), E, T, env_specs, ao_specs, batch_capacity=16, device=device)
  call = fused.graphed(2, obs_shape) if graphed else fused.inference
  acts = _drive(device, call, E, T, A, obs_shape, steps)
  fused.check_errors()
  if not graphed:                     # graph replay draws its action-sampling randoms from the captured generator
    for a, b in zip(acts, acts_ref):  # state, so sampled actions differ; everything below is conditional on them
      assert torch.equal(a, b)
  k, batch = fused.take_batch()
  assert k == sum(int(u.env_outputs.done.shape[1]) for u in unrolls) and k > 0
  if graphed:
    # on-policy consistency instead of equality with the reference run: re-running the training unroll on every
    # emitted unroll reproduces the logits stored at inference time
    out, _ = fused.agent(batch.prev_actions, batch.env_outputs, batch.agent_state, unroll=True, is_training=True)
    assert torch.allclose(out.policy_logits, batch.agent_outputs.policy_logits, atol=2e-5)
    return
  cat = lambda xs, dim: torch.cat(xs, dim)
  ref_first = utils.map_structure(lambda *xs: cat(list(xs), 0), *[u.agent_state for u in unrolls])
  for a, b in zip(utils.flatten(ref_first), utils.flatten(batch.agent_state)):
    assert torch.equal(a, b)
  for name in ('prev_actions', 'env_outputs', 'agent_outputs'):
    ref_f = utils.map_structure(lambda *xs: cat(list(xs), 1), *[getattr(u, name) for u in unrolls])
    for a, b in zip(utils.flatten(ref_f), utils.flatten(getattr(batch, name))):
      assert torch.equal(a.to(b.dtype), b), name
  # episode statistics: same multiset of (frames, return, raw return)
  ref_stats = sorted((int(f), round(float(r), 5), round(float(w), 5)) for i in infos
                     for f, r, w in zip(i.episode_num_frames.tolist(), i.episode_returns.tolist(), i.episode_raw_returns.tolist()))
  ns = int(fused.stats_count[0])
  got = sorted((int(f), round(float(r), 5), round(float(w), 5)) for f, r, w in fused.episode_stats[:ns].tolist())
  assert got == ref_stats and ns > 0
