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


@pytest.mark.parametrize('kind,graphed,T', [('atari', False, 3), ('deep', False, 3), ('atari', True, 3), ('deep', True, 3),
                                            ('atari', False, 4), ('atari', True, 5), ('atari', True, 7)])
def test_fused_inference_matches_reference_structured_inference(device, kind, graphed, T):
  """FusedInferenceState (no host syncs, masks + device scans, optional HIP-graph replay) vs InferenceState (the
  op-by-op mirror of learner.py:350-405) on identical actor traffic incl. an actor restart and episode ends:
  same actions, same completed unrolls (bit-exact, in completion order), same episode statistics.
  T >= 4 with the Atari agent: the six-launch path of csrc/servestep.hip (frame stack read from the store's own frames);
  T = 3: the generic path (stores shorter than five slots, any agent)."""
  from seed_rl_amd import inference, networks, utils
  from seed_rl_amd.unroll_store import Spec
  E, A = 4, 6
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
  assert fused._serve == (kind == 'atari' and T >= 4)
  call = fused.graphed(2, obs_shape) if graphed else fused.inference
  acts = _drive(device, call, E, T, A, obs_shape, steps)
  fused.check_errors()
  # the action sampler is counter-based (seed, call, row) and advanced on the device: the mirror (stand-alone sampling
  # kernel), the fused step (sampling inside inference_post) and its HIP-graph replay draw the SAME actions
  for a, b in zip(acts, acts_ref):
    assert torch.equal(a, b)
  k, batch = fused.take_batch()
  assert k == sum(int(u.env_outputs.done.shape[1]) for u in unrolls) and k > 0
  # on-policy consistency: re-running the training unroll on every emitted unroll reproduces the logits stored at
  # inference time
  out, _ = fused.agent(batch.prev_actions, batch.env_outputs, batch.agent_state, unroll=True, is_training=True)
  assert torch.allclose(out.policy_logits, batch.agent_outputs.policy_logits, atol=2e-5)
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


@pytest.mark.parametrize('T', [2, 4])
def test_fused_inference_batches_above_1024_rows(device, T):
  """(T = 4: the six-launch path, serve_begin walks the rows in 1024-row chunks too.)
  Inference batches of 1300 rows (inference_pre / inference_post walk them in 1024-row chunks; the rank of a
  completed unroll in the training batch is its rank in env_ids order ACROSS chunks): same actions, same completed
  unrolls in the same order, same episode statistics as the op-by-op mirror, with restarts that put the envs' unroll
  phases out of step so that completions fall in both chunks of every batch."""
  from seed_rl_amd import inference, networks, utils
  from seed_rl_amd.unroll_store import Spec
  n, A = 1300, 6
  E = 2 * n
  obs_shape = (84, 84, 1)
  mk = lambda: networks.AtariShallow(A, device=device, seed=0)
  env_specs = utils.EnvOutput(Spec((), torch.float32), Spec((), torch.bool), Spec(obs_shape, torch.uint8),
                              Spec((), torch.bool), Spec((), torch.int32))
  ao_specs = networks.AgentOutput(Spec((), torch.int64), Spec((A,), torch.float32), Spec((), torch.float32))

  def drive(call):
    rng = np.random.default_rng(7)
    torch.manual_seed(5)
    run_ids = np.full(E, 1000, np.int64)
    acts = []
    for step in range(3 * T + 1):
      if step in (2, 3):                                         # a third of the actors restart: their unrolls re-phase
        who = rng.uniform(size=E) < 0.33
        run_ids[who] += step
      perm = rng.permutation(E)
      for ids in (perm[:n], perm[n:]):
        done = rng.uniform(size=n) < (0.0 if step == 0 else 0.2)
        env = utils.EnvOutput(
            reward=torch.tensor(rng.normal(size=n).astype(np.float32), device=device),
            done=torch.tensor(done, device=device),
            observation=torch.tensor(rng.integers(0, 256, (n,) + obs_shape).astype(np.uint8), device=device),
            abandoned=torch.zeros(n, dtype=torch.bool, device=device),
            episode_step=torch.full((n,), step, dtype=torch.int32, device=device))
        raw = torch.tensor(rng.normal(size=n).astype(np.float32), device=device)
        acts.append(call(torch.tensor(ids, dtype=torch.int32), torch.tensor(run_ids[ids]), env, raw).clone())
    return acts

  unrolls, infos = [], []
  ref = inference.InferenceState(mk(), E, T, env_specs, ao_specs, Spec((), torch.int64), device=device,
                                 unroll_sink=unrolls.append, info_sink=infos.append)
  acts_ref = drive(ref.inference)
  fused = inference.FusedInferenceState(mk(), E, T, env_specs, ao_specs, batch_capacity=4 * E, device=device,
                                        stats_capacity=16384)
  assert fused._serve == (T >= 4)
  acts = drive(fused.graphed(n, obs_shape))
  fused.check_errors()
  for a, b in zip(acts, acts_ref):
    assert torch.equal(a, b)
  k, batch = fused.take_batch()
  per_call = [int(u.env_outputs.done.shape[1]) for u in unrolls]
  assert k == sum(per_call) and k > E and max(per_call) > 300       # completions well inside the second chunk too
  cat = lambda xs, dim: torch.cat(xs, dim)
  for name in ('prev_actions', 'env_outputs', 'agent_outputs'):
    ref_f = utils.map_structure(lambda *xs: cat(list(xs), 1), *[getattr(u, name) for u in unrolls])
    for a, b in zip(utils.flatten(ref_f), utils.flatten(getattr(batch, name))):
      if a.dtype.is_floating_point and name == 'agent_outputs':      # two agent instances: same weights, same kernels
        assert torch.equal(a, b), name
      else:
        assert torch.equal(a.to(b.dtype), b), name
  ref_first = utils.map_structure(lambda *xs: cat(list(xs), 0), *[u.agent_state for u in unrolls])
  for a, b in zip(utils.flatten(ref_first), utils.flatten(batch.agent_state)):
    assert torch.equal(a, b)
  ref_stats = sorted((int(f), round(float(r), 5), round(float(w), 5)) for i in infos
                     for f, r, w in zip(i.episode_num_frames.tolist(), i.episode_returns.tolist(), i.episode_raw_returns.tolist()))
  ns = int(fused.stats_count[0])
  got = sorted((int(f), round(float(r), 5), round(float(w), 5)) for f, r, w in fused.episode_stats[:ns].tolist())
  assert got == ref_stats and ns > 100


def test_categorical_sample_kernel(device):
  """seedhip_categorical_sample: frequencies follow softmax(logits) (chi-square over 2e5 draws, A = 18), the sample is
  a pure function of (seed, counter, row), and strided head rows are read in place."""
  from seed_rl_amd import ops
  rng = np.random.default_rng(0)
  A, ld, rows = 18, 20, 200000
  logit_row = rng.normal(size=A).astype(np.float32) * 1.5
  head = np.zeros((rows, ld), np.float32); head[:, :A] = logit_row; head[:, A:] = 50.0     # pad columns must be ignored
  hd = torch.tensor(head, device=device)
  st = torch.tensor([1234, 0], dtype=torch.int64, device=device)
  act = torch.empty(rows, dtype=torch.int64, device=device)
  ops.categorical_sample(hd, ld, rows, A, st, act)
  a = act.cpu().numpy()
  assert a.min() >= 0 and a.max() < A and int(st[1]) == 1
  p = np.exp(logit_row - logit_row.max()); p /= p.sum()
  cnt = np.bincount(a, minlength=A)
  chi2 = float(((cnt - rows * p) ** 2 / (rows * p)).sum())
  assert chi2 < 50.0, chi2                       # 17 degrees of freedom: P(chi2 > 50) ~ 4e-5
  # determinism: same (seed, counter) -> same actions; next counter -> different draw
  st2 = torch.tensor([1234, 0], dtype=torch.int64, device=device)
  act2 = torch.empty_like(act)
  ops.categorical_sample(hd, ld, rows, A, st2, act2)
  assert torch.equal(act, act2)
  ops.categorical_sample(hd, ld, rows, A, st2, act2)
  assert not torch.equal(act, act2)
  # a peaked row always returns its mode; ParametricDistribution.sample goes through the same kernel
  from seed_rl_amd import parametric_distribution as pd
  lg = torch.full((7, 5, A), -30.0, device=device); lg[..., 11] = 30.0
  assert bool((pd.categorical_distribution(A).sample(lg) == 11).all())


def _mk_fused(device, E, T, A, cap):
  from seed_rl_amd import inference, networks, utils
  from seed_rl_amd.unroll_store import Spec
  obs_shape = (84, 84, 1)
  env_specs = utils.EnvOutput(Spec((), torch.float32), Spec((), torch.bool), Spec(obs_shape, torch.uint8),
                              Spec((), torch.bool), Spec((), torch.int32))
  ao_specs = networks.AgentOutput(Spec((), torch.int64), Spec((A,), torch.float32), Spec((), torch.float32))
  agent = networks.AtariShallow(A, device=device, seed=0)
  return inference.FusedInferenceState(agent, E, T, env_specs, ao_specs, batch_capacity=cap, device=device), obs_shape


def _req(device, ids, step, obs_shape, seed):
  from seed_rl_amd import utils
  rng = np.random.default_rng(seed)
  n = len(ids)
  env = utils.EnvOutput(
      reward=torch.tensor(rng.normal(size=n).astype(np.float32), device=device),
      done=torch.zeros(n, dtype=torch.bool, device=device),
      observation=torch.tensor(rng.integers(0, 256, (n,) + obs_shape).astype(np.uint8), device=device),
      abandoned=torch.zeros(n, dtype=torch.bool, device=device),
      episode_step=torch.full((n,), step, dtype=torch.int32, device=device))
  return torch.tensor(ids, dtype=torch.int64, device=device), torch.full((n,), 7, dtype=torch.int64, device=device), env


@pytest.mark.parametrize('T', [3, 4])
def test_fused_inference_bad_ids_are_masked_not_fatal(device, T):
  """an out-of-range or duplicate env id must not touch memory it does not own.  The offending rows are
  skipped (flags 1 / 2, check_errors raises like the reference would) and the valid rows of the same batch are
  processed exactly as without them."""
  E, A = 4, 6
  good, obs_shape = _mk_fused(device, E, T, A, 8)
  bad, _ = _mk_fused(device, E, T, A, 8)
  for step in range(T + 2):
    ids, runs, env = _req(device, [0, 2], step, obs_shape, step)
    good.inference(ids, runs, env, env.reward)
    # same two rows + one out-of-range id + one duplicate of env 2 appended (rows 2, 3)
    ids4 = torch.tensor([0, 2, 1000000, 2], dtype=torch.int64, device=device)
    env4 = env._replace(**{k: torch.cat([getattr(env, k), getattr(env, k)], 0) for k in env._fields})
    bad.inference(ids4, torch.cat([runs, runs]), env4, env4.reward)
  torch.cuda.synchronize()
  good.check_errors()
  assert int(bad.error_flag[0]) & 3 == 3
  with pytest.raises(ValueError):
    bad.check_errors()
  from seed_rl_amd import utils
  for a, b in zip(utils.flatten(good.store) + [good.store_index, good.actions_tab, good.batch_count, good.stack_valid],
                  utils.flatten(bad.store) + [bad.store_index, bad.actions_tab, bad.batch_count, bad.stack_valid]):
    if a.dim() >= 2 and a.shape[1] == E:
      assert torch.equal(a[:, [0, 1, 3]], b[:, [0, 1, 3]])          # env 2's duplicate may have won the race; others exact
    else:
      assert torch.equal(a[[0, 1, 3]], b[[0, 1, 3]]) if a.numel() == E else torch.equal(a, b)


@pytest.mark.parametrize('T', [2, 4])
def test_fused_inference_batch_overflow_keeps_store_consistent(device, T):
  """when the training batch is full (flag 8) a completed unroll is dropped, but its last step is still
  carried to slot 0 -- the env's NEXT unroll starts with the overlap step, as utils.py:237-255 prescribes."""
  E, A = 2, 6
  st, obs_shape = _mk_fused(device, E, T, A, 1)                   # room for ONE unroll; two complete at once
  last_env = None
  for step in range(T + 1):
    ids, runs, env = _req(device, [0, 1], step, obs_shape, 10 + step)
    st.inference(ids, runs, env, env.reward)
    last_env = env
  torch.cuda.synchronize()
  assert int(st.error_flag[0]) == 8 and int(st.batch_count[0]) == 1
  assert st.store_index.tolist() == [1, 1]
  # slot 0 of BOTH envs holds the last step (reward / episode_step / frames), also for the dropped one
  assert torch.equal(st.store[1].reward[0], last_env.reward)
  assert torch.equal(st.store[1].episode_step[0], last_env.episode_step)
  assert torch.equal(st.store[1].observation[0], last_env.observation)


@pytest.mark.parametrize('T', [3, 4])
def test_graphed_inference_packed_request(device, T):
  """fn.replay_packed(request bytes, frames): the transport-facing entry (two copies per batch) equals the
  structured call."""
  from seed_rl_amd import inference
  E, A, n = 4, 6, 2
  a, obs_shape = _mk_fused(device, E, T, A, 8)
  b, _ = _mk_fused(device, E, T, A, 8)
  fa, fb = a.graphed(n, obs_shape), b.graphed(n, obs_shape)
  for step in range(2 * T + 1):
    for ids_l in ([0, 2], [3, 1]):
      ids, runs, env = _req(device, ids_l, step, obs_shape, 100 * step + ids_l[0])
      raw = env.reward * 2
      act_a = fa(ids, runs, env, raw).clone()
      req = inference.pack_request(n, ids.cpu().numpy(), runs.cpu().numpy(), env.reward.cpu().numpy(),
                                   raw.cpu().numpy(), env.done.cpu().numpy(), None, env.episode_step.cpu().numpy())
      act_b = fb.replay_packed(torch.from_numpy(req).pin_memory(), env.observation).clone()
      assert torch.equal(act_a, act_b)
  a.check_errors(); b.check_errors()
  from seed_rl_amd import utils
  for x, y in zip(utils.flatten(a.batch) + [a.info_return, a.info_raw], utils.flatten(b.batch) + [b.info_return, b.info_raw]):
    assert torch.equal(x, y)


def test_stack_state_indexed_matches_gather_scatter(device):
  """csrc/frames.hip *_indexed (the frame-stacking state used in place in its per-env table, atari/networks.py:102-108,
  164-169 through learner.py:381-403) equals gather -> stack_prepare / stack_pack_state -> scatter, bit for bit,
  including restarted actors (state counts as zeros) and rows that must not be written back."""
  from seed_rl_amd import ops
  E, n, HW, T1 = 11, 6, 7056, 1
  rng = np.random.default_rng(0)
  table = torch.as_tensor(rng.integers(0, 2 ** 24, (E, HW)).astype(np.int32)).to(device)
  rows = torch.as_tensor(np.array([7, 2, 9, 0, 4, 10], np.int64)).to(device)
  zero = torch.as_tensor(np.array([0, 1, 0, 0, 1, 0], np.uint8)).to(device)
  valid = torch.as_tensor(np.array([1, 1, 0, 1, 1, 1], np.uint8)).to(device)
  done = torch.as_tensor(np.array([[0, 0, 1, 0, 0, 1]], np.uint8)).to(device)
  frames = torch.as_tensor(rng.integers(0, 256, (n, HW)).astype(np.uint8)).to(device)
  # dense path
  st = table[rows].clone()
  st[zero.bool()] = 0
  ext_a = torch.zeros((T1 + 3, n, HW), dtype=torch.uint8, device=device); ext_a[3] = frames
  nv_a = torch.zeros((T1, n), dtype=torch.uint8, device=device)
  ops.stack_prepare(st.contiguous(), done, T1, n, HW, ext_a, nv_a)
  new_a = torch.empty_like(st)
  ops.stack_pack_state(ext_a, nv_a, T1, n, HW, new_a)
  want = table.clone()
  keep = valid.bool()
  want[rows[keep]] = new_a[keep]
  # in place
  tab_b = table.clone()
  ext_b = torch.zeros_like(ext_a); ext_b[3] = frames
  nv_b = torch.zeros_like(nv_a)
  ops.stack_prepare_indexed(tab_b, rows, zero, done, T1, n, HW, ext_b, nv_b)
  assert torch.equal(ext_b, ext_a) and torch.equal(nv_b, nv_a)
  ops.stack_pack_state_indexed(ext_b, nv_b, T1, n, HW, tab_b, rows, valid)
  assert torch.equal(tab_b, want)
  # rows = None: row b, no masks
  tab_c = table[:n].clone().contiguous()
  ops.stack_prepare_indexed(tab_c, None, None, done, T1, n, HW, ext_b, nv_b)
  ops.stack_prepare(table[:n].contiguous(), done, T1, n, HW, ext_a, nv_a)
  assert torch.equal(ext_b, ext_a)


def test_dequeue_ring_wraps_and_keeps_order(device):
  """The training batch is a ring of columns (learner.py:418-432: dequeue(batch_size) takes the OLDEST unrolls): a small
  capacity that is no multiple of the batch size forces the head to wrap while inference keeps appending; the unrolls
  handed to the learner must be exactly the ones a never-dequeued, large batch collects, in the same order."""
  from seed_rl_amd import inference, learner, networks, utils
  from seed_rl_amd.unroll_store import Spec
  E, T, A, n, B = 6, 2, 5, 3, 4
  obs = (84, 84, 1)
  env_specs = utils.EnvOutput(Spec((), torch.float32), Spec((), torch.bool), Spec(obs, torch.uint8), Spec((), torch.bool),
                              Spec((), torch.int32))
  ao_specs = networks.AgentOutput(Spec((), torch.int64), Spec((A,), torch.float32), Spec((), torch.float32))

  def run(cap, dequeue):
    agent = networks.AtariShallow(A, device=device, seed=0)
    agent.seed_sampler(7)
    st = inference.FusedInferenceState(agent, E, T, env_specs, ao_specs, batch_capacity=cap, device=device)
    z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=device)
    dst = learner.Unroll(agent.initial_state(B), z((T + 1, B), torch.int64),
                         utils.EnvOutput(z((T + 1, B), torch.float32), z((T + 1, B), torch.bool),
                                         z((T + 1, B) + obs, torch.uint8), z((T + 1, B), torch.bool),
                                         z((T + 1, B), torch.int32)),
                         networks.AgentOutput(z((T + 1, B), torch.int64), z((T + 1, B, A), torch.float32),
                                              z((T + 1, B), torch.float32)))
    got = []
    rng = np.random.default_rng(0)
    for step in range(14):
      for lo in (0, n):                                      # two calls of n = 3 envs per round
        ids = torch.arange(lo, lo + n, device=device)
        env = utils.EnvOutput(
            torch.as_tensor((100 * np.arange(lo, lo + n) + step).astype(np.float32)).to(device),   # identifies (env, step)
            torch.as_tensor(rng.uniform(size=n) < 0.1).to(device),
            torch.as_tensor(rng.integers(0, 256, (n,) + obs).astype(np.uint8)).to(device),
            torch.zeros(n, dtype=torch.bool, device=device), torch.full((n,), step, dtype=torch.int32, device=device))
        st.inference(ids, torch.full((n,), 5, dtype=torch.int64, device=device), env, env.reward)
        if dequeue:
          while st.dequeue_into(dst, B):
            got.append((dst.env_outputs.reward.cpu().numpy().copy(), dst.agent_outputs.action.cpu().numpy().copy(),
                        dst.env_outputs.observation.cpu().numpy()[:, :, 0, 0, 0].copy(),
                        dst.agent_state.frame_stacking_state.cpu().numpy()[:, :3].copy()))
    st.check_errors()
    k, rest = st.take_batch()
    return got, k, rest
  got, k_left, rest = run(cap=7, dequeue=True)               # 7 columns, batches of 4: the head wraps every other dequeue
  _, k_all, full = run(cap=64, dequeue=False)
  assert len(got) >= 4 and k_all == 4 * len(got) + k_left
  rew = full.env_outputs.reward.cpu().numpy(); act = full.agent_outputs.action.cpu().numpy()
  ob = full.env_outputs.observation.cpu().numpy()[:, :, 0, 0, 0]
  fs = full.agent_state.frame_stacking_state.cpu().numpy()[:, :3]
  for j, (r, a, o, f) in enumerate(got):
    sl = slice(4 * j, 4 * j + 4)
    np.testing.assert_array_equal(r, rew[:, sl]); np.testing.assert_array_equal(a, act[:, sl])
    np.testing.assert_array_equal(o, ob[:, sl]); np.testing.assert_array_equal(f, fs[sl])
  # what is left in the wrapped ring comes out of take_batch in order too
  np.testing.assert_array_equal(rest.env_outputs.reward.cpu().numpy(), rew[:, 4 * len(got):])
