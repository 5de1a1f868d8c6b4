"""Learner-side batched inference step (central inference): mirror of the `inference` tf.function of
/root/reference/agents/vtrace/learner.py:350-405, on the device-resident store.

One call handles one inference batch (n = inference_batch_size env steps that the transport layer has
already batched -- the gRPC server with its dynamic batching, grpc/ops/grpc.cc:591-861, is out of scope):
run-id bookkeeping / resets (:353-366), episode statistics (:373-378), prev-action / agent-state reads
(:381-383), the single-step agent forward on the HIP kernels (:384-390), UnrollStore.append + completed
unrolls (:394-397), state/action updates (:398-403); returns the actions (:405).  Completed unrolls are
handed to `unroll_sink` time-major (what `unroll_queue.enqueue_many` + `dequeue` + `make_time_major`
produce in the reference, learner.py:396-397,418-432), optionally straight into a training batch.
"""
import collections

import torch

from seed_rl_amd import learner as learner_lib
from seed_rl_amd import ops, unroll_store, utils
from seed_rl_amd.unroll_store import Spec

EpisodeInfo = collections.namedtuple('EpisodeInfo', 'episode_num_frames episode_returns episode_raw_returns')


class InferenceState(object):
  """The per-host state the reference builds in create_host (learner.py:314-336)."""

  def __init__(self, agent, num_envs, unroll_length, env_output_specs, agent_output_specs, action_spec,
               num_action_repeats=1, device='cuda', unroll_sink=None, info_sink=None):
    self.agent, self.num_envs, self.unroll_length = agent, num_envs, unroll_length
    self.num_action_repeats = num_action_repeats
    self.device = torch.device(device)
    self.store = unroll_store.UnrollStore(num_envs, unroll_length,
                                          (action_spec, env_output_specs, agent_output_specs), device=device)
    self.env_run_ids = unroll_store.Aggregator(num_envs, Spec((), torch.int64), device, 'run_ids')
    info_specs = EpisodeInfo(Spec((), torch.int64), Spec((), torch.float32), Spec((), torch.float32))
    self.env_infos = unroll_store.Aggregator(num_envs, info_specs, device, 'env_infos')
    state_specs = unroll_store.specs_like(agent.initial_state(1))
    self.first_agent_states = unroll_store.Aggregator(num_envs, state_specs, device, 'first_agent_states')
    self.agent_states = unroll_store.Aggregator(num_envs, state_specs, device, 'agent_states')
    self.actions = unroll_store.Aggregator(num_envs, action_spec, device, 'actions')
    self.unroll_sink = unroll_sink or (lambda unroll: None)
    self.info_sink = info_sink or (lambda info: None)

  def inference(self, env_ids, run_ids, env_outputs, raw_rewards):
    dev = self.device
    env_ids = torch.as_tensor(env_ids, device=dev).to(torch.int64)
    run_ids = torch.as_tensor(run_ids, device=dev).to(torch.int64)
    # Reset the environments that had their first run or crashed (learner.py:353-366).
    previous_run_ids = self.env_run_ids.read(env_ids)
    self.env_run_ids.replace(env_ids, run_ids)
    need_reset = env_ids[previous_run_ids != run_ids]
    if need_reset.numel():
      self.env_infos.reset(need_reset)
      self.store.reset(need_reset)
      init = self.agent.initial_state(int(need_reset.numel()))
      self.first_agent_states.replace(need_reset, init)
      self.agent_states.replace(need_reset, init)
      self.actions.reset(need_reset)
    if env_outputs.abandoned is not None and bool(env_outputs.abandoned.any()):
      raise ValueError('Abandoned done states are not supported in VTRACE.')            # :368-370
    # Update steps and return (:373-378).
    n = env_ids.numel()
    zeros_i = torch.zeros(n, dtype=torch.int64, device=dev)
    zeros_f = torch.zeros(n, dtype=torch.float32, device=dev)
    self.env_infos.add(env_ids, EpisodeInfo(zeros_i, env_outputs.reward, raw_rewards))
    done_ids = env_ids[env_outputs.done]
    if done_ids.numel():
      self.info_sink(self.env_infos.read(done_ids))
      self.env_infos.reset(done_ids)
    self.env_infos.add(env_ids, EpisodeInfo(zeros_i + self.num_action_repeats, zeros_f, zeros_f))
    # Inference (:381-390).
    prev_actions = self.actions.read(env_ids)
    prev_agent_states = self.agent_states.read(env_ids)
    agent_outputs, curr_agent_states = self.agent(prev_actions, env_outputs, prev_agent_states, unroll=False,
                                                  is_training=False)
    agent_outputs = utils.map_structure(lambda t: t.contiguous(), agent_outputs)
    # Append to the unrolls; hand completed unrolls over (:394-399).
    store_env = env_outputs._replace(
        abandoned=env_outputs.abandoned if env_outputs.abandoned is not None else torch.zeros_like(env_outputs.done),
        episode_step=env_outputs.episode_step if env_outputs.episode_step is not None else zeros_i.to(torch.int32))
    completed_ids, unrolls = self.store.append(env_ids, (prev_actions, store_env, agent_outputs))
    if completed_ids.numel():
      unroll = learner_lib.Unroll(self.first_agent_states.read(completed_ids), *unrolls)
      self.unroll_sink(unroll)
      self.first_agent_states.replace(completed_ids, self.agent_states.read(completed_ids))
    # Update current state (:401-403).
    self.agent_states.replace(env_ids, curr_agent_states)
    self.actions.replace(env_ids, agent_outputs.action)
    return agent_outputs.action


class FusedInferenceState(object):
  """The same inference step with NO host synchronisation and static shapes, so that one call is ~50 kernel
  launches that can be captured once in a HIP graph and replayed (`graphed(n)`).

  Differences from `InferenceState` (which follows the reference op by op and is kept as the executable
  specification this class is tested against):
    * data-dependent subsets (envs needing reset, finished episodes, completed unrolls) are handled with per-row
      masks and device-side scans (csrc/inference.hip) instead of boolean-mask gathers;
    * completed unrolls are written time-major straight into a training batch of `batch_capacity` columns
      (`self.batch`, an `Unroll` of [T+1, capacity, ...] tensors + first agent states [capacity, ...]); `batch_count`
      counts the filled columns on the device; `take_batch()` hands the filled part over (one host read);
    * finished episodes go to a device ring `episode_stats[capacity, 3]` = (frames, return, raw return);
    * unroll overlap 0 only (the V-trace learner; R2D2's burn-in overlap uses `InferenceState`).
  """

  def __init__(self, agent, num_envs, unroll_length, env_output_specs, agent_output_specs, batch_capacity,
               num_action_repeats=1, device='cuda', stats_capacity=4096):
    self.agent, self.E, self.L = agent, num_envs, unroll_length + 1
    self.cap, self.num_action_repeats = batch_capacity, num_action_repeats
    self.device = dev = torch.device(device)
    z64 = lambda: torch.zeros(num_envs, dtype=torch.int64, device=dev)
    self.run_ids_tab, self.info_frames, self.actions_tab, self.store_index = z64(), z64(), z64(), z64()
    self.info_return = torch.zeros(num_envs, dtype=torch.float32, device=dev)
    self.info_raw = torch.zeros(num_envs, dtype=torch.float32, device=dev)
    field_specs = (Spec((), torch.int64), env_output_specs, agent_output_specs)
    mk = lambda lead: unroll_store._map_specs(
        lambda s: torch.zeros(lead + tuple(s.shape), dtype=s.dtype, device=dev), field_specs)
    self.store = mk((self.L, num_envs))                       # time-major [T+1, num_envs, ...]
    fields = mk((self.L, batch_capacity))
    state_specs = unroll_store.specs_like(agent.initial_state(1))
    mks = lambda lead: unroll_store._map_specs(
        lambda s: torch.zeros((lead,) + tuple(s.shape), dtype=s.dtype, device=dev), state_specs)
    self.first_agent_states, self.agent_states = mks(num_envs), mks(num_envs)
    self.batch = learner_lib.Unroll(mks(batch_capacity), *fields)
    self.batch_count = torch.zeros(1, dtype=torch.int32, device=dev)
    self.episode_stats = torch.zeros((stats_capacity, 3), dtype=torch.float32, device=dev)
    self.stats_count = torch.zeros(1, dtype=torch.int32, device=dev)
    self.error_flag = torch.zeros(1, dtype=torch.int32, device=dev)
    self._scratch = {}

  def _bufs(self, n):
    b = self._scratch.get(n)
    if b is None:
      dev, L = self.device, self.L
      i64 = lambda k: torch.zeros(k, dtype=torch.int64, device=dev)
      u8 = lambda k: torch.zeros(k, dtype=torch.uint8, device=dev)
      b = dict(reset=u8(n), prev_actions=i64(n), append_rows=i64(n), complete=u8(n), cols=i64(n), gsrc=i64(L * n),
               gdst=i64(L * n), gmask=u8(L * n), last=i64(n))
      self._scratch[n] = b
    return b

  @staticmethod
  def _rb(t, lead):
    return unroll_store._row_bytes(t, lead)

  def inference(self, env_ids, run_ids, env_outputs, raw_rewards):
    """learner.py:350-405.  env_ids must be unique within a call (flagged on the device otherwise: check_errors)."""
    dev = self.device
    ids = torch.as_tensor(env_ids, device=dev).to(torch.int64).contiguous()
    runs = torch.as_tensor(run_ids, device=dev).to(torch.int64).contiguous()
    n = ids.numel()
    b = self._bufs(n)
    reward = env_outputs.reward.to(torch.float32).contiguous()
    done_u8 = ops.as_u8(env_outputs.done)
    ops.inference_pre(ids, runs, reward, raw_rewards.to(torch.float32).contiguous(), done_u8, n, self.E,
                      self.num_action_repeats, self.run_ids_tab, self.info_frames, self.info_return, self.info_raw,
                      self.actions_tab, self.store_index, b['reset'], b['prev_actions'], self.episode_stats,
                      self.stats_count, self.error_flag)
    # previous agent state (zeros for envs whose actor restarted), first-state table reset (:363-365, :382-383)
    tabs = utils.flatten(self.agent_states)
    prev_leaves = [torch.empty((n,) + tuple(t.shape[1:]), dtype=t.dtype, device=dev) for t in tabs]
    if tabs:
      srb0 = [self._rb(t, 1) for t in tabs]
      ops.rows_move_multi(prev_leaves, tabs, srb0, None, ids, n, b['reset'], zero_where_masked=True)
      firsts0 = utils.flatten(self.first_agent_states)
      ops.rows_move_multi(firsts0, [None] * len(firsts0), srb0, ids, None, n, b['reset'])
    it = iter(prev_leaves)
    prev_state = utils.map_structure(lambda t: next(it), self.agent_states)
    # single-step agent forward (:384-390)
    agent_outputs, curr_state = self.agent(b['prev_actions'], env_outputs, prev_state, unroll=False, is_training=False)
    agent_outputs = utils.map_structure(lambda t: t.contiguous(), agent_outputs)
    ops.inference_post(ids, agent_outputs.action.to(torch.int64).contiguous(), n, self.E, self.L, self.cap,
                       self.store_index, self.actions_tab, self.batch_count, b['append_rows'], b['complete'], b['cols'],
                       b['gsrc'], b['gdst'], b['gmask'], b['last'], self.error_flag)
    zeros_b = torch.zeros_like(env_outputs.done)
    store_env = env_outputs._replace(
        abandoned=env_outputs.abandoned if env_outputs.abandoned is not None else zeros_b,
        episode_step=env_outputs.episode_step if env_outputs.episode_step is not None else zeros_b.to(torch.int32))
    values = (b['prev_actions'], store_env, agent_outputs)
    batch_fields = (self.batch.prev_actions, self.batch.env_outputs, self.batch.agent_outputs)
    stores = utils.flatten(self.store)
    vals = [v.to(s.dtype).contiguous() for s, v in zip(stores, utils.flatten(values))]
    outs = utils.flatten(batch_fields)
    rbs = [self._rb(s, 2) for s in stores]
    ops.rows_move_multi(stores, vals, rbs, b['append_rows'], None, n)                          # :394 append
    ops.rows_move_multi(outs, stores, rbs, b['gdst'], b['gsrc'], self.L * n, b['gmask'])       # completed unrolls -> batch
    ops.rows_move_multi(stores, stores, rbs, ids, b['last'], n, b['complete'])                 # carry the last step
    firsts, prevs = utils.flatten(self.first_agent_states), utils.flatten(prev_state)
    if firsts:
      srb = [self._rb(t, 1) for t in firsts]
      ops.rows_move_multi(utils.flatten(self.batch.agent_state), firsts, srb, b['cols'], ids, n, b['complete'])  # :396
      ops.rows_move_multi(firsts, prevs, srb, ids, None, n, b['complete'])                     # :398-399
      ops.rows_move_multi(utils.flatten(self.agent_states), [c.contiguous() for c in utils.flatten(curr_state)], srb,
                          ids, None, n)                                                        # :401
    return agent_outputs.action

  def graphed(self, n, observation_shape, warmup=3):
    """Captures one inference call for batch size n in a HIP graph.  Returns fn(env_ids, run_ids, env_outputs,
    raw_rewards) -> actions that copies its arguments into the graph's static inputs and replays it."""
    dev = self.device
    si = dict(ids=torch.zeros(n, dtype=torch.int64, device=dev), runs=torch.zeros(n, dtype=torch.int64, device=dev),
              raw=torch.zeros(n, device=dev))
    senv = utils.EnvOutput(torch.zeros(n, device=dev), torch.zeros(n, dtype=torch.bool, device=dev),
                           torch.zeros((n,) + tuple(observation_shape), dtype=torch.uint8, device=dev),
                           torch.zeros(n, dtype=torch.bool, device=dev), torch.zeros(n, dtype=torch.int32, device=dev))
    saved = [t.clone() for t in self._state_tensors()]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    si['ids'].copy_(torch.arange(n, device=dev))
    with torch.cuda.stream(side):
      for _ in range(warmup):
        self.inference(si['ids'], si['runs'], senv, si['raw'])
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, capture_error_mode='relaxed'):
      actions = self.inference(si['ids'], si['runs'], senv, si['raw'])
    for t, s in zip(self._state_tensors(), saved):          # warm-up calls must not leave traces in the tables
      t.copy_(s)

    def fn(env_ids, run_ids, env_outputs, raw_rewards):
      si['ids'].copy_(torch.as_tensor(env_ids, device=dev)); si['runs'].copy_(torch.as_tensor(run_ids, device=dev))
      si['raw'].copy_(raw_rewards)
      senv.reward.copy_(env_outputs.reward); senv.done.copy_(env_outputs.done)
      senv.observation.copy_(env_outputs.observation)
      if env_outputs.abandoned is not None:
        senv.abandoned.copy_(env_outputs.abandoned)
      if env_outputs.episode_step is not None:
        senv.episode_step.copy_(env_outputs.episode_step)
      graph.replay()
      return actions
    fn.graph, fn.static_inputs, fn.static_env = graph, si, senv
    return fn

  def _state_tensors(self):
    return ([self.run_ids_tab, self.info_frames, self.actions_tab, self.store_index, self.info_return, self.info_raw,
             self.batch_count, self.stats_count, self.error_flag, self.episode_stats] +
            utils.flatten(self.store) + utils.flatten(self.first_agent_states) + utils.flatten(self.agent_states) +
            utils.flatten(self.batch))

  def check_errors(self):
    f = int(self.error_flag[0])
    if f:
      raise ValueError('inference bookkeeping error flags %d (1 id out of range, 2 duplicate ids, 4 store overflow, '
                       '8 training batch overflow)' % f)

  def take_batch(self):
    """Host read of the fill count; returns (count, Unroll of views over the filled columns) and restarts filling."""
    k = int(self.batch_count[0])
    first = utils.map_structure(lambda t: t[:k], self.batch.agent_state)
    rest = utils.map_structure(lambda t: t[:, :k], (self.batch.prev_actions, self.batch.env_outputs,
                                                    self.batch.agent_outputs))
    self.batch_count.zero_()
    return k, learner_lib.Unroll(first, *rest)
