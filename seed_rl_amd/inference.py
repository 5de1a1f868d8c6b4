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
import ctypes
import os

import torch

from seed_rl_amd import learner as learner_lib
from seed_rl_amd import _lib, ops, unroll_store, utils
from seed_rl_amd.unroll_store import Spec

EpisodeInfo = collections.namedtuple('EpisodeInfo', 'episode_num_frames episode_returns episode_raw_returns')


def request_layout(n):
  """Byte layout of one packed inference request batch of n env steps (structure of arrays): name -> (offset, bytes);
  'bytes' = total (16-byte multiple).  The fields are the per-step scalars of the reference's inference call
  (env_id, run_id, EnvOutput.reward/done/abandoned/episode_step, raw reward; learner.py:350-352)."""
  lay, off = {}, 0
  for name, width in (('ids', 8), ('runs', 8), ('reward', 4), ('raw', 4), ('episode_step', 4), ('done', 1),
                      ('abandoned', 1)):
    lay[name] = (off, width * n)
    off += width * n
  lay['bytes'] = (off + 15) // 16 * 16
  return lay


def pack_request(n, env_ids, run_ids, reward, raw_reward, done, abandoned=None, episode_step=None, out=None):
  """Host-side packing of a request batch into `request_layout(n)` (numpy in, uint8 numpy out)."""
  import numpy as np
  lay = request_layout(n)
  buf = np.zeros(lay['bytes'], np.uint8) if out is None else out
  def put(name, arr, dt):
    o, nb = lay[name]
    buf[o:o + nb] = np.ascontiguousarray(np.asarray(arr).astype(dt, copy=False)).view(np.uint8).reshape(-1)
  put('ids', env_ids, np.int64); put('runs', run_ids, np.int64)
  put('reward', reward, np.float32); put('raw', raw_reward, np.float32)
  put('episode_step', np.zeros(n, np.int32) if episode_step is None else episode_step, np.int32)
  put('done', done, np.uint8)
  put('abandoned', np.zeros(n, np.uint8) if abandoned is None else abandoned, np.uint8)
  return buf


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
  """The same inference step with NO host synchronisation and static shapes, so that one call is ~15 kernel
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
    self._obs_dtype = env_output_specs.observation.dtype      # uint8 frames, uint16 bit planes (GFootball), float32 vectors
    mk = lambda lead: unroll_store._map_specs(
        lambda s: torch.zeros(lead + tuple(s.shape), dtype=s.dtype, device=dev), field_specs)
    self.store = mk((self.L, num_envs))                       # time-major [T+1, num_envs, ...]
    fields = mk((self.L, batch_capacity))
    state_specs = unroll_store.specs_like(agent.initial_state(1))
    mks = lambda lead: unroll_store._map_specs(
        lambda s: torch.zeros((lead,) + tuple(s.shape), dtype=s.dtype, device=dev), state_specs)
    # Agents whose only recurrent state is the frame stack (AtariShallow) are served in six launches by
    # csrc/servestep.hip: the stack is read from the store's own observation field, the bit-packed per-env state table
    # does not exist (only the first state of each env's CURRENT unroll is kept, packed when an unroll completes).
    # SEEDHIP_SERVE_STEP=0: the generic sixteen-launch path below (any agent).
    obs_spec = env_output_specs.observation
    self._serve = (os.environ.get('SEEDHIP_SERVE_STEP', '1') != '0' and
                   getattr(agent, 'serve_step_supported', None) is not None and agent.serve_step_supported(self.L) and
                   obs_spec.dtype == torch.uint8 and len(obs_spec.shape) == 3 and obs_spec.shape[2] == 1 and
                   len(utils.flatten(agent.initial_state(1))) == 1 and batch_capacity >= 1)
    self.first_agent_states = mks(num_envs)
    self.agent_states = None if self._serve else mks(num_envs)
    self.stack_valid = torch.zeros(num_envs, dtype=torch.uint8, device=dev)    # serve path: frames of the stack in the episode
    self.first_zero = torch.zeros(num_envs, dtype=torch.uint8, device=dev)     # serve path: unroll starts from the zero state
    self.batch = learner_lib.Unroll(mks(batch_capacity), *fields)
    self.batch_count = torch.zeros(1, dtype=torch.int32, device=dev)
    # the batch is a RING of columns: head on the device (read by inference_post) and mirrored on the host (it only
    # moves when the host dequeues), fill = batch_count
    self.batch_start = torch.zeros(1, dtype=torch.int32, device=dev)
    self._start_host = 0
    self._deq_rows = {}
    self.episode_stats = torch.zeros((stats_capacity, 3), dtype=torch.float32, device=dev)
    self.stats_count = torch.zeros(1, dtype=torch.int32, device=dev)
    self.error_flag = torch.zeros(1, dtype=torch.int32, device=dev)
    self.stamp_tab = torch.zeros(num_envs, dtype=torch.int32, device=dev)      # duplicate-id detection (inference_pre)
    # the frame-stacking state (28 KB per Atari env) is used IN PLACE in its table by agents that support it: no gather
    # into a scratch before the forward, no scatter back after it
    self._frame_leaf = None
    if not self._serve and getattr(agent, 'accepts_indexed_frame_state', False) and hasattr(self.agent_states, '_fields') and \
        'frame_stacking_state' in self.agent_states._fields:
      leaves = utils.flatten(self.agent_states)
      fs = self.agent_states.frame_stacking_state
      if torch.is_tensor(fs) and fs.dtype == torch.int32 and fs.shape[1] % 4 == 0:
        self._frame_leaf = [k for k, t in enumerate(leaves) if t is fs][0]
    self.call_counter = torch.zeros(1, dtype=torch.int32, device=dev)
    self._scratch = {}

  def _bufs(self, n):
    b = self._scratch.get(n)
    if b is None:
      dev, L = self.device, self.L
      i64 = lambda k: torch.zeros(k, dtype=torch.int64, device=dev)
      u8 = lambda k: torch.zeros(k, dtype=torch.uint8, device=dev)
      b = dict(reset=u8(n), prev_actions=i64(n), append_rows=i64(n), complete=u8(n), carry=u8(n), cols=i64(n),
               emit_env=i64(n), emit_col=i64(n), emit_count=torch.zeros(1, dtype=torch.int32, device=dev),
               last=i64(n), ids_safe=i64(n), valid=u8(n),
               will_complete=u8(n),
               actions=i64(n), zeros_bool=torch.zeros(n, dtype=torch.bool, device=dev),
               zeros_i32=torch.zeros(n, dtype=torch.int32, device=dev))
      if self._serve:
        b.update(hist_rows=i64(4 * n), nvalid=u8(n), prev_valid=u8(n),
                 emit_row=torch.zeros(n, dtype=torch.int32, device=dev), rng_snapshot=i64(2))
        b['step'], b['fields'] = self._serve_structs(n, b)
      else:
        tabs = utils.flatten(self.agent_states)
        b['prev_state'] = [torch.empty((n,) + tuple(t.shape[1:]), dtype=t.dtype, device=dev) for t in tabs]
      self._scratch[n] = b
    return b

  @staticmethod
  def _rb(t, lead):
    return unroll_store._row_bytes(t, lead)

  def inference(self, env_ids, run_ids, env_outputs, raw_rewards):
    """learner.py:350-405.  env_ids must be unique and in range within a call: offending rows are skipped and flagged
    on the device (check_errors) -- the reference raises.

    Launches: inference_pre, ONE row-mover launch (previous states), the agent's single-step forward (5-6 kernels),
    inference_post (samples the actions from the head rows), two row-mover launches (append + state tables; carry-over)
    and the emission of the completed unrolls from inference_post's compact list -- no eager tensor op in between."""
    dev = self.device
    ids = torch.as_tensor(env_ids, device=dev).to(torch.int64).contiguous()
    runs = torch.as_tensor(run_ids, device=dev).to(torch.int64).contiguous()
    n = ids.numel()
    b = self._bufs(n)
    if self._serve:
      return self._inference_serve(ids, runs, env_outputs, raw_rewards, n, b)
    op = ops.row_op
    reward = env_outputs.reward.to(torch.float32).contiguous()
    done_u8 = ops.as_u8(env_outputs.done)
    ops.inference_pre(ids, runs, reward, raw_rewards.to(torch.float32).contiguous(), done_u8, n, self.E,
                      self.num_action_repeats, self.run_ids_tab, self.info_frames, self.info_return, self.info_raw,
                      self.actions_tab, self.store_index, b['reset'], b['prev_actions'], self.episode_stats,
                      self.stats_count, self.error_flag, b['ids_safe'], b['valid'], self.stamp_tab, self.call_counter,
                      will_complete=b['will_complete'], full_length=self.L)
    sid = b['ids_safe']
    # previous agent state (zeros for envs whose actor restarted), first-state table reset (:363-365, :382-383)
    tabs = utils.flatten(self.agent_states)
    firsts = utils.flatten(self.first_agent_states)
    prev_leaves = b['prev_state']
    srb = [self._rb(t, 1) for t in tabs]
    fi = self._frame_leaf
    # the in-place frame state is only set aside for the envs whose unroll completes with this step (their previous
    # state becomes the next unroll's first state, :398-399); everything else of it never leaves the table
    ops.rows_move_ops(
        [op(p, t, rb, n, src_rows=sid, mask=b['will_complete']) if k == fi else
         op(p, t, rb, n, src_rows=sid, mask=b['reset'], zero_where_masked=True)
         for k, (p, t, rb) in enumerate(zip(prev_leaves, tabs, srb))] +
        [op(f, None, rb, n, dst_rows=sid, mask=b['reset']) for f, rb in zip(firsts, srb)])
    from seed_rl_amd.networks import IndexedFrameState
    agent_leaves = list(prev_leaves)
    if fi is not None:
      agent_leaves[fi] = IndexedFrameState(tabs[fi], sid, b['reset'], b['valid'])
    it = iter(agent_leaves)
    prev_state = utils.map_structure(lambda t: next(it), self.agent_states)
    # single-step agent forward (:384-390); the action is sampled by inference_post from the head rows
    fused_sampling = getattr(self.agent, 'accepts_sample_actions', False)
    kw = dict(sample_actions=False) if fused_sampling else {}
    agent_outputs, curr_state = self.agent(b['prev_actions'], env_outputs, prev_state, unroll=False, is_training=False,
                                           **kw)
    if fused_sampling:
      head, _, ldh = self.agent.head_buffers()
      A = agent_outputs.policy_logits.shape[-1]
      logits_src, rng = head, self.agent.rng_state()
    else:
      head, ldh, A, logits_src, rng = None, 0, 0, None, None
      b['actions'].copy_(agent_outputs.action)
    ops.inference_post(sid, b['valid'], b['actions'], logits_src, ldh, A, rng, n, self.E, self.L, self.cap,
                       self.store_index, self.actions_tab, self.batch_count, b['append_rows'], b['complete'], b['carry'],
                       b['cols'], b['emit_env'], b['emit_col'], b['emit_count'], b['last'], self.error_flag,
                       batch_start=self.batch_start)
    store_env = env_outputs._replace(
        abandoned=env_outputs.abandoned if env_outputs.abandoned is not None else b['zeros_bool'],
        episode_step=env_outputs.episode_step if env_outputs.episode_step is not None else b['zeros_i32'])
    values = (b['prev_actions'], store_env, agent_outputs._replace(action=b['actions']))
    batch_fields = (self.batch.prev_actions, self.batch.env_outputs, self.batch.agent_outputs)
    stores = utils.flatten(self.store)
    outs = utils.flatten(batch_fields)
    rbs = [self._rb(s_, 2) for s_ in stores]
    # the step's values, read where they are: the head-GEMM output row [logits | baseline | pad] is a strided source
    srcs = []
    for s_, v in zip(stores, utils.flatten(values)):
      if v.dtype != s_.dtype:
        v = v.to(s_.dtype)
      if fused_sampling and v.data_ptr() >= head.data_ptr() and \
          v.data_ptr() < head.data_ptr() + head.numel() * 4 and not v.is_contiguous():
        srcs.append((v, ldh * 4))                               # a column slice of the head buffer
      else:
        srcs.append((v.contiguous(), 0))
    def leaves_of(struct):                         # flatten that keeps an in-place frame state as ONE leaf
      if isinstance(struct, IndexedFrameState) or not isinstance(struct, (tuple, list)):
        return [struct]
      return [x for part in struct for x in leaves_of(part)]
    prevs = list(prev_leaves)
    currs = [c if isinstance(c, IndexedFrameState) else c.contiguous() for c in leaves_of(curr_state)]
    ops.rows_move_ops(
        [op(s_, v, rb, n, dst_rows=b['append_rows'], mask=b['valid'], src_pitch=sp)                 # :394 append
         for s_, (v, sp), rb in zip(stores, srcs, rbs)] +
        [op(o, f, rb, n, dst_rows=b['cols'], src_rows=sid, mask=b['complete'])                      # :396 first states
         for o, f, rb in zip(utils.flatten(self.batch.agent_state), firsts, srb)] +
        [op(t, c, rb, n, dst_rows=sid, mask=b['valid'])
         for k, (t, c, rb) in enumerate(zip(tabs, currs, srb)) if k != fi])                         # :401 (fi: done in place)
    ops.emit_unrolls(outs, stores, rbs, b['emit_env'], b['emit_col'], b['emit_count'], n, self.L, self.E,
                     self.cap)                                                                     # completed unrolls -> batch
    ops.rows_move_ops(
        [op(s_, s_, rb, n, dst_rows=sid, src_rows=b['last'], mask=b['carry']) for s_, rb in zip(stores, rbs)] +   # carry
        [op(f, p, rb, n, dst_rows=sid, mask=b['carry']) for f, p, rb in zip(firsts, prevs, srb)])   # :398-399
    return b['actions']

  def _serve_structs(self, n, b):
    """The seedhip_serve_step / seedhip_serve_fields of batch size n: every table and scratch pointer is static; the
    request pointers are filled in per call."""
    p = lambda t: t.data_ptr()
    st = _lib.ServeStep()
    st.n, st.num_envs, st.num_action_repeats = n, self.E, self.num_action_repeats
    st.full_length, st.batch_capacity = self.L, self.cap
    st.run_ids_table, st.info_frames, st.info_return, st.info_raw_return = (
        p(self.run_ids_tab), p(self.info_frames), p(self.info_return), p(self.info_raw))
    st.actions_table, st.store_index = p(self.actions_tab), p(self.store_index)
    st.stack_valid, st.first_zero = p(self.stack_valid), p(self.first_zero)
    st.stamp_table, st.call_counter = p(self.stamp_tab), p(self.call_counter)
    st.episode_stats, st.stats_capacity, st.stats_count = p(self.episode_stats), self.episode_stats.shape[0], p(self.stats_count)
    st.error_flag, st.batch_count, st.batch_start = p(self.error_flag), p(self.batch_count), p(self.batch_start)
    st.rng_state = p(self.agent.rng_state())
    st.ids_safe, st.valid, st.prev_actions, st.append_rows = p(b['ids_safe']), p(b['valid']), p(b['prev_actions']), p(b['append_rows'])
    st.hist_rows, st.nvalid, st.prev_valid = p(b['hist_rows']), p(b['nvalid']), p(b['prev_valid'])
    st.emit_env, st.emit_col, st.emit_row, st.emit_count = p(b['emit_env']), p(b['emit_col']), p(b['emit_row']), p(b['emit_count'])
    st.rng_snapshot = p(b['rng_snapshot'])
    prev, env, ao = self.store
    f = _lib.ServeFields()
    f.prev_actions, f.reward, f.done, f.abandoned, f.episode_step = p(prev), p(env.reward), p(env.done), p(env.abandoned), p(env.episode_step)
    f.action, f.policy_logits, f.baseline = p(ao.action), p(ao.policy_logits), p(ao.baseline)
    return st, f

  def _inference_serve(self, ids, runs, env_outputs, raw_rewards, n, b):
    """learner.py:350-405 in SIX launches (csrc/servestep.hip): serve_begin, the first conv from the request frames +
    the store's history, the second conv, the Dense partial sums, serve_finish (Dense epilogue + heads + sampling +
    append of the step's fields + action table) and serve_emit (completed unrolls -> training batch,
    carry, first agent state) -- no row mover, no unpack / re-pack of the stacking state."""
    st, fields = b['step'], b['fields']
    reward = env_outputs.reward.to(torch.float32).contiguous()
    raw = raw_rewards.to(torch.float32).contiguous()
    done_u8 = ops.as_u8(env_outputs.done).contiguous()
    obs = env_outputs.observation
    if obs.dtype != torch.uint8 or not obs.is_contiguous():
      raise ValueError('observations must be contiguous uint8 frames')
    ab = None if env_outputs.abandoned is None else ops.as_u8(env_outputs.abandoned).contiguous()
    es = None if env_outputs.episode_step is None else env_outputs.episode_step.to(torch.int32).contiguous()
    st.env_ids, st.run_ids, st.reward, st.raw_reward, st.done = (ids.data_ptr(), runs.data_ptr(), reward.data_ptr(),
                                                                 raw.data_ptr(), done_u8.data_ptr())
    st.abandoned = None if ab is None else ab.data_ptr()
    st.episode_step = None if es is None else es.data_ptr()
    agent = self.agent
    st.rng_state = agent.rng_state().data_ptr()
    ops.serve_begin(st, *agent.serve_begin_weights(), like=ids)
    store_obs = self.store[1].observation
    hw = self._rb(store_obs, 2)
    part, slices, fc_b, feat, himg, hb_, ldh, A = agent.serve_forward(n, obs, store_obs, b['hist_rows'], b['nvalid'])
    ops.serve_finish(st, fields, part, slices, fc_b, feat, himg, hb_, ldh, A, b['actions'], obs, store_obs, hw)
    stores = utils.flatten(self.store)
    outs = utils.flatten((self.batch.prev_actions, self.batch.env_outputs, self.batch.agent_outputs))
    rbs = [self._rb(s_, 2) for s_ in stores]
    ops.serve_emit(st, outs, stores, rbs, utils.flatten(self.first_agent_states)[0],
                   utils.flatten(self.batch.agent_state)[0], store_obs, hw)
    return b['actions']

  def graphed(self, n, observation_shape, warmup=3, input_slot=0):
    """Captures one inference call for batch size n in a HIP graph.  Returns fn(env_ids, run_ids, env_outputs,
    raw_rewards) -> actions that copies its arguments into the graph's static inputs and replays it.

    The static inputs are laid out for a transport layer: every per-request scalar lives in ONE byte buffer
    (`fn.request`, layout `request_layout(n)`: ids i64 | run ids i64 | reward f32 | raw reward f32 | episode_step i32 |
    done u8 | abandoned u8), so a front-end that batches requests in pinned host memory hands a batch over with two
    copies -- the packed scalars and the frames -- through `fn.replay_packed(request, observation)`; for the Atari
    agents the frames land directly in the agent's frame buffer (no device-side copy before the first conv).

    input_slot: graphs captured with different slots have DIFFERENT static inputs (their own request bytes and, for
    the Atari agents, their own frame buffer), so the host->device copies of batch i+1 can run on a copy stream while
    the graph of batch i executes: `fn.stage(request, observation)` does the two copies on the current stream,
    `fn.launch()` replays (replay_packed = both, on one stream)."""
    dev = self.device
    prev_slot = getattr(self.agent, 'frames_slot', 0)
    if hasattr(self.agent, 'frames_buffer'):
      self.agent.frames_slot = input_slot            # read by the agent's forward while it is warmed up / captured
    lay = request_layout(n)
    req = torch.zeros(lay['bytes'], dtype=torch.uint8, device=dev)
    view = lambda k, dt: req[lay[k][0]:lay[k][0] + lay[k][1]].view(dt)
    si = dict(ids=view('ids', torch.int64), runs=view('runs', torch.int64), raw=view('raw', torch.float32))
    if hasattr(self.agent, 'frames_buffer') and len(observation_shape) == 3 and observation_shape[2] == 1 and \
        self._obs_dtype == torch.uint8:
      obs = self.agent.frames_buffer(1, n)[3:].view((n,) + tuple(observation_shape))   # what the first conv reads
    else:
      obs = torch.zeros((n,) + tuple(observation_shape), dtype=self._obs_dtype, device=dev)
    senv = utils.EnvOutput(view('reward', torch.float32), view('done', torch.bool), obs,
                           view('abandoned', torch.bool), view('episode_step', torch.int32))
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
    # a capture stream of its own: per-stream scratch (ops._splitk_ws) is then not shared with graphs captured on
    # torch's default capture stream (the train step), so the two can replay concurrently
    if getattr(self, '_capture_stream', None) is None:
      self._capture_stream = torch.cuda.Stream(device=dev)
    with torch.cuda.graph(graph, stream=self._capture_stream, capture_error_mode='relaxed'):
      actions = self.inference(si['ids'], si['runs'], senv, si['raw'])
    for t, s in zip(self._state_tensors(), saved):          # warm-up calls must not leave traces in the tables
      t.copy_(s)
    if hasattr(self.agent, 'frames_buffer'):
      self.agent.frames_slot = prev_slot

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

    def stage(request, observation):
      """request: uint8[request_layout(n)['bytes']] (host pinned or device), observation [n, ...]: the two copies into
      this graph's static inputs, on the current stream."""
      req.copy_(request, non_blocking=True)
      senv.observation.copy_(observation, non_blocking=True)

    def launch():
      graph.replay()
      return actions

    def replay_packed(request, observation):
      stage(request, observation)
      return launch()
    fn.graph, fn.static_inputs, fn.static_env, fn.request, fn.replay_packed = graph, si, senv, req, replay_packed
    fn.stage, fn.launch, fn.actions = stage, launch, actions
    return fn

  def _state_tensors(self):
    rng = [self.agent.rng_state()] if hasattr(self.agent, 'rng_state') else []
    return ([self.run_ids_tab, self.info_frames, self.actions_tab, self.store_index, self.info_return, self.info_raw,
             self.batch_count, self.batch_start, self.stats_count, self.error_flag, self.episode_stats, self.stamp_tab,
             self.call_counter] + rng +
            utils.flatten(self.store) + utils.flatten(self.first_agent_states) +
            (utils.flatten(self.agent_states) if self.agent_states is not None else [self.stack_valid, self.first_zero]) +
            utils.flatten(self.batch))

  def check_errors(self):
    f = int(self.error_flag[0])
    if f:
      raise ValueError('inference bookkeeping error flags %d (1 id out of range, 2 duplicate ids, 4 store overflow, '
                       '8 training batch overflow)' % f)

  def dequeue_into(self, dst, batch_size):
    """unroll_queue.dequeue(batch_size) + make_time_major of the reference (learner.py:418-432) without either: copies
    the first `batch_size` completed unrolls -- already time-major -- from the device batch into the learner's STATIC
    training unroll `dst` (an Unroll of [T+1, batch_size, ...] tensors + first agent states [batch_size, ...]; the input
    of a captured GraphedStep) and advances the ring head: the batch is a ring of columns, nothing is compacted.  Returns
    False (nothing copied) when fewer than batch_size unrolls are complete.  One host read of the fill count.
    Call it on the stream the inference calls are submitted to (`with torch.cuda.stream(s)`: the read then orders after
    every inference call in flight) and under the lock that serialises those submissions.  That read BLOCKS until the
    replays already submitted to the stream are through (up to `inference_slots` of them: hundreds of microseconds, not
    the copies' few): it is what makes the count exact -- the host mirror the caller gates on (BatchGate.fill) is only
    a lower bound of what is complete and an upper bound would be needed for back-pressure at the same time.  Callers
    therefore attempt it only when the mirror already says a batch is there (LearnerServer.train_step)."""
    k = int(self.batch_count[0])
    self.last_fill = k                               # exact fill after this call
    if k < batch_size:
      return False
    self.last_fill = k - batch_size
    self._dequeue_copy(dst, batch_size)
    self.batch_count.fill_(k - batch_size)
    return True

  def _dequeue_copy(self, dst, batch_size):
    B, cap, L = batch_size, self.cap, self.L
    start = self._start_host
    # columns [start, start + B) of the ring, time-major: one row move per field (the strided torch copies it replaces
    # ran at a quarter of the HBM rate, and the tail of the batch had to be cloned and shifted to the front)
    key = (B, start)
    idx = self._deq_rows.get(key)
    if idx is None:
      cols = (torch.arange(B, dtype=torch.int64, device=self.device) + start) % cap
      rows = (torch.arange(L, dtype=torch.int64, device=self.device)[:, None] * cap + cols[None, :]).reshape(-1).contiguous()
      idx = (cols.contiguous(), rows)
      if len(self._deq_rows) < 64:
        self._deq_rows[key] = idx
    cols, rows = idx
    ds, ss = utils.flatten(dst.agent_state), utils.flatten(self.batch.agent_state)
    if ds:
      ops.rows_move_multi(ds, ss, [self._rb(t, 1) for t in ss], None, cols, B)
    rest_d = (dst.prev_actions, dst.env_outputs, dst.agent_outputs)
    rest_s = (self.batch.prev_actions, self.batch.env_outputs, self.batch.agent_outputs)
    fd, fs = [], []
    for d, s_ in zip(utils.flatten(rest_d), utils.flatten(rest_s)):
      if d is None:
        continue
      if d.dtype != s_.dtype or not d.is_contiguous():
        d.copy_(s_.reshape((L * cap,) + tuple(s_.shape[2:]))[rows].reshape(d.shape).to(d.dtype))
        continue
      fd.append(d); fs.append(s_)
    ops.rows_move_multi(fd, fs, [self._rb(t, 2) for t in fs], None, rows, L * B)
    self._start_host = (start + B) % cap
    self.batch_start.fill_(self._start_host)

  def dequeue_async(self, dst, batch_size):
    """dequeue_into WITHOUT the host read: the caller knows that batch_size unrolls are complete (learner_server.BatchGate
    keeps a lower bound of the count from the per-batch mirrors) -- copies columns [head, head + batch_size) of the ring
    into `dst`, advances the head and subtracts batch_size from the device count, all enqueued on the current stream (the
    inference stream, under the submission lock).  Nothing waits for the replays in flight."""
    self._dequeue_copy(dst, batch_size)
    self.batch_count.sub_(batch_size)

  def take_batch(self):
    """Host read of the fill count; returns (count, Unroll over the filled columns, oldest first) and restarts filling
    at column 0.  Views when the ring head is at 0 (nothing was dequeued since the last restart), a gather otherwise."""
    k = int(self.batch_count[0])
    start, cap = self._start_host, self.cap
    if start == 0:
      first = utils.map_structure(lambda t: t[:k], self.batch.agent_state)
      rest = utils.map_structure(lambda t: t[:, :k], (self.batch.prev_actions, self.batch.env_outputs,
                                                      self.batch.agent_outputs))
    else:
      cols = (torch.arange(k, dtype=torch.int64, device=self.device) + start) % cap
      first = utils.map_structure(lambda t: t[cols], self.batch.agent_state)
      rest = utils.map_structure(lambda t: t[:, cols], (self.batch.prev_actions, self.batch.env_outputs,
                                                        self.batch.agent_outputs))
    self.batch_count.zero_()
    self.batch_start.zero_()
    self._start_host = 0
    return k, learner_lib.Unroll(first, *rest)
