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
from seed_rl_amd import unroll_store, utils
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
