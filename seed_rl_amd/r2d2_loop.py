"""R2D2 end to end around the train step: actor-side exploration, initial priorities, prioritized replay, sampling,
priority write-back and the target-network schedule.

Mirrors the data path of /root/reference/agents/r2d2/learner.py outside `minimize`:
  * `get_envs_epsilon` / `apply_epsilon_greedy`           :129-177 (csrc/replay.hip: seedhip_epsilon_greedy)
  * `R2D2InferenceState.inference`                        :709-830 (run-id resets, episode stats, single-step agent
                                                          forward, epsilon-greedy, training-env filter, UnrollStore
                                                          with burn-in overlap, initial priorities, unroll queue)
  * `ReplayTrainer.insert / train_step`                   :387-468 (`create_dataset`: dequeue -> replay insert ->
                                                          min-size gate -> prioritized sample -> time-major) and
                                                          :856-885 (target update every N steps, minimize,
                                                          update_priorities)
The replay buffer and the unroll store live in HBM; unrolls travel store -> replay -> training batch time-major with one
row move per field and direction (replay.UnrollReplay).  No checkpointing / logging thread (control plane, out of scope).
"""
import collections

import numpy as np
import torch

from seed_rl_amd import inference as inference_lib
from seed_rl_amd import networks, ops, r2d2_learner, replay, unroll_store, utils
from seed_rl_amd.unroll_store import Spec

SampledUnrolls = collections.namedtuple('SampledUnrolls', 'unrolls indices importance_weights')   # learner.py:99-100


def get_envs_epsilon_table(num_training_envs, num_eval_envs, eval_epsilon, device='cuda'):
  """learner.py:129-145: epsilons[i] = 0.4 ** linspace(1, 8, num_training_envs)[i] for training envs, eval_epsilon
  for the eval envs that follow them.  fp32 like the reference (tf.linspace / tf.math.pow on float32)."""
  lin = np.linspace(np.float32(1.), np.float32(8.), num_training_envs, dtype=np.float32)
  eps = np.concatenate([np.power(np.float32(0.4), lin).astype(np.float32),
                        np.full((num_eval_envs,), eval_epsilon, np.float32)])
  return torch.as_tensor(eps).to(device)


def apply_epsilon_greedy(actions, env_ids, epsilons, num_actions, rng_state, replaced=None):
  """learner.py:147-177 on the device, in place: returns `actions` (int64)."""
  ops.epsilon_greedy(actions, env_ids, epsilons, num_actions, rng_state, replaced)
  return actions


def initial_priorities(agent, unroll_tm, burn_in, config):
  """learner.py:802-816: priorities of freshly completed unrolls from the BEHAVIOUR q-values (used as training and as
  target network outputs), suffix after the burn-in prefix.  unroll_tm: time-major fields [T1, n, ...]."""
  cfg = config
  q = unroll_tm.agent_outputs.q_values[burn_in:].to(torch.float32).contiguous()
  T, n, A = q.shape
  dev = q.device
  loss_b = torch.empty(n, device=dev)
  prio = torch.empty(n, device=dev)
  total = torch.empty(1, device=dev)
  dq = torch.empty_like(q)
  ws = torch.empty(ops.r2d2_loss_workspace_bytes(T, n, cfg.n_steps) // 4 + 4, device=dev)
  ops.r2d2_loss_fwd_bwd(q, q, unroll_tm.agent_outputs.action[burn_in:].to(torch.int32).contiguous(),
                        unroll_tm.env_outputs.reward[burn_in:].to(torch.float32).contiguous(),
                        ops.as_u8(unroll_tm.env_outputs.done[burn_in:].contiguous()), None, T, n, A, cfg.discounting,
                        cfg.n_steps, cfg.eta, cfg.value_function_rescaling_epsilon, n, loss_b, prio, dq, total, ws)
  return prio


def unroll_specs(agent, unroll_length, burn_in, observation_shape, num_actions, observation_dtype=torch.uint8):
  """Per-row Specs of the replay buffer (learner.py:659-668): Unroll(agent_state, priority, prev_actions, env_outputs,
  agent_outputs) with T1 = burn_in + unroll_length + 1 leading steps on the per-timestep fields."""
  T1 = burn_in + unroll_length + 1
  t = lambda shape, dt: Spec((T1,) + tuple(shape), dt)
  return r2d2_learner.Unroll(
      agent_state=unroll_store.specs_like(agent.initial_state(1)),
      priority=Spec((), torch.float32),
      prev_actions=t((), torch.int64),
      env_outputs=utils.EnvOutput(t((), torch.float32), t((), torch.bool), t(observation_shape, observation_dtype),
                                  t((), torch.bool), t((), torch.int32)),
      agent_outputs=networks.R2D2AgentOutput(t((), torch.int64), t((num_actions,), torch.float32)))


class ReplayTrainer(object):
  """create_dataset + the body of the training loop (learner.py:387-468, 856-885) for one replica."""

  def __init__(self, learner, specs, replay_buffer_size=int(1e5), replay_buffer_min_size=5000, priority_exponent=0.9,
               importance_sampling_exponent=0.6, batch_size=64, device='cuda'):
    """Flag defaults of learner.py:55-76.  `learner`: an r2d2_learner.R2D2Learner."""
    self.learner = learner
    self.replay = replay.UnrollReplay(replay_buffer_size, specs, importance_sampling_exponent, device=device)
    self.min_size, self.priority_exponent, self.batch_size = replay_buffer_min_size, priority_exponent, batch_size
    self.last = None

  def insert(self, unroll_tm, priorities=None):
    """replay_buffer.insert(unrolls, unrolls.priority) (:436) for time-major completed unrolls."""
    pr = unroll_tm.priority if priorities is None else priorities
    return self.replay.insert_time_major(unroll_tm._replace(priority=pr), pr)

  def ready(self):
    return self.replay.num_inserted >= self.min_size                                  # :442

  def sample(self, uniforms=None):
    idx, w, u = self.replay.sample_time_major(self.batch_size, self.priority_exponent, uniforms)
    return SampledUnrolls(u, idx, w)

  def train_step(self, uniforms=None):
    """One iteration of the loop at :856-885: (target update on its schedule -- R2D2Learner.minimize does it after the
    step whose count hits the period, the constructor at step 0) sample, minimize, update_priorities.
    Returns (loss, priorities [B], indices [B], gradient norm before clipping)."""
    if not self.ready():
      raise RuntimeError('replay buffer holds %d unrolls, fewer than replay_buffer_min_size=%d'
                         % (self.replay.num_inserted, self.min_size))
    s = self.sample(uniforms)
    loss, prio, gnorm = self.learner.minimize(s.unrolls, s.importance_weights)
    self.replay.update_priorities(s.indices, prio)                                     # :885
    self.last = s
    return loss, prio, s.indices, gnorm


class R2D2InferenceState(inference_lib.InferenceState):
  """The R2D2 `inference` function (learner.py:709-830) on the device store: InferenceState's bookkeeping with the
  R2D2 agent's call signature, epsilon-greedy exploration, eval environments that are served but never stored,
  burn-in overlap between consecutive unrolls and initial priorities for the replay."""

  def __init__(self, agent, num_training_envs, num_eval_envs, unroll_length, burn_in, observation_shape,
               config=None, eval_epsilon=1e-3, num_action_repeats=1, device='cuda', unroll_sink=None, info_sink=None,
               observation_dtype=torch.uint8, seed=0x5EED):
    self.num_training_envs, self.num_eval_envs = num_training_envs, num_eval_envs
    self.burn_in, self.config = burn_in, config or r2d2_learner.R2D2Config(burn_in=burn_in)
    A = agent._num_actions                                  # pylint: disable=protected-access
    num_envs = num_training_envs + num_eval_envs
    env_specs = utils.EnvOutput(Spec((), torch.float32), Spec((), torch.bool),
                                Spec(tuple(observation_shape), observation_dtype), Spec((), torch.bool),
                                Spec((), torch.int32))
    ao_specs = networks.R2D2AgentOutput(Spec((), torch.int64), Spec((A,), torch.float32))
    super(R2D2InferenceState, self).__init__(agent, num_envs, unroll_length, env_specs, ao_specs, Spec((), torch.int64),
                                             num_action_repeats, device, unroll_sink, info_sink)
    # learner.py:671-676: only training envs are stored; burn-in steps overlap between consecutive unrolls
    self.store = unroll_store.UnrollStore(num_training_envs, unroll_length, (Spec((), torch.int64), env_specs, ao_specs),
                                          num_overlapping_steps=burn_in, device=device)
    self.epsilons = get_envs_epsilon_table(num_training_envs, num_eval_envs, eval_epsilon, self.device)
    self.rng = torch.tensor([seed, 0], dtype=torch.int64, device=self.device)
    self.num_actions = A

  def inference(self, env_ids, run_ids, env_outputs, raw_rewards):
    dev = self.device
    env_ids = torch.as_tensor(env_ids, device=dev).to(torch.int64)
    run_ids = torch.as_tensor(run_ids, device=dev).to(torch.int64)
    ntrain = self.num_training_envs
    # Reset the environments that had their first run or crashed (:737-752).
    previous_run_ids = self.env_run_ids.read(env_ids)
    self.env_run_ids.replace(env_ids, run_ids)
    need_reset = env_ids[previous_run_ids != run_ids]
    if need_reset.numel():
      self.env_infos.reset(need_reset)
      self.store.reset(need_reset[need_reset < ntrain])
      init = self.agent.initial_state(int(need_reset.numel()))
      self.first_agent_states.replace(need_reset, init)
      self.agent_states.replace(need_reset, init)
      self.actions.reset(need_reset)
    if env_outputs.abandoned is not None and bool(env_outputs.abandoned.any()):
      raise ValueError('Abandoned done states are not supported in R2D2.')             # :754-756
    # Update steps and return (:759-764).
    n = env_ids.numel()
    zeros_i = torch.zeros(n, dtype=torch.int64, device=dev)
    zeros_f = torch.zeros(n, dtype=torch.float32, device=dev)
    EI = inference_lib.EpisodeInfo
    self.env_infos.add(env_ids, EI(zeros_i, env_outputs.reward, raw_rewards))
    done_ids = env_ids[env_outputs.done]
    if done_ids.numel():
      self.info_sink((self.env_infos.read(done_ids), done_ids))
      self.env_infos.reset(done_ids)
    self.env_infos.add(env_ids, EI(zeros_i + self.num_action_repeats, zeros_f, zeros_f))
    # Inference (:767-786) + exploration (:788-792).
    prev_actions = self.actions.read(env_ids)
    prev_agent_states = self.agent_states.read(env_ids)
    agent_outputs, curr_agent_states = self.agent((prev_actions, env_outputs), prev_agent_states, unroll=False)
    action = agent_outputs.action.to(torch.int64).contiguous()
    apply_epsilon_greedy(action, env_ids.contiguous(), self.epsilons, self.num_actions, self.rng)
    agent_outputs = networks.R2D2AgentOutput(action, agent_outputs.q_values.contiguous())
    # Append for training envs only; completed unrolls get their initial priorities and go to the queue (:794-823).
    train = env_ids < ntrain
    tids = env_ids[train]
    store_env = env_outputs._replace(
        abandoned=env_outputs.abandoned if env_outputs.abandoned is not None else torch.zeros_like(env_outputs.done),
        episode_step=env_outputs.episode_step if env_outputs.episode_step is not None else zeros_i.to(torch.int32))
    sel = lambda t: t[train].contiguous()
    completed_ids, unrolls = self.store.append(
        tids, utils.map_structure(sel, (prev_actions, store_env, agent_outputs)))
    if completed_ids.numel():
      states = self.first_agent_states.read(completed_ids)
      u = r2d2_learner.Unroll(states, None, *unrolls)
      u = u._replace(priority=initial_priorities(self.agent, u, self.burn_in, self.config))
      self.unroll_sink(u)
      self.first_agent_states.replace(completed_ids, self.agent_states.read(completed_ids))
    # Update current state (:826-827).
    self.agent_states.replace(env_ids, curr_agent_states)
    self.actions.replace(env_ids, agent_outputs.action)
    return agent_outputs.action
