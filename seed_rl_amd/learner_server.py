"""The data path of the reference's `learner_loop` around the hot path (agents/vtrace/learner.py:300-483), minimal:
actors --gRPC--> dynamic batching --> central inference on the device store --> completed unrolls (time-major, in HBM)
--> dequeue(batch_size) --> Learner.minimize.  No checkpoint manager, logger thread or tf.data pipeline (control plane,
out of scope: SURVEY.md section 2) -- what is here is exactly what turns the pieces into a servable learner:

  server = LearnerServer(agent, learner, unroll_length=20, batch_size=512, inference_batch_size=256, num_envs=1024,
                         observation_shape=(84, 84, 1), server_addresses=['unix:/tmp/seed', 'localhost:8686'])
  server.start()
  while training: loss, session = server.train_step()     # blocks until batch_size unrolls are complete
  server.shutdown()

Threads: the gRPC service thread(s) run inference batches (grpc_service's executor), the caller's thread trains.  Both
submit to the device's default stream; ONE lock orders their submissions, so a replayed inference graph can never write
into training-batch columns between the dequeue's read of the fill count and its copies.  Inference and training time-share
the GPU (as they do on one TPU core in the reference); several FusedInferenceStates on other devices can be bound
round-robin exactly like the reference's inference_devices (learner.py:406-414).
"""
import threading
import time

import torch

from seed_rl_amd import grpc_service, inference, learner as learner_lib, networks, utils
from seed_rl_amd.unroll_store import Spec


class LearnerServer(object):

  def __init__(self, agent, learner, unroll_length, batch_size, inference_batch_size, num_envs, observation_shape,
               server_addresses, observation_dtype=torch.uint8, device='cuda', graphed=False, batch_capacity=None):
    self.agent, self.learner = agent, learner
    self.T, self.B, self.n = unroll_length, batch_size, inference_batch_size
    self.device = dev = torch.device(device)
    A = agent._num_actions                            # pylint: disable=protected-access
    env_specs = utils.EnvOutput(Spec((), torch.float32), Spec((), torch.bool), Spec(tuple(observation_shape), observation_dtype),
                                Spec((), torch.bool), Spec((), torch.int32))
    ao_specs = networks.AgentOutput(Spec((), torch.int64), Spec((A,), torch.float32), Spec((), torch.float32))
    cap = batch_capacity or max(2 * batch_size, batch_size + num_envs)
    self.state = inference.FusedInferenceState(agent, num_envs, unroll_length, env_specs, ao_specs, batch_capacity=cap,
                                               device=dev)
    self.lock = threading.Lock()
    self.server = grpc_service.Server(list(server_addresses))
    grpc_service.bind_inference(self.server, self.state, inference_batch_size, observation_shape, lock=self.lock)
    # static training unroll (the input of the train step, and of its HIP graph when graphed)
    T1 = unroll_length + 1
    z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)
    init = agent.initial_state(batch_size)
    env = utils.EnvOutput(z((T1, batch_size), torch.float32), z((T1, batch_size), torch.bool),
                          self._obs_buffer(T1, batch_size, observation_shape, observation_dtype),
                          z((T1, batch_size), torch.bool), z((T1, batch_size), torch.int32))
    ao = networks.AgentOutput(z((T1, batch_size), torch.int64), z((T1, batch_size, A), torch.float32),
                              z((T1, batch_size), torch.float32))
    self.unroll = learner_lib.Unroll(init, z((T1, batch_size), torch.int64), env, ao)
    self._graphed = learner_lib.GraphedStep(learner, self.unroll) if graphed else None
    self.steps = 0

  def _obs_buffer(self, T1, B, shape, dtype):
    # Atari agents read their frames from frames_buffer(T1, B)[3:]: dequeue straight into it (no copy in the step)
    if hasattr(self.agent, 'frames_buffer') and len(shape) == 3 and shape[2] == 1 and dtype == torch.uint8:
      return self.agent.frames_buffer(T1, B)[3:].view((T1, B) + tuple(shape))
    return torch.zeros((T1, B) + tuple(shape), dtype=dtype, device=self.device)

  def start(self):
    self.server.start()

  def shutdown(self):
    self.server.shutdown()

  def train_step(self, timeout=None, poll_s=0.0005):
    """learner.py:435-470: dequeue batch_size completed unrolls, minimize.  Returns (loss, session) or None on timeout."""
    t0 = time.time()
    while True:
      with self.lock:
        with torch.cuda.device(self.device):
          ready = self.state.dequeue_into(self.unroll, self.B)
          if ready:
            out = self._graphed() if self._graphed is not None else self.learner.minimize(self.unroll)
      if ready:
        self.steps += 1
        return out
      if timeout is not None and time.time() - t0 > timeout:
        return None
      time.sleep(poll_s)
