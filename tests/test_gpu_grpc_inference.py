"""Actors -> gRPC (wire format of the reference's TensorService) -> dynamic batching -> pinned host request ->
FusedInferenceState's captured HIP graph -> actions back to every actor (learner.py:339-414 end to end).  Eight actor
threads, one env each, single-step requests (the batching dimension left out, as reference actors send them); the server
batches 4 of them per inference call."""
import concurrent.futures as futures
import os
import tempfile
import uuid

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('transport', ['python', 'native'])
def test_actors_through_grpc_into_fused_inference(device, transport):
  from seed_rl_amd import grpc_native as gn, grpc_service as gs, inference, networks, utils
  from seed_rl_amd.unroll_store import Spec
  T, E, A, n = 3, 8, 6, 4
  obs_shape = (84, 84, 1)
  agent = networks.AtariShallow(A, device=device, seed=0)
  env_specs = utils.EnvOutput(Spec((), torch.float32), Spec((), torch.bool), Spec(obs_shape, torch.uint8),
                              Spec((), torch.bool), Spec((), torch.int32))
  ao_specs = networks.AgentOutput(Spec((), torch.int64), Spec((A,), torch.float32), Spec((), torch.float32))
  fused = inference.FusedInferenceState(agent, E, T, env_specs, ao_specs, batch_capacity=4 * E, device=device)
  path = os.path.join(tempfile.gettempdir(), 'seedrl_' + uuid.uuid4().hex[:12])
  address = 'unix:' + path
  if transport == 'native':                            # libseedserve.so: pinned request_layout slots filled by the C++ side
    server = gn.NativeServer([address], num_io_threads=2)
    gn.bind_inference(server, fused, n, obs_shape)
  else:
    server = gs.Server([address])
    gs.bind_inference(server, fused, n, obs_shape)
  server.start()
  steps = 2 * T + 1
  # closed-loop actors must stay in step: were some of them allowed to run ahead, the last calls of the slow ones
  # (fewer than n of them, each waiting for its answer) could never fill a batch -- the reference blocks the same way
  import threading
  in_step = threading.Barrier(E)

  def actor(env_id):
    rng = np.random.default_rng(env_id)
    client = gs.Client(address)
    acts = []
    for step in range(steps):
      in_step.wait(timeout=120)
      env = utils.EnvOutput(np.float32(rng.normal()), np.bool_(step > 0 and rng.uniform() < 0.2),
                            rng.integers(0, 256, obs_shape).astype(np.uint8), np.bool_(False), np.int32(step))
      a = client.inference(np.int32(env_id), np.int64(1000 + env_id), env, np.float32(rng.normal()))
      assert a.shape == () and a.dtype == np.int64 and 0 <= int(a) < A
      acts.append(int(a))
    client.close()
    return acts
  try:
    with futures.ThreadPoolExecutor(max_workers=E) as ex:
      results = [f.result(timeout=120) for f in [ex.submit(actor, e) for e in range(E)]]
  finally:
    server.shutdown()
    if os.path.exists(path):
      os.remove(path)
  torch.cuda.synchronize()
  fused.check_errors()
  # every env contributed T+1 then T more steps: two completed unrolls each, written time-major into the batch
  k, batch = fused.take_batch()
  assert k == 2 * E
  assert tuple(batch.env_outputs.observation.shape[:2]) == (T + 1, k)
  # the actions the actors received are the ones stored in the unrolls (first unroll of env e: its first T+1 actions)
  stored = batch.agent_outputs.action.cpu().numpy()                    # [T+1, k]
  steps_of = batch.env_outputs.episode_step.cpu().numpy()
  found = 0
  for col in range(k):
    if steps_of[0, col] != 0:
      continue
    for e in range(E):
      if stored[:, col].tolist() == results[e][:T + 1]:
        found += 1
        break
  assert found == E
  # on-policy consistency: re-running the training unroll on the emitted unrolls reproduces the stored logits
  out, _ = agent(batch.prev_actions, batch.env_outputs, batch.agent_state, unroll=True, is_training=True)
  assert torch.allclose(out.policy_logits, batch.agent_outputs.policy_logits, atol=2e-5)
  assert torch.allclose(out.baseline, batch.agent_outputs.baseline, atol=2e-5)
