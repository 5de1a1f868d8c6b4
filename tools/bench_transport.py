#!/usr/bin/env python
"""Transport-only throughput (host code, no GPU): actor processes -> gRPC (reference wire format) -> server-side
batching -> a no-op inference function -> actions back.  Compares the native front-end (libseedserve.so,
seed_rl_amd/grpc_native.py) with the asyncio one (seed_rl_amd/grpc_service.py).

  python tools/bench_transport.py [--server native|python] [--procs 8] [--envs-per-proc 64] [--n 256] [--seconds 5]
Each actor process holds ONE stream and sends its `envs-per-proc` environments as one client-side batch per call (the
reference's env_batch_size, common/actor.py); requests are serialized once and re-sent (the load generator must not
be the bottleneck), responses are parsed.  Prints env-steps/s, calls/s and MB/s of observation bytes.
"""
import argparse
import multiprocessing as mp
import os
import sys
import tempfile
import time
import uuid

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

OBS = (84, 84, 1)


def actor_proc(address, first_env, k, seconds, out, start):
  import grpc
  from seed_rl_amd import grpc_service as gs
  rng = np.random.default_rng(first_env)
  req = gs.CallRequest()
  req.function = 'inference'
  ids = np.arange(first_env, first_env + k, dtype=np.int32)
  for a in (ids, np.full(k, 7, np.int64), rng.normal(size=k).astype(np.float32), rng.uniform(size=k) < 0.01,
            rng.integers(0, 256, (k,) + OBS).astype(np.uint8), np.zeros(k, np.bool_), np.zeros(k, np.int32),
            rng.normal(size=k).astype(np.float32)):
    req.tensor.append(gs.encode_tensor(a))
  blob = req.SerializeToString()
  channel = grpc.insecure_channel(address, options=[('grpc.max_receive_message_length', -1),
                                                    ('grpc.max_send_message_length', -1),
                                                    ('grpc.use_local_subchannel_pool', 1)])
  init = channel.unary_unary('/%s/Init' % gs.SERVICE, request_serializer=lambda b: b, response_deserializer=lambda b: b)
  init(b'', wait_for_ready=True, timeout=60)
  call = channel.stream_stream('/%s/Call' % gs.SERVICE, request_serializer=lambda b: b, response_deserializer=lambda b: b)
  import queue
  q = queue.SimpleQueue()

  def gen():
    while True:
      item = q.get()
      if item is None:
        return
      yield item
  responses = call(gen())
  start.wait()
  t_end = time.time() + seconds
  calls = 0
  try:
    while time.time() < t_end:
      q.put(blob)
      raw = next(responses)
      resp = gs.CallResponse.FromString(raw)
      if resp.status_code != 0:
        raise RuntimeError(resp.status_error_message)
      calls += 1
  except (grpc.RpcError, StopIteration):
    pass
  q.put(None)
  out.put(calls * k)


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--server', default='native', choices=['native', 'python'])
  ap.add_argument('--procs', type=int, default=8)
  ap.add_argument('--envs-per-proc', type=int, default=64)
  ap.add_argument('--n', type=int, default=256)
  ap.add_argument('--seconds', type=float, default=5.0)
  ap.add_argument('--io-threads', type=int, default=0)
  ap.add_argument('--slots', type=int, default=4)
  a = ap.parse_args()
  from seed_rl_amd import grpc_native as gn, grpc_service as gs
  n = a.n
  assert n % a.envs_per_proc == 0 and (a.procs * a.envs_per_proc) % n == 0, 'actors must fill whole batches'
  path = os.path.join(tempfile.gettempdir(), 'seedrl_t_' + uuid.uuid4().hex[:12])
  sig = gn.inference_signature(n, OBS)

  @gs.function(sig, gs.TensorSpec((n,), np.int64, 'action'))
  def inference(env_ids, run_ids, env_outputs, raw_rewards):
    return env_ids.astype(np.int64)
  if a.server == 'native':
    server = gn.NativeServer(['unix:' + path], num_io_threads=a.io_threads or None)
    server.bind(inference, num_slots=a.slots)
  else:
    server = gs.Server(['unix:' + path])
    server.bind(inference)
  server.start()
  ctx = mp.get_context('spawn')
  q, start = ctx.Queue(), ctx.Event()
  procs = [ctx.Process(target=actor_proc, args=('unix:' + path, i * a.envs_per_proc, a.envs_per_proc, a.seconds, q, start))
           for i in range(a.procs)]
  for p in procs:
    p.start()
  time.sleep(3.0)                                      # imports + connects
  t0 = time.time()
  start.set()
  time.sleep(a.seconds)
  dt = time.time() - t0
  st = server.stats() if a.server == 'native' else None
  server.shutdown()                                    # unblocks the actors whose last batch can never fill
  total = sum(q.get(timeout=120) for _ in procs)
  for p in procs:
    p.join(timeout=30)
  if os.path.exists(path):
    os.remove(path)
  per = int(np.prod(OBS))
  print('%s server: %.0f env-steps/s, %.0f calls/s, %.0f MB/s of observations (%d actor processes x %d envs per call, '
        'inference batch %d, %.1f s)' % (a.server, total / dt, total / a.envs_per_proc / dt, total * per / dt / 1e6,
                                         a.procs, a.envs_per_proc, n, dt))
  if st is not None:
    print(st)


if __name__ == '__main__':
  main()
