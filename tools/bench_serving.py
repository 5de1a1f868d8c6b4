#!/usr/bin/env python
"""Central inference and the train step TOGETHER on one GPU: env-steps/s served while the learner
trains on exactly those steps (closed loop: every served step ends up in a training unroll, so in steady state the
learner's env-frames/s equals the inference side's env-steps/s -- that common rate is the number reported).

  python tools/bench_serving.py [--mode inprocess|transport|both] [--n 1024] [--batch 512] [--envs 4096]
                                [--procs 16] [--envs-per-proc 64] [--seconds 6]

inprocess  a host thread replays the captured inference graph on pre-staged pinned request batches as fast as the
           back-pressure gate admits them (what a transport that is never the bottleneck would deliver); the main
           thread runs LearnerServer.train_step().  Measures the GPU side: two streams, one device.
transport  the same learner behind the NATIVE gRPC front-end (libseedserve.so): actor PROCESSES, one stream each,
           `envs-per-proc` environments per call (the reference's env_batch_size), requests serialized once per actor
           (the load generator is Python: it must not be the bottleneck) -- reference wire format end to end.
Reference semantics: agents/vtrace/learner.py:350-470 (inference on its own devices beside the training loop).
"""
import argparse
import multiprocessing as mp
import os
import sys
import tempfile
import threading
import time
import uuid

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)
import numpy as np

OBS = (84, 84, 1)


# CUs of every XCD (of 32) that serve inference on a stream of their own (0: one shared pool, inference on a high-priority
# stream).  Measured on one box, in-process feeder (env-steps/s served):
#   n = 2 048 / 8 192 envs:   shared 4.61 M   split 12 (96 + 160 CUs) 5.00-5.03 M   split 16: 4.96 M
#   n = 4 096 / 16 384 envs:  shared 5.48 M   split 12               5.02 M  (the train step on 160 CUs, 2.05 ms, is the bound)
#   transport, n = 2 048:     shared 4.44 M   split 12               4.13 M  (actors are latency-bound, the GPU is not saturated)
# Disjoint CU sets stop an inference kernel from waiting for a train kernel's persistent workgroups to retire, but the
# train step's grids are sized for 256 CUs and lose more than the inference side gains once the batches are large; the
# default runs share the pool, SEEDRL_CU_SPLIT=12 selects the split.
CU_SPLIT = 0


def _make_server(dev, T, B, n, envs, address, transport='native', A=18, io_threads=None, slots=4, pipeline=None,
                 cu_split=0):
  import torch
  from seed_rl_amd import learner, learner_server, networks, optimizers, parametric_distribution as pd
  agent = networks.AtariShallow(A, device=dev, seed=0)
  opt = optimizers.Adam(optimizers.PolynomialDecay(4.8e-4, 10 ** 7), beta_1=0.0, epsilon=3.125e-7, capturable=True)
  lrn = learner.Learner(agent, opt, pd.categorical_distribution(A))
  kw = {} if pipeline is None else dict(inference_pipeline=pipeline)
  srv = learner_server.LearnerServer(agent, lrn, T, B, n, envs, OBS, [address], device=dev, transport=transport,
                                     graphed=True, num_io_threads=io_threads, inference_slots=slots,
                                     cu_split=int(os.environ.get('SEEDRL_CU_SPLIT', cu_split)), **kw)
  return srv


def _stagger(srv, T):
  """Real actors are never in lock step.  Synthetic ones started together would all complete their unrolls in the same
  inference call (a burst of num_envs / batch_size train steps, then nothing for T calls): give every env a random
  phase by starting its store row index somewhere inside the unroll (its first unroll then begins with zero steps)."""
  import torch
  st = srv.state
  g = torch.Generator(device='cpu').manual_seed(1)
  st.store_index.copy_(torch.randint(0, T + 1, (st.E,), generator=g).to(st.store_index.device))


def run_inprocess(dev, seconds=5.0, T=20, B=512, n=1024, envs=4096, warm_steps=3, pipeline=None, depth=None):
  """Returns a dict: env_steps_per_s served == learner env-frames/s consumed, measured over the same wall interval."""
  import torch
  from seed_rl_amd import inference
  path = os.path.join(tempfile.gettempdir(), 'seedrl_s_' + uuid.uuid4().hex[:12])
  groups = envs // n
  srv = _make_server(dev, T, B, n, envs, 'unix:' + path, slots=groups, pipeline=pipeline, cu_split=CU_SPLIT)
  gate = srv.gate
  # the bound inference function's pinned slot buffers, filled the way the C++ front-end fills them: slot k always
  # carries env group k; `compute(slot)` submits a batch (copy stream + inference stream), `finish()` waits for it
  bound = srv.server._bound[0]                                  # pylint: disable=protected-access
  req, obs = bound.keep[0], bound.keep[1]
  g = torch.Generator(device='cpu').manual_seed(0)
  for k in range(groups):
    ids = np.arange(k * n, (k + 1) * n, dtype=np.int64)
    r = torch.randn(n, generator=g).numpy()
    req[k].copy_(torch.from_numpy(inference.pack_request(n, ids, np.full(n, 7, np.int64), r, r,
                                                         (torch.rand(n, generator=g) < 0.01).numpy())))
    obs[k].copy_(torch.randint(0, 256, (n,) + OBS, dtype=torch.uint8, generator=g))
  _stagger(srv, T)
  stop = threading.Event()
  served = [0]
  import queue
  inflight = queue.Queue()
  free = queue.Queue()                                          # env groups whose previous batch was answered
  for k in range(groups if depth is None else min(groups, depth)):
    free.put(k)

  prof = dict(compute=0.0, put=0.0, finish=0.0, get=0.0, calls=0)
  stage, launch = getattr(bound.compute, 'stage', None), getattr(bound.compute, 'launch', None)

  def submitter():
    """The native server's compute loop (grpc_native.NativeServer._compute_loop) with the free list standing in for
    seedserve_next_batch: a group's next batch is 'filled' as soon as its previous one was answered."""
    pending = None
    try:
      while not stop.is_set():
        try:
          slot = free.get(block=pending is None, timeout=0.2)
        except queue.Empty:
          if pending is not None:
            inflight.put((pending[0], launch(pending)))
            pending = None
          continue
        t = time.perf_counter()
        if stage is None:
          inflight.put((slot, bound.compute(slot)))
        else:
          token = stage(slot, 0 if pending is None else 1)
          if pending is not None:
            inflight.put((pending[0], launch(pending)))
          pending = token
        prof['compute'] += time.perf_counter() - t; prof['calls'] += 1
    except Exception:                                           # the gate was closed under us: the run is over
      pass
    inflight.put(None)

  def finisher():
    while True:
      t = time.perf_counter()
      item = inflight.get()
      t1 = time.perf_counter()
      if item is None:
        return
      item[1]()
      prof['get'] += t1 - t; prof['finish'] += time.perf_counter() - t1
      served[0] += n
      free.put(item[0])
  # three busy Python threads in one process (submitter, finisher, trainer): hand the GIL over every 0.1 ms instead of
  # every 5 ms, or the trainer sees its turn a few times per step (the transport mode has no such problem: its
  # per-message work is in C++ threads)
  switch = sys.getswitchinterval()
  sys.setswitchinterval(1e-4)
  ths = [threading.Thread(target=submitter, daemon=True), threading.Thread(target=finisher, daemon=True)]
  for th in ths:
    th.start()
  try:
    for _ in range(warm_steps):                                 # includes the HIP-graph capture of both unroll slots
      assert srv.train_step(timeout=120) is not None
    srv.synchronize()
    s0, t0, steps = served[0], time.perf_counter(), 0
    while time.perf_counter() - t0 < seconds:
      assert srv.train_step(timeout=60) is not None
      steps += 1
    srv.train_stream.synchronize()
    from seed_rl_amd import grpc_native as _gn
    _calls = max(_gn.PROF.get('calls', 0), 1)
    submit_prof = {k: round(v / _calls * 1e6, 1) for k, v in _gn.PROF.items() if k != 'calls'}
    dt = time.perf_counter() - t0
    s1 = served[0]
  finally:
    stop.set()
    gate.open = False                                           # release a submitter blocked in admit()
    for th in ths:
      th.join(timeout=10)
    sys.setswitchinterval(switch)
  srv.state.check_errors()
  srv.shutdown()
  return dict(mode='in-process feeder (no transport): request copies on a copy stream, inference graph replays on the '
                   'high-priority stream, train step graph on its own stream, same device',
              seconds=round(dt, 2), inference_batch=n, train_batch=B, unroll_length=T, envs=envs,
              env_steps_per_s_served=round((s1 - s0) / dt, 0), learner_env_frames_per_s=round(steps * B * T / dt, 0),
              train_steps=steps, ms_per_train_step_wall=round(dt / max(steps, 1) * 1e3, 3),
              inference_calls_per_s=round((s1 - s0) / n / dt, 0), gate_waits=gate.waits,
              cu_split=(dict(inference_cus=srv.infer_cus, train_cus=srv.train_cus) if srv.cu_split else None),
              feeder_us_per_call={k: round(v / max(prof['calls'], 1) * 1e6, 1) for k, v in prof.items() if k != 'calls'},
              submit_us_per_call=submit_prof)


def actor_proc(address, first_env, k, out, start, stop_at):
  import queue
  import grpc
  from seed_rl_amd import grpc_service as gs
  rng = np.random.default_rng(first_env)
  req = gs.CallRequest()
  req.function = 'inference'
  for a in (np.arange(first_env, first_env + k, dtype=np.int32), np.full(k, 7, np.int64),
            rng.normal(size=k).astype(np.float32), rng.uniform(size=k) < 0.01,
            rng.integers(0, 256, (k,) + OBS).astype(np.uint8), np.zeros(k, np.bool_), np.zeros(k, np.int32),
            rng.normal(size=k).astype(np.float32)):
    req.tensor.append(gs.encode_tensor(a))
  blob = req.SerializeToString()
  channel = grpc.insecure_channel(address, options=[('grpc.max_receive_message_length', -1),
                                                    ('grpc.max_send_message_length', -1),
                                                    ('grpc.use_local_subchannel_pool', 1)])
  ident = lambda b: b
  channel.unary_unary('/%s/Init' % gs.SERVICE, request_serializer=ident, response_deserializer=ident)(
      b'', wait_for_ready=True, timeout=120)
  q = queue.SimpleQueue()

  def gen():
    while True:
      item = q.get()
      if item is None:
        return
      yield item
  responses = channel.stream_stream('/%s/Call' % gs.SERVICE, request_serializer=ident, response_deserializer=ident)(gen())
  start.wait()
  calls = 0
  try:
    while time.time() < stop_at.value:
      q.put(blob)
      resp = gs.CallResponse.FromString(next(responses))
      if resp.status_code != 0:
        raise RuntimeError(resp.status_error_message)
      calls += 1
  except (grpc.RpcError, StopIteration):
    pass
  q.put(None)
  out.put(calls * k)


def run_transport(dev, seconds=5.0, T=20, B=512, n=256, procs=16, envs_per_proc=64, io_threads=None, pipeline=None):
  import torch
  envs = procs * envs_per_proc
  assert n % envs_per_proc == 0 and envs % n == 0, 'actors must fill whole inference batches'
  path = os.path.join(tempfile.gettempdir(), 'seedrl_s_' + uuid.uuid4().hex[:12])
  srv = _make_server(dev, T, B, n, envs, 'unix:' + path, io_threads=io_threads, pipeline=pipeline)
  _stagger(srv, T)
  srv.start()
  ctx = mp.get_context('spawn')
  q, start, stop_at = ctx.Queue(), ctx.Event(), ctx.Value('d', time.time() + 3600.0)
  ps = [ctx.Process(target=actor_proc, args=('unix:' + path, i * envs_per_proc, envs_per_proc, q, start, stop_at))
        for i in range(procs)]
  for p in ps:
    p.start()
  time.sleep(4.0)                                               # imports + connects
  start.set()
  try:
    for _ in range(3):                                          # warm-up incl. graph capture
      assert srv.train_step(timeout=120) is not None
    srv.synchronize()
    st0 = srv.server.stats()
    t0, steps = time.perf_counter(), 0
    while time.perf_counter() - t0 < seconds:
      assert srv.train_step(timeout=60) is not None
      steps += 1
    srv.train_stream.synchronize()
    dt = time.perf_counter() - t0
    st1 = srv.server.stats()
  finally:
    stop_at.value = 0.0
    time.sleep(0.2)
    srv.shutdown()                                              # unblocks actors whose last batch can never fill
  total = 0
  for _ in ps:
    try:
      total += q.get(timeout=60)
    except Exception:                                           # pylint: disable=broad-except
      pass
  for p in ps:
    p.join(timeout=30)
  if os.path.exists(path):
    os.remove(path)
  served = (st1['batches'] - st0['batches']) * n
  return dict(mode='actor processes -> native gRPC front-end (libseedserve.so, unix socket, reference wire format) -> '
                   'central inference -> train step, same device',
              seconds=round(dt, 2), actor_processes=procs, envs_per_call=envs_per_proc, inference_batch=n, train_batch=B,
              unroll_length=T, env_steps_per_s_served=round(served / dt, 0),
              learner_env_frames_per_s=round(steps * B * T / dt, 0), train_steps=steps,
              observation_MB_per_s=round((st1['bytes_in'] - st0['bytes_in']) / dt / 1e6, 0),
              calls_per_s=round((st1['calls'] - st0['calls']) / dt, 0), gate_waits=srv.gate.waits,
              cu_split=(dict(inference_cus=srv.infer_cus, train_cus=srv.train_cus) if srv.cu_split else None),
              host_cpus=os.cpu_count())


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--mode', default='both', choices=['inprocess', 'transport', 'both'])
  ap.add_argument('--n', type=int, default=1024)
  ap.add_argument('--tn', type=int, default=256, help='inference batch of the transport run')
  ap.add_argument('--batch', type=int, default=512)
  ap.add_argument('--unroll', type=int, default=20)
  ap.add_argument('--envs', type=int, default=4096)
  ap.add_argument('--procs', type=int, default=16)
  ap.add_argument('--envs-per-proc', type=int, default=64)
  ap.add_argument('--io-threads', type=int, default=0)
  ap.add_argument('--seconds', type=float, default=5.0)
  ap.add_argument('--pipeline', type=int, default=0)
  ap.add_argument('--depth', type=int, default=0)
  a = ap.parse_args()
  import json
  import torch
  dev = torch.device('cuda:0')
  if a.mode in ('inprocess', 'both'):
    print(json.dumps(run_inprocess(dev, a.seconds, a.unroll, a.batch, a.n, a.envs, pipeline=a.pipeline or None,
                                   depth=a.depth or None)))
  if a.mode in ('transport', 'both'):
    print(json.dumps(run_transport(dev, a.seconds, a.unroll, a.batch, a.tn, a.procs, a.envs_per_proc,
                                   a.io_threads or None, pipeline=a.pipeline or None)))


if __name__ == '__main__':
  main()
