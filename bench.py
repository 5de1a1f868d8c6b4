#!/usr/bin/env python
"""bench.py -- learner env-frames/s (T=20) of the MI355X-native SEED-RL learner step.

Workload (BASELINE.json configs[1]): Atari 84x84x4 IMPALA shallow ConvNet, T=20, B=512
per GPU, synthetic uint8 frames resident in HBM, fp32 everywhere (the reference never
uses mixed precision).  One "step" = Learner.minimize(unroll): frame stacking -> conv
torso -> heads -> fused V-trace/loss head -> full backward -> (N>1: RCCL all-reduce of
the flat gradient bucket) -> fused Adam.  env-frames/s = N*B*T*num_action_repeats/step
with num_action_repeats=1 (common/common_flags.py:43; learner.py:236-237).

  python bench.py --gpus 1 --steps 20 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
      --master-port P bench.py --gpus N --steps K --warmup W

The default single-GPU run also reports, outside the timed region and in the same JSON line:
  parity         one train step at the bench shape against the CPU oracle (tests/parity.py)
  other_configs  BASELINE configs[2] (DMLab ImpalaDeep+LSTM) and configs[4] (R2D2), a few steps each
  inference      central-inference step (learner.py:350-405) at n = 64 / 256 / 1024
  serving        inference and training together on the one GPU, without and with the native gRPC transport
  ingest         the same step with the NEXT unroll's host->device copy in flight on its own stream
  cpu_baseline   the oracle's eager PyTorch-CPU learner step on the host cores
(--quick skips them.)
"""
import argparse
import gc
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
HBM_ACHIEVABLE_GBS = 6300.0  # the same guide: 6.29 TB/s measured (float4 copy) -- the whole-step floor is priced with this
MFMA_F32_PEAK_TF = 157.3   # fp32-input MFMA dense peak (v_mfma_f32_16x16x4_f32)
MFMA_BF16_PEAK_TF = 2500.0 # bf16 dense MFMA peak (MI355X_MICROARCH.md; no sparsity)
# Kernels that evaluate their fp32 algorithm on the bf16 pipe through the exact three-way operand split
# (csrc/stackconv.hip): every algorithmic MAC costs three bf16 MACs, so their ceiling for ALGORITHMIC flops is peak / 3.
BF16X3_KERNELS = ('stack_conv_fwd', 'stack_conv_wgrad')
PARITY_NOTE = ('oracle = torch-CPU fp32 restatement of the reference graph (oracle/nets_torch.py); V-trace / R2D2 loss '
               'math pinned to outputs of the reference code, Keras Conv2D/LSTMCell/Dense/Adam numerics UNPINNED (no TF here). '
               'grad_q99 = worst tensor\'s 99th-percentile |g - g_ref| / max|g_ref|; grad_max additionally sees the ReLU units '
               'whose pre-activation is within fp32 rounding of 0 and resolves differently (cfg2: ONE of 2.75 M Dense outputs '
               '-> 3.3e-3 on one fc/kernel column; against an fp64 evaluation the fp32 oracle itself is 3.4e-4 / 4.7e-4 away, '
               'the HIP path 3.7e-4 at q99: tests/test_gpu_fullsize.py, tools/diag_parity.py)')


def parse():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=20)
  ap.add_argument('--warmup', type=int, default=5)
  ap.add_argument('--batch', type=int, default=0, help='per-GPU batch columns B (default 512 atari / 256 dmlab)')
  ap.add_argument('--unroll', type=int, default=0, help='T (default 20; 120 for r2d2)')
  ap.add_argument('--actions', type=int, default=0, help='default 18 atari / 9 dmlab')
  ap.add_argument('--config', default='atari', choices=['atari', 'dmlab', 'r2d2'],
                  help='atari = BASELINE configs[1] (headline); dmlab = configs[2] (ImpalaDeep + LSTM, B=256); '
                       'r2d2 = configs[4] (DuelingLSTMDQNNet, replayed T=120 B=256 sequences, burn-in 40)')
  ap.add_argument('--torso', default='shallow', choices=['shallow', 'dqn'])
  ap.add_argument('--no-cpu-baseline', action='store_true')
  ap.add_argument('--cpu-batch', type=int, default=0,
                  help='batch columns of the CPU-baseline sample (default: 256 atari, 16 dmlab, 8 r2d2: 10-30 s of host time)')
  ap.add_argument('--reduction', default='mean', choices=['mean', 'sum'])
  ap.add_argument('--graph', type=int, default=1,
                  help='1 (default): replay the step from captured HIP graphs (learner.GraphedStep; N > 1: graph segments '
                       'with the gradient exchange launched between them); 0: eager launches.')
  ap.add_argument('--ingest', default='resident', choices=['resident', 'pinned'],
                  help="resident (headline): the unroll is in HBM when the timed region starts; pinned: every step's "
                       'unroll comes from pinned host memory, the copy of step i+1 on its own stream under step i')
  ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'],
                  help="process group for --gpus > 1: nccl (= RCCL over xGMI, one rank per GPU) or gloo -- a DRY RUN of "
                       'the N-rank code path on fewer devices (ranks share GPUs, the exchange goes through host memory); '
                       'its throughput is not a scaling measurement and the line says so')
  ap.add_argument('--force-exchange', action='store_true',
                  help='run the N-replica launch mode (graph segments, asynchronous range all-reduces on the collective '
                       'stream, update graph) even with ONE rank: the RCCL rehearsal of the multi-GPU line on a 1-GPU box')
  ap.add_argument('--windows', type=int, default=6,
                  help='timed windows of --steps steps each: the first is the contract\'s (value), all of them give the '
                       'median / min / max of the line\'s `windows` record')
  ap.add_argument('--quick', action='store_true',
                  help='only the timed learner step: no parity / other_configs / inference / ingest / cpu_baseline records')
  return ap.parse_args()


def vtrace_checks(dev):
  """V-trace fp32 max-abs-err vs the oracle on configs[0] + HBM roofline of the scan."""
  from oracle import vtrace_np
  from seed_rl_amd import vtrace
  from tests import synth
  err = 0.0
  for seed in (0, 1, 2):
    inp = synth.vtrace_inputs(seed, 20, 32, 6)
    out = vtrace.from_importance_weights(**{k: torch.as_tensor(v).to(dev) for k, v in inp.items()})
    ref = vtrace_np.from_importance_weights(**inp)
    err = max(err, float(np.max(np.abs(out.vs.cpu().numpy() - ref.vs))),
              float(np.max(np.abs(out.pg_advantages.cpu().numpy() - ref.pg_advantages))))
  # ... and against the committed outputs of the reference's own common/vtrace.py (tests/golden/make_golden.py),
  # on the cases run with the learner's clipping (rho_bar = 1)
  gpath = os.path.join(ROOT, 'tests', 'golden', 'reference_outputs.npz')
  err_ref = None
  if os.path.exists(gpath):
    G = np.load(gpath)
    err_ref = 0.0
    for n in range(int(G['vtrace_num_cases'])):
      seed, stress, lam, cr, cp = G['vtrace_%02d_meta' % n]
      if cr != 1.0:
        continue
      inp = synth.vtrace_inputs(int(seed), 20, 32, 6, stress=bool(stress))
      out = vtrace.from_importance_weights(**{k: torch.as_tensor(v).to(dev) for k, v in inp.items()}, lambda_=float(lam))
      err_ref = max(err_ref, float(np.max(np.abs(out.vs.cpu().numpy() - G['vtrace_%02d_vs' % n]))),
                    float(np.max(np.abs(out.pg_advantages.cpu().numpy() - G['vtrace_%02d_pg' % n]))))
  sweep = []
  for B in (512, 1 << 14, 1 << 17, 1 << 20, 1 << 22):
    T = 20
    t = [torch.rand((T, B), device=dev) for _ in range(5)]
    boot = torch.rand(B, device=dev)
    for _ in range(3):
      vtrace.from_importance_weights(t[0], t[1], t[2], t[3], t[4], boot)
    reps = 20
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    from seed_rl_amd import _lib
    vs, pg = torch.empty_like(t[0]), torch.empty_like(t[0])
    s.record()
    for _ in range(reps):
      _lib.lib().seedhip_vtrace_from_importance_weights(
          _lib.ptr(t[0]), _lib.ptr(t[1]), _lib.ptr(t[2]), _lib.ptr(t[3]), _lib.ptr(t[4]), _lib.ptr(boot),
          1.0, 1.0, 1.0, T, B, _lib.ptr(vs), _lib.ptr(pg), _lib.stream())
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / reps
    nbytes = T * B * 28 + B * 4
    sweep.append(dict(B=B, us=round(us, 2), GBs=round(nbytes / us / 1e3, 1),
                      frac_hbm=round(nbytes / us / 1e3 / HBM_PEAK_GBS, 4)))
  return err, err_ref, sweep


PIPES = {
    'f32': (MFMA_F32_PEAK_TF, 'fp32 MFMA (v_mfma_f32_16x16x4_f32)'),
    'bf16x3': (MFMA_BF16_PEAK_TF / 3.0, 'bf16 MFMA, exact 3-way split of the fp32 operand (the other is uint8): peak = 2500 / 3 '
                                        'algorithmic TFLOP/s'),
    'bf16x6': (MFMA_BF16_PEAK_TF / 6.0, 'bf16 MFMA, exact 3-way split of BOTH fp32 operands, six of the nine products (every '
                                        'term above 2^-25 relative; csrc/xgemm.h): peak = 2500 / 6 algorithmic TFLOP/s'),
}


def _peak(entry):
  """(peak algorithmic TFLOP/s, pipe description) of a Profiler.summary() entry."""
  return PIPES[entry.get('pipe', 'f32')]


def step_roofline(kern, ms_per_step):
  """Whole-step roofline (VERDICT r3 task 4): sum over the step's kernels of max(algorithmic bytes / achievable HBM
  rate, algorithmic flops / that kernel's pipe peak), divided by the measured step.  Bytes and flops are the figures
  every ops.* wrapper states for its launch; the HBM rate is the chip's measured 6.3 TB/s
  (MI355X_MICROARCH.md), not the 8 TB/s spec, so that the floor is one a kernel could actually reach."""
  floor, parts = 0.0, {}
  for k, v in kern.items():
    # per group of equally sized calls (ops.aggregate): a region name can cover launches of different sizes
    groups = v.get('groups') or [dict(calls=v['calls'], flops=v['flops'], bytes=v['bytes'])]
    f, t_hbm, t_mfma = 0.0, 0.0, 0.0
    for g in groups:
      g_hbm = g['bytes'] / (HBM_ACHIEVABLE_GBS * 1e9) * 1e3
      g_mfma = g['flops'] / (_peak(v)[0] * 1e12) * 1e3
      f += max(g_hbm, g_mfma) * g['calls']
      t_hbm += g_hbm * g['calls']; t_mfma += g_mfma * g['calls']
    floor += f
    if f >= 0.02 * ms_per_step or v['total_ms'] >= 0.02 * ms_per_step:
      parts[k] = dict(floor_ms=round(f, 4), measured_ms=round(v['total_ms'], 4), bound='hbm' if t_hbm > t_mfma else 'mfma',
                      pipe=v.get('pipe', 'f32'))
  return dict(floor_ms=round(floor, 4), ms_per_step=round(ms_per_step, 4), frac=round(floor / ms_per_step, 4),
              hbm_GBs=HBM_ACHIEVABLE_GBS, kernels=parts,
              note='sum over kernels of max(algorithmic bytes / 6.3 TB/s, algorithmic flops / pipe peak) / ms_per_step; '
                   'measured_ms is the serialized attribution pass (sums above the free-running step)')


def _release():
  from seed_rl_amd import ops
  ops._SPLITK_WS.clear()            # pylint: disable=protected-access
  gc.collect()
  torch.cuda.empty_cache()


def build_workload(config, torso, T, B, A, dev, reduction, graph, world, seed, force_exchange=False):
  """Agent + learner + one synthetic unroll resident in HBM for a BASELINE config."""
  from seed_rl_amd import learner, networks, optimizers, parametric_distribution as pd, smoke_step
  T1 = T + 1
  final_iteration = 10 ** 9 // (T * B * max(world, 1))
  extra = ()
  if config == 'r2d2':
    from seed_rl_amd import r2d2_learner
    agent = networks.DuelingLSTMDQNNet(A, device=dev, seed=0)
    target = networks.DuelingLSTMDQNNet(A, device=dev, seed=0)
    u0 = smoke_step.make_unroll(agent, T1, B, A, dev, seed=seed)
    unroll = r2d2_learner.Unroll(agent.initial_state(B), None, u0.prev_actions, u0.env_outputs,
                                 networks.R2D2AgentOutput(u0.agent_outputs.action.to(torch.int32), None))
    extra = (torch.rand(B, device=dev) * 0.9 + 0.1,)                          # importance weights
    opt = optimizers.Adam(4.8e-4, epsilon=1e-3, capturable=bool(graph))       # atari/r2d2_main.py:36-39
    lrn = r2d2_learner.R2D2Learner(agent, target, opt, r2d2_learner.R2D2Config(), reduction=reduction)
    workload = 'Atari R2D2 DuelingLSTMDQNNet (conv 32/64/64, FC512, LSTM512, dueling heads) learner step on replayed ' \
               'sequences, burn-in 40, n-step(5) double-Q target, training + target network'
    return agent, lrn, unroll, extra, workload
  opt = optimizers.Adam(optimizers.PolynomialDecay(4.8e-4, final_iteration), beta_1=0.0, epsilon=3.125e-7,
                        capturable=bool(graph))
  if config == 'dmlab':
    agent = networks.ImpalaDeep(A, device=dev, seed=0)                        # identical params on all ranks
    unroll = smoke_step.make_deep_unroll(agent, T1, B, A, dev, seed=seed)
    workload = 'DeepMind Lab 72x96x3 IMPALA deep ResNet + LSTM(256) learner step'
  else:
    agent = networks.AtariShallow(A, torso=torso, device=dev, seed=0)
    unroll = smoke_step.make_unroll(agent, T1, B, A, dev, seed=seed)
    # place the frames directly in the agent's extended trajectory buffer (no per-step copy)
    ext = agent.frames_buffer(T1, B)
    ext[3:].copy_(unroll.env_outputs.observation.reshape(T1, B, -1))
    unroll = unroll._replace(env_outputs=unroll.env_outputs._replace(
        observation=ext[3:].view(T1, B, agent._obs[0], agent._obs[1], 1)))
    workload = 'Atari 84x84x4 IMPALA %s ConvNet learner step' % torso
  lrn = learner.Learner(agent, opt, pd.categorical_distribution(A), reduction=reduction, force_exchange=force_exchange)
  return agent, lrn, unroll, extra, workload


def run_learner(config, steps, warmup, dev, rank=0, world=1, distributed=False, torso='shallow', batch=0, unroll_len=0,
                actions=0, reduction='mean', graph=1, attribution_steps=3, force_exchange=False, extra_windows=0):
  """Times `steps` learner steps of one BASELINE config; returns the record the JSON line is built from."""
  from seed_rl_amd import learner, ops
  T = unroll_len or (120 if config == 'r2d2' else 20)
  B = batch or (256 if config in ('dmlab', 'r2d2') else 512)
  A = actions or (9 if config == 'dmlab' else 18)
  agent, lrn, unroll, extra, workload = build_workload(config, torso, T, B, A, dev, reduction, graph, world, 1000 + rank,
                                                       force_exchange and config != 'r2d2')

  def barrier():
    if distributed:
      torch.distributed.barrier()
    torch.cuda.synchronize()

  # ---- warmup, then the per-kernel attribution pass: the device is drained before every region, i.e. every kernel
  # is timed the way rocprofv3 times it (serialised dispatches) -- profiles/*_kernel_stats.csv is the same view ----
  for _ in range(max(warmup - attribution_steps, 1 if warmup else 0)):
    lrn.minimize(unroll, *extra)
  prof_all = ops.Profiler(serialize=True)
  ops.set_profiler(prof_all)
  for _ in range(min(attribution_steps, max(warmup, 1))):
    lrn.minimize(unroll, *extra)
  ops.set_profiler(None)
  kern = prof_all.summary()
  # what an event pair around ONE dispatch on a drained device reads for a kernel that does (almost) nothing: the
  # dispatch latency of an idle queue sits inside every serialized region above and not inside rocprofv3's kernel
  # durations (r6: stack_conv_fwd 139.1 us by events, 130.0 us in the kernel trace of the same box)
  tiny = torch.zeros(64, device=dev)
  floor_ms = []
  for _ in range(20):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); tiny.add_(1.0); e1.record()
    torch.cuda.synchronize()
    floor_ms.append(e0.elapsed_time(e1))
  dispatch_floor_ms = float(np.median(floor_ms))
  nattr = min(attribution_steps, max(warmup, 1))
  for v in kern.values():
    v['total_ms'] /= nattr; v['calls'] /= nattr    # (true division in both places: step_roofline prices group calls)
    for g in v['groups']:
      g['calls'] /= nattr
  # dominant kernel = largest share of the step in that pass (averaged over the attribution steps so that two kernels
  # a few microseconds apart do not swap places between runs), among the MFMA kernels and the HBM kernels that move
  # >= 32 MB, averaging >= 50 us per launch and launched at most 64 times per step: event pairs around
  # few-microsecond kernels launched hundreds of times per step are not a reliable ranking
  big = [k for k in kern if (kern[k]['flops'] > 0 or kern[k]['bytes'] >= (1 << 25)) and kern[k]['avg_ms'] >= 0.05
         and kern[k]['calls'] <= 64]
  big = big or list(kern)
  dominant = max(big, key=lambda k: kern[k]['total_ms'])

  # ---- timed region: exactly K steps, barrier + sync on both sides ----
  step_fn = lambda: lrn.minimize(unroll, *extra)
  mode = 'eager'
  gs = None
  if graph:
    try:
      gs = learner.GraphedStep(lrn, unroll, *extra, warmup=1)
      step_fn, mode = (lambda: gs()), 'hip-graph'
      if gs.split:
        mode = 'hip-graph x%d segments + eager RCCL exchange + update graph' % len(gs.segments)
      # untimed replays of the captured graph: the first window after a capture ran 8 % slower than every later one
      # (0.993 against 0.914-0.922 ms, six windows, r6) -- the warm-up the contract asks for has to warm THIS launch path,
      # not only the eager one the W steps above went through.  Ten replays (9 ms) still left window 0 3-4 % above the
      # others (0.901 / 0.894 against 0.865-0.880); fifty (45 ms) put it among them (0.877 / 0.873 against 0.878-0.889, same
      # box, alternating runs: the later windows are the chip's steady state, slightly SLOWER than a cold start's).
      for _ in range(max(warmup, int(os.environ.get('SEEDRL_BENCH_GRAPH_WARMUP', '50')))):
        step_fn()
      torch.cuda.synchronize()
    except Exception as e:                       # pylint: disable=broad-except
      sys.stderr.write('HIP-graph capture unavailable (%s); timing eager launches\n' % e)
      if world > 1:
        raise                                    # ranks must not diverge into different launch modes
      gs, step_fn, mode = None, (lambda: lrn.minimize(unroll, *extra)), 'eager'
  prof = ops.Profiler(only=[dominant])
  if mode == 'eager':
    ops.set_profiler(prof)
  barrier()
  t0 = time.perf_counter()
  for _ in range(steps):
    out = step_fn()
  barrier()
  dt = time.perf_counter() - t0
  ops.set_profiler(None)
  loss = out[0]
  # The contract's window above is what `value` / `ms_per_step` are computed from.  Behind it, extra_windows more windows
  # of the SAME K steps with the same brackets: boxes of the pool differ by up to 18 % and one 19 ms window says nothing
  # about its own spread -- the line carries median / min / max over all of them.
  window_ms = [dt / steps * 1e3]
  for _ in range(extra_windows if mode != 'eager' else 0):
    barrier()
    tw = time.perf_counter()
    for _ in range(steps):
      step_fn()
    barrier()
    window_ms.append((time.perf_counter() - tw) / steps * 1e3)
  if distributed and len(window_ms) > 1:
    wt = torch.tensor(window_ms, device=dev, dtype=torch.float64)
    torch.distributed.all_reduce(wt, op=torch.distributed.ReduceOp.MAX)
    window_ms = [float(x) for x in wt]
  if mode != 'eager':
    # free-running kernel time of the dominant kernel: HIP events around it on the launch stream, in a few eager
    # steps after the timed region (events cannot be recorded inside a replayed graph)
    ops.set_profiler(prof)
    for _ in range(min(steps, 5)):
      lrn.minimize(unroll, *extra)
    ops.set_profiler(None)
  exchange = None
  if world > 1 or (force_exchange and distributed):
    # per-rank wall time of the timed region, then its maximum (the contract's clock)
    mine = torch.tensor([dt], device=dev, dtype=torch.float64)
    every = [torch.zeros_like(mine) for _ in range(world)]
    torch.distributed.all_gather(every, mine)
    per_rank = [float(t[0]) for t in every]
    dt = max(per_rank)
    # what the exchange costs: the same K steps with the all-reduce left out (replicas diverge from here on -- this
    # is the last thing the run does with them).  exposed = (step with exchange) - (step without).
    bucket = agent.flat.grads.numel() * 4
    exposed = None
    if gs is not None:
      gs.exchange = False
      for _ in range(2):
        step_fn()
      barrier()
      t1 = time.perf_counter()
      for _ in range(steps):
        step_fn()
      barrier()
      tt = torch.tensor([time.perf_counter() - t1], device=dev, dtype=torch.float64)
      torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
      exposed = max(0.0, dt - float(tt[0])) / steps * 1e3
      gs.exchange = True
    exchange = dict(ranks=world, backend=torch.distributed.get_backend(), bucket_bytes=bucket,
                    per_rank_ms_per_step=[round(t / steps * 1e3, 4) for t in per_rank],
                    exposed_ms_per_step=None if exposed is None else round(exposed, 4),
                    overlapped_ranges=[list(r) for _, r in gs.segments if r] if gs is not None and gs.split else [],
                    note='exposed = ms_per_step minus the same steps replayed without the all-reduce; the range listed '
                         'under overlapped_ranges (float offsets into the flat gradient bucket) is exchanged on the '
                         'collective stream under the rest of the backward pass, the remainder after it')
  loss_val = float(loss)
  assert np.isfinite(loss_val)
  if hasattr(agent, 'check_errors'):
    agent.check_errors()

  free = prof.summary().get(dominant) or dict(avg_ms=0.0)
  d = kern[dominant]
  if not free.get('avg_ms', 0.0) > 0.0:          # tiny shapes: an event pair around a few-microsecond kernel can read 0
    free = dict(free, avg_ms=d['avg_ms'])
  if not d['avg_ms'] > 0.0:
    d = dict(d, avg_ms=1e-6); free = dict(free, avg_ms=1e-6)
  flops, nbytes = d['flops'], d['bytes']
  # the roofline that BOUNDS the kernel: the larger of its two floors at the spec peaks (r5: the first conv's kernels have
  # flops, but their 351 MB at 8 TB/s take longer than their 35 GF on the bf16x3 ceiling -- they are priced against HBM)
  mfma_bound = flops > 0 and flops / (_peak(d)[0] * 1e12) >= nbytes / (HBM_PEAK_GBS * 1e9)
  if mfma_bound:
    peak, pipe_desc = _peak(d)
    ach, ach_free = flops / (d['avg_ms'] * 1e-3) / 1e12, flops / (free['avg_ms'] * 1e-3) / 1e12
    roofline = dict(bound='mfma', kernel=dominant, achieved=round(ach, 2), peak=round(peak, 1),
                    unit='TFLOP/s', frac=round(ach / peak, 4), traffic=None,
                    pipe=pipe_desc,
                    algorithmic_flops=flops, algorithmic_bytes=nbytes)
  else:
    peak = HBM_PEAK_GBS
    ach, ach_free = nbytes / (d['avg_ms'] * 1e-3) / 1e9, nbytes / (free['avg_ms'] * 1e-3) / 1e9
    roofline = dict(bound='hbm', kernel=dominant, achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit='GB/s',
                    frac=round(ach / HBM_PEAK_GBS, 4), traffic=None, algorithmic_bytes=nbytes)
    if flops > 0:                                            # (its matrix-pipe side, for reference)
      roofline.update(algorithmic_flops=flops, mfma_tflops=round(flops / (d['avg_ms'] * 1e-3) / 1e12, 2),
                      mfma_frac=round(flops / (d['avg_ms'] * 1e-3) / 1e12 / _peak(d)[0], 4), pipe=_peak(d)[1])
  # two clocks, both reported: `frac` / `frac_serialized` divide by the kernel's duration with the device drained before
  # it (what rocprofv3 --kernel-trace --stats shows: profiles/ must agree with THIS one); `frac_free_running` by its
  # duration in back-to-back steps, where it starts behind its producer with operands still in L2 / MALL
  roofline.update(frac_serialized=round(ach / peak, 4), avg_kernel_ms=round(d['avg_ms'], 4),
                  dispatch_floor_ms=round(dispatch_floor_ms, 4),
                  avg_kernel_ms_less_dispatch=round(max(d['avg_ms'] - dispatch_floor_ms, 0.0), 4),
                  frac_free_running=round(ach_free / peak, 4), avg_kernel_ms_free_running=round(free['avg_ms'], 4),
                  clock='HIP events on the launch stream; serialized = device drained before the kernel (rocprofv3 view)')
  rec = dict(
      T=T, B=B, A=A, workload=workload, mode=mode, params=agent.flat.num_params(), loss=loss_val,
      ms_per_step=dt / steps * 1e3, frames_per_s=world * B * T / (dt / steps), roofline=roofline, dominant=dominant,
      windows=dict(count=len(window_ms), steps_each=steps, ms_per_step=[round(x, 4) for x in window_ms],
                   median=round(float(np.median(window_ms)), 4), min=round(min(window_ms), 4), max=round(max(window_ms), 4),
                   note='window 0 is the contract\'s timed region (value / ms_per_step); the others repeat it'),
      step_roofline=step_roofline(kern, dt / steps * 1e3),
      kernels_ms_per_step={k: round(v['total_ms'], 4) for k, v in kern.items()}, exchange=exchange,
      # the other MFMA kernels of the attribution pass (>= 50 us per launch), same (serialized) accounting as `roofline`
      mfma_kernels={
          k: dict(avg_ms=round(v['avg_ms'], 4), tflops=round(v['flops'] / (v['avg_ms'] * 1e-3) / 1e12, 1),
                  frac=round(v['flops'] / (v['avg_ms'] * 1e-3) / 1e12 / _peak(v)[0], 3), pipe=v.get('pipe', 'f32'))
          for k, v in kern.items() if v['flops'] > 0 and v['avg_ms'] >= 0.05})
  del agent, lrn, unroll, extra
  _release()
  return rec


def inference_record(dev, sizes=(64, 256, 1024), unroll_len=20, calls=200):
  """Central inference (agents/vtrace/learner.py:350-405) on the device-resident store, Atari agent: env steps per
  second through FusedInferenceState for a few inference batch sizes, the whole call replayed from a HIP graph
  (run-id bookkeeping, single-step agent forward, action sampling, store append, completed unrolls written
  time-major into the training batch).  Inputs (one request batch) are copied into the graph's static buffers on every
  call, as the transport layer would."""
  from seed_rl_amd import inference, networks, utils
  from seed_rl_amd.unroll_store import Spec
  A, obs = 18, (84, 84, 1)
  out = {}
  for n in sizes:
    envs = max(512, 2 * n)
    agent = networks.AtariShallow(A, device=dev)
    env_specs = utils.EnvOutput(Spec((), torch.float32), Spec((), torch.bool), Spec(obs, torch.uint8),
                                Spec((), torch.bool), Spec((), torch.int32))
    ao_specs = networks.AgentOutput(Spec((), torch.int64), Spec((A,), torch.float32), Spec((), torch.float32))
    g = torch.Generator(device='cpu').manual_seed(0)
    groups = [torch.arange(i, i + n, dtype=torch.int64).to(dev) for i in range(0, envs, n)]
    reqs = [utils.EnvOutput(
        reward=torch.randn(n, generator=g).to(dev), done=(torch.rand(n, generator=g) < 0.01).to(dev),
        observation=torch.randint(0, 256, (n,) + obs, dtype=torch.uint8, generator=g).to(dev),
        abandoned=torch.zeros(n, dtype=torch.bool, device=dev), episode_step=torch.zeros(n, dtype=torch.int32, device=dev))
            for _ in groups]
    run_ids = np.full((n,), 7, np.int64)
    # one packed request batch per group, as a transport front-end delivers it (inference.request_layout)
    packed = [torch.from_numpy(inference.pack_request(
        n, g_.cpu().numpy(), run_ids, r.reward.cpu().numpy(), r.reward.cpu().numpy(), r.done.cpu().numpy())).to(dev)
              for g_, r in zip(groups, reqs)]
    fused = inference.FusedInferenceState(agent, envs, unroll_len, env_specs, ao_specs, batch_capacity=2 * envs, device=dev)
    fn = fused.graphed(n, obs)
    per_round = len(groups) * (unroll_len + 1)

    def call(i):
      k = i % len(groups)
      fn.replay_packed(packed[k], reqs[k].observation)
      if (i + 1) % per_round == 0:
        fused.batch_count.zero_()             # the learner took the filled training batch
    for i in range(2 * len(groups)):
      call(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(calls):
      call(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    fused.check_errors()
    out['n%d' % n] = dict(us_per_call=round(dt / calls * 1e6, 1), env_steps_per_s=round(calls * n / dt, 0), envs=envs)
    del fused, fn, agent, reqs, groups
    _release()
  out['note'] = ('FusedInferenceState.graphed().replay_packed(): per inference batch two copies (packed request scalars, '
                 'frames) into the static inputs + one HIP-graph replay (bookkeeping, agent forward, in-kernel action '
                 'sampling, store append, completed unrolls -> training batch); Atari shallow agent, A=18, unroll 20')
  return out


def r2d2_replay_record(dev, steps=5, T=120, B=256, A=18, burn_in=40, replay_size=2048):
  """cfg5 END TO END (agents/r2d2/learner.py:387-468, 856-885): every iteration inserts freshly completed unrolls
  (B / replay_ratio of them, replay_ratio 1.5: learner.py:62-66) with initial priorities, samples a prioritized batch
  time-major out of the device replay, trains and writes the new priorities back (r2d2_loop.ReplayTrainer)."""
  from seed_rl_amd import networks, optimizers, r2d2_learner, r2d2_loop
  T1 = T + 1
  agent = networks.DuelingLSTMDQNNet(A, device=dev, seed=0)
  target = networks.DuelingLSTMDQNNet(A, device=dev, seed=0)
  cfg = r2d2_learner.R2D2Config(burn_in=burn_in)
  lrn = r2d2_learner.R2D2Learner(agent, target, optimizers.Adam(4.8e-4, epsilon=1e-3), cfg)
  specs = r2d2_loop.unroll_specs(agent, T - burn_in, burn_in, (84, 84, 1), A)
  tr = r2d2_loop.ReplayTrainer(lrn, specs, replay_buffer_size=replay_size, replay_buffer_min_size=2 * B,
                               batch_size=B, device=dev)
  g = torch.Generator(device='cpu').manual_seed(0)

  def fresh(n):
    from seed_rl_amd import utils
    env = utils.EnvOutput(torch.randn((T1, n), generator=g).to(dev), (torch.rand((T1, n), generator=g) < 0.01).to(dev),
                          torch.randint(0, 256, (T1, n, 84, 84, 1), dtype=torch.uint8, generator=g).to(dev),
                          torch.zeros((T1, n), dtype=torch.bool, device=dev),
                          torch.zeros((T1, n), dtype=torch.int32, device=dev))
    ao = networks.R2D2AgentOutput(torch.randint(0, A, (T1, n), generator=g).to(dev),
                                  torch.rand((T1, n, A), generator=g).to(dev))
    u = r2d2_learner.Unroll(agent.initial_state(n), None, torch.randint(0, A, (T1, n), generator=g).to(dev), env, ao)
    return u._replace(priority=r2d2_loop.initial_priorities(agent, u, burn_in, cfg))
  n_ins = int(round(B / 1.5))
  tr.insert(fresh(2 * B))
  new = fresh(n_ins)
  for _ in range(2):
    tr.insert(new); tr.train_step()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(steps):
    tr.insert(new)
    loss, _, _, _ = tr.train_step()
  torch.cuda.synchronize()
  dt = (time.perf_counter() - t0) / steps
  rec = dict(ms_per_iteration=round(dt * 1e3, 3), env_frames_per_s=round(B * T / dt, 1), loss=round(float(loss), 6),
             inserted_per_iteration=n_ins, replay_unrolls=replay_size, launch='eager',
             note='insert (initial priorities) + prioritized sample, time-major out of the HBM replay + train step + '
                  'update_priorities per iteration; the plain train step on a resident batch is ms_per_step above')
  del agent, target, lrn, tr
  _release()
  return rec


def ingest_record(dev, steps, T=20, B=512, A=18):
  """The cfg2 step fed from PINNED HOST memory: the whole unroll of step i+1 (frames = 98.7 % of its bytes) is copied
  host->device on a copy stream while step i computes (double-buffered device unrolls, one HIP graph per buffer)."""
  from seed_rl_amd import learner, networks, optimizers, parametric_distribution as pd, smoke_step, utils
  T1 = T + 1
  agent = networks.AtariShallow(A, device=dev, seed=0)
  opt = optimizers.Adam(optimizers.PolynomialDecay(4.8e-4, 10 ** 5), beta_1=0.0, epsilon=3.125e-7, capturable=True)
  lrn = learner.Learner(agent, opt, pd.categorical_distribution(A))
  slots, hosts = [], []
  for s in range(2):
    agent.frames_slot = s
    u = smoke_step.make_unroll(agent, T1, B, A, dev, seed=2000 + s)
    ext = agent.frames_buffer(T1, B)
    ext[3:].copy_(u.env_outputs.observation.reshape(T1, B, -1))
    u = u._replace(env_outputs=u.env_outputs._replace(observation=ext[3:].view(T1, B, agent._obs[0], agent._obs[1], 1)))
    leaves = [t for t in utils.flatten(u) if t is not None and t.numel() > 0]
    slots.append((u, leaves))
    hosts.append([t.cpu().pin_memory() for t in leaves])
  nbytes = sum(t.numel() * t.element_size() for t in hosts[0])
  graphed = []                                        # one captured step per device unroll (its frames buffer is static)
  for s in range(2):
    agent.frames_slot = s
    graphed.append(learner.GraphedStep(lrn, slots[s][0]))
  copy_stream = torch.cuda.Stream()
  ready = [torch.cuda.Event(), torch.cuda.Event()]
  free = [torch.cuda.Event(), torch.cuda.Event()]
  cur = torch.cuda.current_stream()

  def prefetch(k):
    with torch.cuda.stream(copy_stream):
      copy_stream.wait_event(free[k & 1])
      for d, h in zip(slots[k & 1][1], hosts[k & 1]):
        d.copy_(h, non_blocking=True)
      ready[k & 1].record(copy_stream)

  def run(n):
    free[0].record(cur); free[1].record(cur)
    prefetch(0)
    for i in range(n):
      agent.frames_slot = i & 1
      cur.wait_event(ready[i & 1])
      prefetch(i + 1)
      graphed[i & 1]()
      free[i & 1].record(cur)
    torch.cuda.synchronize()

  run(3)
  t0 = time.perf_counter()
  run(steps)
  dt = time.perf_counter() - t0
  # the copy alone, for reference
  torch.cuda.synchronize()
  t1 = time.perf_counter()
  for k in range(4):
    prefetch(k)
  torch.cuda.synchronize()
  copy_ms = (time.perf_counter() - t1) / 4 * 1e3
  agent.frames_slot = 0
  rec = dict(mode='pinned host unrolls, H2D of step i+1 on a copy stream under step i (HIP-graph replay per device buffer)',
             ms_per_step=round(dt / steps * 1e3, 4), env_frames_per_s=round(B * T / (dt / steps), 1),
             h2d_bytes_per_step=nbytes, h2d_ms_alone=round(copy_ms, 4), h2d_GBs=round(nbytes / copy_ms / 1e6, 1))
  del agent, lrn, slots, hosts
  _release()
  return rec


def _free_port():
  import socket
  so = socket.socket()
  so.bind(('127.0.0.1', 0))
  port = so.getsockname()[1]
  so.close()
  return port


def self_launch(args):
  """`python bench.py --gpus N` without a launcher: re-executes this script as N ranks under torch.distributed.run
  (one process per GPU, rendezvous on 127.0.0.1) with the same arguments and hands back its exit code, so that the
  printed line is the N-rank measurement -- or fails loudly when the box has fewer than N devices."""
  import subprocess
  ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
  if ndev < args.gpus and args.backend != 'gloo':
    sys.stderr.write('bench.py: --gpus %d asked for but this box exposes %d GPU(s): RCCL needs one device per rank.  '
                     '(--backend gloo runs the %d-rank code path as a dry run on the devices there are.)\n'
                     % (args.gpus, ndev, args.gpus))
    return 2
  if ndev < 1:
    sys.stderr.write('bench.py needs a GPU (no CPU fallback for the HIP path)\n')
    return 2
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
         '--master-addr', '127.0.0.1', '--master-port', str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
  env = dict(os.environ)
  env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
  env.setdefault('OMP_NUM_THREADS', '8')
  return subprocess.call(cmd, env=env)


def main():
  args = parse()
  if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
    sys.exit(self_launch(args))
  rank = int(os.environ.get('RANK', '0'))
  world = int(os.environ.get('WORLD_SIZE', '1'))
  local_rank = int(os.environ.get('LOCAL_RANK', '0'))
  if world != args.gpus:
    sys.stderr.write('bench.py: --gpus %d but the launcher started %d rank(s)\n' % (args.gpus, world))
    sys.exit(2)
  if not torch.cuda.is_available():
    sys.stderr.write('bench.py needs a GPU (no CPU fallback for the HIP path)\n')
    sys.exit(2)
  ndev = torch.cuda.device_count()
  if local_rank >= ndev and args.backend != 'gloo':
    sys.stderr.write('bench.py: rank %d has no device (%d GPU(s) visible, %d ranks): RCCL needs one device per rank\n'
                     % (rank, ndev, world))
    sys.exit(2)
  dev_index = local_rank % ndev
  torch.cuda.set_device(dev_index)
  dev = torch.device('cuda', dev_index)
  distributed = world > 1 or 'RANK' in os.environ or args.force_exchange   # launched by torch.distributed.run (also at N=1)
  if distributed:
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29500')
    os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
    if args.backend == 'gloo':
      torch.distributed.init_process_group('gloo')
    else:
      torch.distributed.init_process_group('nccl', device_id=dev)        # "nccl" is RCCL on ROCm
  deep, r2 = args.config == 'dmlab', args.config == 'r2d2'

  rec = run_learner(args.config, args.steps, args.warmup, dev, rank, world, distributed, args.torso, args.batch,
                    args.unroll, args.actions, args.reduction, args.graph, force_exchange=args.force_exchange,
                    extra_windows=args.windows - 1)
  T, B, A = rec['T'], rec['B'], rec['A']
  roofline = rec['roofline']
  headline = args.config == 'atari' and args.torso == 'shallow' and B == 512 and T == 20 and A == 18

  # HBM traffic of the dominant kernel from the committed PMC passes (same config only; latest profiling round)
  tfiles = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_cfg2_traffic.json')))
  if headline and tfiles:
    from seed_rl_amd import build as _build
    tj = json.load(open(tfiles[-1]))
    tb = tj['traffic_bytes']
    here = _build.csrc_digest()
    fresh = tj.get('csrc_sha256') == here
    stamp = 'profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes; taken at git %s, kernel sources sha256 %s)' % (
        os.path.basename(tfiles[-1]), (tj.get('git_sha') or 'unrecorded')[:12], (tj.get('csrc_sha256') or 'unrecorded')[:12])
    if not fresh:
      # the committed counters describe OTHER kernels than the ones this run timed: say so instead of printing them
      roofline['traffic'] = None
      roofline['traffic_source'] = stamp + ' -- STALE: this tree\'s kernel sources hash to %s; traffic withheld' % here[:12]
    elif rec['dominant'] in tb:
      roofline['traffic'] = tb[rec['dominant']]
      roofline['traffic_source'] = stamp
    if fresh and rec['dominant'] in tj.get('mfma_pmc', {}):
      # the matrix pipe's busy share from the SQ counters of the committed profiling round (same kernel, same shape)
      roofline['mfma_pmc'] = tj['mfma_pmc'][rec['dominant']]
  if rank != 0:
    if distributed:
      torch.distributed.destroy_process_group()
    return
  result = {
      'metric': 'learner env-frames/s (T=%d)' % T, 'value': round(rec['frames_per_s'], 1), 'unit': 'env-frames/s',
      'n_gpus': min(world, ndev), 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(rec['ms_per_step'], 4),
      'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
      'config': {'workload': '%s, T=%d B=%d/GPU A=%d, synthetic uint8 frames in HBM, num_action_repeats=1'
                             % (rec['workload'], T, B, A),
                 'global_batch': B * world, 'unroll_length': T, 'parallelism': 'dp%d' % world,
                 'grad_reduction': args.reduction, 'params': rec['params'], 'launch': rec['mode'],
                 'ingest': 'resident (unroll in HBM before the timed region)',
                 'process_group': (('rccl' if args.backend == 'nccl' else 'gloo') if distributed else None)},
      'windows': rec['windows'],
      'roofline': roofline,
      'step_roofline': rec['step_roofline'],
      'exchange': rec['exchange'],
      'loss': round(rec['loss'], 6),
      'kernels_ms_per_step': rec['kernels_ms_per_step'],
      'mfma_kernels': rec['mfma_kernels'],
  }
  if world > ndev:
    result['dry_run'] = ('%d ranks share %d device(s) over gloo: exercises the N-rank code path (sharded columns, graph '
                         'segments, overlapped exchange, update graph); the throughput is NOT a scaling measurement'
                         % (world, ndev))
  if world == 1:
    err, err_ref, sweep = vtrace_checks(dev)
    result['vtrace_max_abs_err'] = err
    result['vtrace_max_abs_err_vs_reference_code'] = err_ref
    result['vtrace_scan_hbm'] = sweep
  if world == 1 and not args.quick:
    from tests import parity
    # one train step at the bench shape against the CPU oracle (outside the timed region; same code as
    # tests/test_gpu_fullsize.py)
    # (fp32 oracle only: its fp64 evaluation -- tests/test_gpu_fullsize.py -- takes minutes of host time at this size)
    if r2:
      p = parity.r2d2_step(dev, T1=T + 1, B=B, A=A, truth=False)
    elif deep:
      p = parity.deep_step(dev, T1=T + 1, B=B, A=A, truth=False)
    else:
      p = parity.atari_step(dev, T1=T + 1, B=B, A=A, torso=args.torso, truth=False)
    result['parity'] = dict(parity.public(p), note=PARITY_NOTE)
    _release()
    if headline:
      others = {}
      for name, cfg in (('dmlab', 'dmlab'), ('r2d2', 'r2d2')):
        try:
          r = run_learner(cfg, 5, 3, dev, graph=args.graph, attribution_steps=1)
          others[name] = dict(
              workload='%s, T=%d B=%d A=%d' % (r['workload'], r['T'], r['B'], r['A']), steps=5,
              ms_per_step=round(r['ms_per_step'], 3), env_frames_per_s=round(r['frames_per_s'], 1), launch=r['mode'],
              roofline={k: r['roofline'][k] for k in ('kernel', 'bound', 'achieved', 'peak', 'unit', 'frac',
                                                      'frac_free_running', 'avg_kernel_ms')},
              step_roofline={k: r['step_roofline'][k] for k in ('floor_ms', 'frac')},
              loss=round(r['loss'], 6))
          # one train step at THAT shape against the CPU oracle (tests/test_gpu_fullsize.py runs the same comparison)
          pf = parity.deep_step if cfg == 'dmlab' else parity.r2d2_step
          pr = parity.public(pf(dev, T1=r['T'] + 1, B=r['B'], A=r['A'], truth=False))
          if cfg == 'r2d2':
            try:
              others[name]['replay_loop'] = r2d2_replay_record(dev)
            except Exception as e:               # pylint: disable=broad-except
              others[name]['replay_loop'] = dict(error=repr(e))
          others[name]['parity'] = {k: pr[k] for k in (
              'loss', 'loss_ref', 'loss_rel_err', 'logits_max_abs_err', 'baseline_max_abs_err', 'q_max_abs_err',
              'priority_max_rel_err', 'grad_q99_rel_err', 'grad_max_rel_err', 'grad_worst', 'grad_norm_rel_err',
              'param_max_abs_err', 'oracle_s', 'shape') if k in pr}
          _release()
        except Exception as e:                   # pylint: disable=broad-except
          others[name] = dict(others.get(name, {}), error=repr(e))
      result['other_configs'] = others
      try:
        result['inference'] = inference_record(dev)
      except Exception as e:                     # pylint: disable=broad-except
        result['inference'] = dict(error=repr(e))
      try:
        result['ingest'] = ingest_record(dev, args.steps)
      except Exception as e:                     # pylint: disable=broad-except
        result['ingest'] = dict(error=repr(e))
      # inference and training TOGETHER on this GPU (tools/bench_serving.py): the learner trains on exactly the steps
      # central inference serves, so both rates are one number; once without a transport, once through the native gRPC
      # front-end with actor processes on the host cores
      try:
        sys.path.insert(0, os.path.join(ROOT, 'tools'))
        import bench_serving
        serving = dict(inprocess=bench_serving.run_inprocess(dev, seconds=3.0, n=4096, envs=16384))
        _release()
        try:
          procs = 32 if (os.cpu_count() or 8) >= 64 else 8
          serving['transport'] = bench_serving.run_transport(dev, seconds=3.0, n=2048 if procs == 32 else 256,
                                                             procs=procs, envs_per_proc=256 if procs == 32 else 64)
        except Exception as e:                   # pylint: disable=broad-except
          serving['transport'] = dict(error=repr(e))
        serving['note'] = ('closed loop on ONE GPU: inference batches on a high-priority stream (inference twin of the '
                           'agent, same parameters), train step graph on its own stream, dequeue as the only ordered '
                           'hand-over; learner alone = `value`, inference alone = `inference`')
        result['serving'] = serving
        _release()
      except Exception as e:                     # pylint: disable=broad-except
        result['serving'] = dict(error=repr(e))
  elif world == 1 and args.ingest == 'pinned' and headline:
    result['ingest'] = ingest_record(dev, args.steps)
  if world == 1 and not args.no_cpu_baseline and not args.quick:
    from oracle import cpu_learner
    T1 = T + 1
    # BASELINE.md section 3 protocol: 3 warm-up steps, median of >= 10 timed steps, all host cores
    CW, CS = 3, 10
    if r2:
      cb = args.cpu_batch or 8
      fps, sec, thr = cpu_learner.time_cpu_r2d2_learner(A, T1, cb, steps=CS, warmup=CW)
    elif deep:
      cb = args.cpu_batch or 16
      fps, sec, thr = cpu_learner.time_cpu_deep_learner(A, T1, cb, steps=CS, warmup=CW)
    else:
      cb = args.cpu_batch or B              # the quoted configuration itself (B=512: ~2 s per step on 128 threads)
      kind = 'atari_shallow' if args.torso == 'shallow' else 'atari_dqn_body'
      fps, sec, thr = cpu_learner.time_cpu_learner(kind, A, T1, cb, steps=CS, warmup=CW)
    result['cpu_baseline'] = {
        'value': round(fps, 1), 'unit': 'env-frames/s', 'cores': thr, 'kind': 'port',
        'sample': 'same learner step as eager PyTorch-CPU fp32 restatement of the reference graph '
                  '(oracle/cpu_learner.py; NOT the reference\'s TF graph), T=%d B=%d (%s), '
                  '%d warm-up steps then the median of %d timed steps (BASELINE.md section 3), %.2f s/step; '
                  'host cpu_count=%d, torch threads=%d' % (
                      T, cb, 'the benched configuration itself' if cb == B else 'a bounded sample of the workload: whole columns, same T',
                      CW, CS, sec, os.cpu_count(), thr)}
    result['speedup_vs_cpu_baseline'] = round(rec['frames_per_s'] / fps, 1)
  # the numbers the driver's tail (last ~2000 characters of stdout) must keep: repeated compactly as the LAST key
  tail = {'cfg2': dict(ms_per_step=result['ms_per_step'], env_frames_per_s=result['value'],
                       dominant=roofline['kernel'], frac=roofline['frac'], step_frac=rec['step_roofline']['frac'],
                       step_floor_ms=rec['step_roofline']['floor_ms'])}
  for name, key in (('dmlab', 'cfg3'), ('r2d2', 'cfg5')):
    o = result.get('other_configs', {}).get(name)
    if o and 'ms_per_step' in o:
      tail[key] = dict(ms_per_step=o['ms_per_step'], env_frames_per_s=o['env_frames_per_s'],
                       dominant=o['roofline']['kernel'], frac=o['roofline']['frac'], step_frac=o.get('step_roofline', {}).get('frac'))
    elif o:
      tail[key] = dict(error=o.get('error'))
  sv = result.get('serving') or {}
  if 'inprocess' in sv:
    tail['serving'] = dict(inprocess=sv['inprocess'].get('env_steps_per_s_served'),
                           transport=(sv.get('transport') or {}).get('env_steps_per_s_served'))
  if 'cpu_baseline' in result:
    tail['cpu_baseline'] = dict(value=result['cpu_baseline']['value'], B=cb, speedup=result['speedup_vs_cpu_baseline'])
  if result.get('exchange'):
    tail['exchange'] = dict(backend=result['exchange']['backend'], ranks=result['exchange']['ranks'],
                            exposed_ms_per_step=result['exchange']['exposed_ms_per_step'])
  result['tail_summary'] = tail
  # the process group is torn down BEFORE the line is printed: RCCL writes to stdout on teardown ("Librccl path : ...")
  # and the JSON line must stay the last line
  if distributed:
    torch.distributed.destroy_process_group()
  sys.stdout.flush()
  print(json.dumps(result), flush=True)


if __name__ == '__main__':
  main()
