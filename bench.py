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
"""
import argparse
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
MFMA_F32_PEAK_TF = 157.3   # fp32-input MFMA dense peak (v_mfma_f32_16x16x4_f32)
MFMA_BF16_PEAK_TF = 2500.0 # bf16 dense MFMA peak (MI355X_MICROARCH.md; no sparsity)
# Kernels that evaluate their fp32 algorithm on the bf16 pipe through the exact three-way operand split
# (csrc/stackconv.hip): every algorithmic MAC costs three bf16 MACs, so their ceiling for ALGORITHMIC flops is peak / 3.
BF16X3_KERNELS = ('stack_conv_fwd', 'stack_conv_wgrad')


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
  ap.add_argument('--cpu-batch', type=int, default=64)
  ap.add_argument('--reduction', default='mean', choices=['mean', 'sum'])
  ap.add_argument('--graph', type=int, default=1,
                  help='1 (default, single GPU): replay the step from a captured HIP graph (learner.GraphedStep; measured '
                       '0-2%% over eager on MI355X); 0: eager launches.  Multi-GPU runs launch eagerly.')
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


def main():
  args = parse()
  rank = int(os.environ.get('RANK', '0'))
  world = int(os.environ.get('WORLD_SIZE', '1'))
  local_rank = int(os.environ.get('LOCAL_RANK', '0'))
  assert torch.cuda.is_available(), 'bench.py needs a GPU (no CPU fallback for the HIP path)'
  torch.cuda.set_device(local_rank)
  dev = torch.device('cuda', local_rank)
  distributed = world > 1 or 'RANK' in os.environ       # launched by torch.distributed.run (also at N=1)
  if distributed:
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29500')
    os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
    torch.distributed.init_process_group('nccl', device_id=dev)          # "nccl" is RCCL on ROCm
  assert world == args.gpus or world == 1, 'launch with torchrun for --gpus > 1'

  from seed_rl_amd import learner, networks, ops, optimizers, parametric_distribution as pd, smoke_step
  deep, r2 = args.config == 'dmlab', args.config == 'r2d2'
  T = args.unroll or (120 if r2 else 20)
  B = args.batch or (256 if (deep or r2) else 512)
  A = args.actions or (9 if deep else 18)
  T1 = T + 1
  final_iteration = 10 ** 9 // (T * B * max(world, 1))
  opt = optimizers.Adam(optimizers.PolynomialDecay(4.8e-4, final_iteration), beta_1=0.0, epsilon=3.125e-7,
                        capturable=bool(args.graph))
  if r2:
    from seed_rl_amd import r2d2_learner
    agent = networks.DuelingLSTMDQNNet(A, device=dev, seed=0)
    target = networks.DuelingLSTMDQNNet(A, device=dev, seed=0)
    u0 = smoke_step.make_unroll(agent, T1, B, A, dev, seed=1000 + rank)
    unroll = r2d2_learner.Unroll(agent.initial_state(B), None, u0.prev_actions, u0.env_outputs,
                                 networks.R2D2AgentOutput(u0.agent_outputs.action.to(torch.int32), None))
    iw = torch.rand(B, device=dev) * 0.9 + 0.1
    workload = 'Atari R2D2 DuelingLSTMDQNNet (conv 32/64/64, FC512, LSTM512, dueling heads) learner step on replayed ' \
               'sequences, burn-in 40, n-step(5) double-Q target, training + target network'
  elif deep:
    agent = networks.ImpalaDeep(A, device=dev, seed=0)                          # identical params on all ranks
    unroll = smoke_step.make_deep_unroll(agent, T1, B, A, dev, seed=1000 + rank)
    workload = 'DeepMind Lab 72x96x3 IMPALA deep ResNet + LSTM(256) learner step'
  else:
    agent = networks.AtariShallow(A, torso=args.torso, device=dev, seed=0)
    unroll = smoke_step.make_unroll(agent, T1, B, A, dev, seed=1000 + rank)
    # place the frames directly in the agent's extended trajectory buffer (no per-step copy)
    ext = agent.frames_buffer(T1, B)
    ext[3:].copy_(unroll.env_outputs.observation.reshape(T1, B, -1))
    unroll = unroll._replace(env_outputs=unroll.env_outputs._replace(
        observation=ext[3:].view(T1, B, agent._obs[0], agent._obs[1], 1)))
    workload = 'Atari 84x84x4 IMPALA %s ConvNet learner step' % args.torso
  if r2:
    opt = optimizers.Adam(4.8e-4, epsilon=1e-3, capturable=bool(args.graph))    # atari/r2d2_main.py:36-39
    r2l = r2d2_learner.R2D2Learner(agent, target, opt, r2d2_learner.R2D2Config(), reduction=args.reduction)

    class _Step(object):
      def minimize(self, unroll):
        total, _, _ = r2l.minimize(unroll, iw)
        return total, None
    lrn = _Step()
  else:
    lrn = learner.Learner(agent, opt, pd.categorical_distribution(A), reduction=args.reduction)

  def barrier():
    if distributed:
      torch.distributed.barrier()
    torch.cuda.synchronize()

  # ---- warmup (+ per-kernel profile pass to find the dominant kernel) ----
  for _ in range(max(args.warmup - 1, 0)):
    lrn.minimize(unroll)
  prof_all = ops.Profiler(serialize=True)
  ops.set_profiler(prof_all)
  lrn.minimize(unroll)
  ops.set_profiler(None)
  kern = prof_all.summary()
  # dominant kernel = largest share of the step in the attribution pass (device drained before every region),
  # among the MFMA kernels and the HBM kernels that move >= 32 MB, and only those averaging >= 50 us per launch and
  # launched at most 64 times per step:
  # event pairs around few-microsecond kernels launched hundreds of times per step (LSTM gates, the per-step
  # recurrent GEMM) are not a reliable ranking -- the rocprofv3 --stats tables under profiles/ are the reference
  # view and agree with this choice
  big = [k for k in kern if (kern[k]['flops'] > 0 or kern[k]['bytes'] >= (1 << 25)) and kern[k]['avg_ms'] >= 0.05
         and kern[k]['calls'] <= 64]
  big = big or list(kern)
  dominant = max(big, key=lambda k: kern[k]['total_ms'])
  if args.warmup == 0:
    dominant = sorted(kern)[0]

  # ---- timed region: exactly K steps, barrier + sync on both sides ----
  step_fn = lambda: lrn.minimize(unroll)
  mode = 'eager'
  if args.graph and world == 1:
    try:
      gs = learner.GraphedStep(r2l, unroll, iw, warmup=1) if r2 else learner.GraphedStep(lrn, unroll, warmup=1)
      step_fn, mode = (lambda: gs()), 'hip-graph'
      step_fn(); torch.cuda.synchronize()
    except Exception as e:                       # pylint: disable=broad-except
      sys.stderr.write('HIP-graph capture unavailable (%s); timing eager launches\n' % e)
      step_fn, mode = (lambda: lrn.minimize(unroll)), 'eager'
  prof = ops.Profiler(only=[dominant])
  if mode == 'eager':
    ops.set_profiler(prof)
  barrier()
  t0 = time.perf_counter()
  for _ in range(args.steps):
    out = step_fn()
  barrier()
  dt = time.perf_counter() - t0
  ops.set_profiler(None)
  loss = out[0]
  if mode != 'eager':
    # kernel time of the dominant kernel for the roofline: HIP events around it on the launch stream, in a
    # few eager steps after the timed region (events cannot be recorded inside a replayed graph)
    ops.set_profiler(prof)
    for _ in range(min(args.steps, 5)):
      lrn.minimize(unroll)
    ops.set_profiler(None)
  if world > 1:
    tt = torch.tensor([dt], device=dev, dtype=torch.float64)
    torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
    dt = float(tt[0])
  loss_val = float(loss)
  assert np.isfinite(loss_val)

  ms_per_step = dt / args.steps * 1e3
  frames_per_s = world * B * T / (dt / args.steps)
  d = prof.summary()[dominant]
  flops, nbytes = d['flops'], d['bytes']
  if flops > 0:
    ach = flops / (d['avg_ms'] * 1e-3) / 1e12
    bf16x3 = dominant in BF16X3_KERNELS and os.environ.get('SEEDHIP_STACK_BF16', '1') != '0'
    peak = round(MFMA_BF16_PEAK_TF / 3.0, 1) if bf16x3 else MFMA_F32_PEAK_TF
    roofline = dict(bound='mfma', kernel=dominant, achieved=round(ach, 2), peak=peak,
                    unit='TFLOP/s', frac=round(ach / peak, 4), traffic=None,
                    pipe=('bf16 MFMA, exact 3-way split of the fp32 operand: peak = 2500 / 3 algorithmic TFLOP/s'
                          if bf16x3 else 'fp32 MFMA (v_mfma_f32_16x16x4_f32)'),
                    avg_kernel_ms=round(d['avg_ms'], 4), algorithmic_flops=flops, algorithmic_bytes=nbytes)
  else:
    ach = nbytes / (d['avg_ms'] * 1e-3) / 1e9
    roofline = dict(bound='hbm', kernel=dominant, achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit='GB/s',
                    frac=round(ach / HBM_PEAK_GBS, 4), traffic=None, avg_kernel_ms=round(d['avg_ms'], 4),
                    algorithmic_bytes=nbytes)

  # HBM traffic of the dominant kernel from the committed PMC passes (same config only)
  tfiles = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_cfg2_traffic.json')))      # latest profiling round
  tpath = tfiles[-1] if tfiles else ''
  if args.config == 'atari' and args.torso == 'shallow' and B == 512 and T == 20 and A == 18 and tpath:
    tb = json.load(open(tpath))['traffic_bytes']
    if dominant in tb:
      roofline['traffic'] = tb[dominant]
      roofline['traffic_source'] = 'profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes)' % os.path.basename(tpath)
  if rank != 0:
    if distributed:
      torch.distributed.destroy_process_group()
    return
  result = {
      'metric': 'learner env-frames/s (T=%d)' % T, 'value': round(frames_per_s, 1), 'unit': 'env-frames/s',
      'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 4),
      'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
      'config': {'workload': '%s, T=%d B=%d/GPU A=%d, synthetic uint8 frames in HBM, num_action_repeats=1'
                             % (workload, T, B, A),
                 'global_batch': B * world, 'unroll_length': T, 'parallelism': 'dp%d' % world,
                 'grad_reduction': args.reduction, 'params': agent.flat.num_params(), 'launch': mode},
      'roofline': roofline,
      'loss': round(loss_val, 6),
      'kernels_ms_per_step': {k: round(v['total_ms'], 4) for k, v in kern.items()},
      # the other MFMA kernels of the attribution pass (>= 50 us per launch), same accounting as `roofline`
      'mfma_kernels': {
          k: dict(avg_ms=round(v['avg_ms'], 4), tflops=round(v['flops'] / (v['avg_ms'] * 1e-3) / 1e12, 1),
                  frac=round(v['flops'] / (v['avg_ms'] * 1e-3) / 1e12 /
                             (MFMA_BF16_PEAK_TF / 3.0 if (k in BF16X3_KERNELS and
                                                          os.environ.get('SEEDHIP_STACK_BF16', '1') != '0')
                              else MFMA_F32_PEAK_TF), 3))
          for k, v in kern.items() if v['flops'] > 0 and v['avg_ms'] >= 0.05},
  }
  if world == 1:
    err, err_ref, sweep = vtrace_checks(dev)
    result['vtrace_max_abs_err'] = err
    result['vtrace_max_abs_err_vs_reference_code'] = err_ref
    result['vtrace_scan_hbm'] = sweep
    if not args.no_cpu_baseline:
      from oracle import cpu_learner
      if r2:
        cb = args.cpu_batch if args.cpu_batch != 64 else 8
        fps, sec, thr = cpu_learner.time_cpu_r2d2_learner(A, T1, cb, steps=2, warmup=1)
      elif deep:
        cb = args.cpu_batch if args.cpu_batch != 64 else 16
        fps, sec, thr = cpu_learner.time_cpu_deep_learner(A, T1, cb, steps=2, warmup=1)
      else:
        cb = args.cpu_batch
        kind = 'atari_shallow' if args.torso == 'shallow' else 'atari_dqn_body'
        fps, sec, thr = cpu_learner.time_cpu_learner(kind, A, T1, cb, steps=3, warmup=1)
      result['cpu_baseline'] = {
          'value': round(fps, 1), 'unit': 'env-frames/s', 'cores': thr, 'kind': 'port',
          'sample': 'same learner step as eager PyTorch-CPU fp32 restatement of the reference graph '
                    '(oracle/cpu_learner.py), T=%d B=%d (per-frame cost is B-independent), median of timed '
                    'steps, %.2f s/step; host cpu_count=%d' % (T, cb, sec, os.cpu_count())}
      result['speedup_vs_cpu_baseline'] = round(frames_per_s / fps, 1)
  print(json.dumps(result))
  if distributed:
    torch.distributed.destroy_process_group()


if __name__ == '__main__':
  main()
