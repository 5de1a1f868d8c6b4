#!/usr/bin/env python
"""Per-kernel micro-benchmarks at the cfg2 / cfg3 layer shapes (HIP events on the launch stream).

  python tools/bench_kernels.py [stack|conv|fc|all] [--B 512]
Prints one line per kernel: avg ms, TFLOP/s (fp32 MFMA peak 157.3) or GB/s.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from seed_rl_amd import ops


def timeit(fn, reps=20, warm=3):
  for _ in range(warm):
    fn()
  s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  s.record()
  for _ in range(reps):
    fn()
  e.record()
  torch.cuda.synchronize()
  return s.elapsed_time(e) / reps


def report(name, ms, flops=0, nbytes=0):
  print('%-44s %8.4f ms  %7.1f TF/s (%.1f%% fp32-MFMA)  %7.1f GB/s' %
        (name, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3 * 100, nbytes / ms / 1e6))


def bench_stack(B, T1=21, cout=16):
  dev = torch.device('cuda')
  HW = 84 * 84
  ext = torch.randint(0, 256, (T1 + 3, B, HW), dtype=torch.uint8, device=dev)
  nv = torch.full((T1, B), 4, dtype=torch.uint8, device=dev)
  w = torch.randn(8, 8, 4, cout, device=dev) / 16
  b = torch.randn(cout, device=dev)
  g = ops.StackConvGeom(T1, B, 84, 84, 20, 20, 8, 8, 4, cout, cout)
  out = torch.empty((T1 * B, 20, 20, cout), device=dev)
  dy = torch.randn((T1 * B, 20, 20, cout), device=dev)
  dw, db = torch.empty_like(w), torch.empty_like(b)
  ws = torch.empty(ops.conv2d_stack_bwd_weight_workspace_bytes(g) // 4 + 4, device=dev)
  fl = 2.0 * T1 * B * 400 * cout * 256
  report('stack_conv_fwd cout=%d' % cout, timeit(lambda: ops.conv2d_stack_fwd(g, ext, nv, w, b, out)), fl,
         T1 * B * HW + out.numel() * 4)
  report('stack_conv_wgrad cout=%d' % cout, timeit(lambda: ops.conv2d_stack_bwd_weight(g, ext, nv, dy, dw, db, ws)), fl,
         T1 * B * HW + out.numel() * 4)


ONLY = None                                                # --only fwd|dgrad|wgrad


def bench_conv(name, n, ih, iw, cin, k, s, padding, cout):
  dev = torch.device('cuda')
  g = ops.conv_geom(n, ih, iw, cin, k, k, s, padding, cout)
  x = torch.randn((n, ih, iw, cin), device=dev)
  w = torch.randn((k, k, cin, cout), device=dev) / (k * k * cin) ** 0.5
  b = torch.randn(cout, device=dev)
  out = torch.empty((n, g.oh, g.ow, cout), device=dev)
  dy = torch.randn_like(out)
  dx = torch.empty_like(x)
  dw, db = torch.empty_like(w), torch.empty_like(b)
  ws = torch.empty(ops.conv2d_bwd_weight_workspace_bytes(g) // 4 + 4, device=dev)
  fl = 2.0 * n * g.oh * g.ow * cout * k * k * cin
  by = (x.numel() + out.numel() + w.numel()) * 4
  if ONLY in (None, 'fwd'):
    report(name + ' fwd', timeit(lambda: ops.conv2d_fwd(g, x, w, b, out, out_relu=True)), fl, by)
  if ONLY in (None, 'dgrad'):
    report(name + ' dgrad', timeit(lambda: ops.conv2d_bwd_data(g, dy, w, dx, relu_mask=x)), fl, by)
  if ONLY in (None, 'wgrad'):
    report(name + ' wgrad', timeit(lambda: ops.conv2d_bwd_weight(g, x, dy, dw, db, ws)), fl, by)


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('what', nargs='?', default='all')
  ap.add_argument('--B', type=int, default=512)
  ap.add_argument('--only', default=None, choices=['fwd', 'dgrad', 'wgrad'])
  a = ap.parse_args()
  global ONLY
  ONLY = a.only
  N = 21 * a.B
  if a.what in ('stack', 'all'):
    bench_stack(a.B)
  if a.what in ('conv', 'all'):
    bench_conv('conv 4x4/2 16->32 @20x20', N, 20, 20, 16, 4, 2, 'valid', 32)
  if a.what in ('fc', 'all'):
    bench_conv('fc 2592->256', N, 1, 1, 2592, 1, 1, 'valid', 256)
    bench_conv('heads 256->20', N, 1, 1, 256, 1, 1, 'valid', 20)
  if a.what in ('gemm',):
    for nm, n, k, c in [('atari fc 2592->256', 21 * 512, 2592, 256), ('atari heads 256->20', 21 * 512, 256, 20),
                        ('deep fc 3456->256', 21 * 256, 3456, 256), ('deep lstm-x 256->1024', 21 * 256, 256, 1024),
                        ('r2d2 fc 3136->512', 121 * 256, 3136, 512), ('r2d2 lstm-x 512->2048', 121 * 256, 512, 2048),
                        ('r2d2 recurrent 512->2048', 256, 512, 2048), ('deep recurrent 256->1024', 256, 256, 1024),
                        ('inference fc 2592->256', 64, 2592, 256)]:
      bench_conv(nm, n, 1, 1, k, 1, 1, 'valid', c)
  if a.what in ('r2d2',):                                   # the Dense-layer GEMMs of the cfg5 step (81 and 40 steps x B=256)
    for nm, n, k, c in [('r2d2 fc 3136->512 M=20736', 20736, 3136, 512), ('r2d2 fc 3136->512 M=10240', 10240, 3136, 512),
                        ('r2d2 lstm-x 532->2048 M=20736', 20736, 532, 2048), ('r2d2 lstm-x 532->2048 M=10240', 10240, 532, 2048),
                        ('r2d2 heads 512->512 M=20736', 20736, 512, 512)]:
      bench_conv(nm, n, 1, 1, k, 1, 1, 'valid', c)
  if a.what in ('r2d2conv',):                               # the DQN convs of the cfg5 step at its 81 x 256 training frames
    bench_conv('r2d2 conv2 4x4/2 32->64 @20x20', 20736, 20, 20, 32, 4, 2, 'valid', 64)
    bench_conv('r2d2 conv3 3x3/1 64->64 @9x9', 20736, 9, 9, 64, 3, 1, 'valid', 64)
  if a.what in ('entry',):                                  # ImpalaDeep's 16 -> 32 stack entry: plain forward / data gradient
    dev = torch.device('cuda')
    n = 21 * 256
    g = ops.conv_geom(n, 36, 48, 16, 3, 3, 1, 'same', 32)
    x = torch.randn((n, 36, 48, 16), device=dev); w = torch.randn((3, 3, 16, 32), device=dev) / 12
    b = torch.randn(32, device=dev); out = torch.empty((n, 36, 48, 32), device=dev); dy = torch.randn_like(out); dx = torch.empty_like(x)
    fl = 2.0 * n * 36 * 48 * 32 * 144; by = (x.numel() + out.numel()) * 4
    report('deep entry 16->32 fwd (pipe %d)' % ops.conv2d_pipe(g, 0), timeit(lambda: ops.conv2d_fwd(g, x, w, b, out)), fl, by)
    report('deep entry 16->32 dgrad (pipe %d)' % ops.conv2d_pipe(g, 1), timeit(lambda: ops.conv2d_bwd_data(g, dy, w, dx)), fl, by)
  if a.what in ('entrypool',):                              # ... its data gradient with the max-pool backward in the loader
    dev = torch.device('cuda')
    n = 21 * 256
    g = ops.conv_geom(n, 36, 48, 16, 3, 3, 1, 'same', 32)
    act = torch.randn((n, 36, 48, 32), device=dev); w = torch.randn((3, 3, 16, 32), device=dev) / 12
    y = torch.empty((n, 18, 24, 32), device=dev); arg = torch.empty((n, 18, 24, 32), dtype=torch.uint8, device=dev)
    ops.maxpool_fwd(act, y, arg)
    dp = torch.randn_like(y); d_a = torch.empty_like(act); dx = torch.empty((n, 36, 48, 16), device=dev)
    fl = 2.0 * n * 36 * 48 * 32 * 144; by = dp.numel() * 5 + d_a.numel() * 4 + dx.numel() * 4
    report('deep entry dgrad + pool backward', timeit(lambda: ops.conv2d_bwd_data_pool(g, dp, arg, w, dx, d_a)), fl, by)
    def two():
      ops.maxpool_bwd(dp, arg, d_a); ops.conv2d_bwd_data(g, d_a, w, dx)
    report('maxpool_bwd, then dgrad', timeit(two), fl, by)
  if a.what in ('deep',):
    n = 21 * 256
    bench_conv('deep s0 3x3 16->16 @36x48', n, 36, 48, 16, 3, 1, 'same', 16)
    bench_conv('deep s1 3x3 16->32 @36x48', n, 36, 48, 16, 3, 1, 'same', 32)
    bench_conv('deep s1 3x3 32->32 @18x24', n, 18, 24, 32, 3, 1, 'same', 32)
    bench_conv('deep s2 3x3 32->32 @9x12', n, 9, 12, 32, 3, 1, 'same', 32)


if __name__ == '__main__':
  main()
