#!/usr/bin/env python
"""A/B of the Dense kernels: fp32 MFMA (gemm.h) against the bf16x6 core (xgemm.h), one process per setting because
libseedhip.so reads its knobs once.   python tools/bench_x6.py [shapes...]
Each child prints avg ms per kernel (HIP events) and, with --check, the max error against an fp64 evaluation."""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = {'atari': (21 * 512, 2592, 256), 'r2d2': (20736, 3136, 512), 'deep': (21 * 256, 3456, 256),
          'lstmx': (20736, 532, 2048), 'deeplstm': (21 * 256, 256, 1024)}


def child(names, check):
  sys.path.insert(0, ROOT)
  import numpy as np
  import torch
  from seed_rl_amd import ops
  from tools.bench_kernels import timeit
  dev = torch.device('cuda')
  for nm in names:
    n, cin, cout = SHAPES[nm]
    g = ops.dense_geom(n, cin, cout)
    x = torch.randn((n, cin), device=dev); w = torch.randn((cin, cout), device=dev) / cin ** 0.5
    b = torch.randn(cout, device=dev); dy = torch.randn((n, cout), device=dev)
    out = torch.empty((n, cout), device=dev); dx = torch.empty_like(x); dw = torch.empty_like(w); db = torch.empty_like(b)
    ws = torch.empty(ops.conv2d_bwd_weight_workspace_bytes(g) // 4 + 4, device=dev)
    fl = 2.0 * n * cin * cout
    t = [timeit(lambda: ops.conv2d_fwd(g, x, w, b, out, out_relu=True)),
         timeit(lambda: ops.conv2d_bwd_data(g, dy, w, dx, relu_mask=x)),
         timeit(lambda: ops.conv2d_bwd_weight(g, x, dy, dw, db, ws))]
    line = '%-9s M=%d K=%d N=%d  fwd %.4f ms (%.0f TF)  dgrad %.4f (%.0f)  wgrad %.4f (%.0f)' % (
        nm, n, cin, cout, t[0], fl / t[0] / 1e9, t[1], fl / t[1] / 1e9, t[2], fl / t[2] / 1e9)
    if check:
      ops.conv2d_fwd(g, x, w, b, out)
      ops.conv2d_bwd_data(g, dy, w, dx)
      ops.conv2d_bwd_weight(g, x, dy, dw, db, ws)
      x64, w64, dy64 = x.double(), w.double(), dy.double()
      refs = [(out, x64 @ w64 + b.double(), x @ w + b), (dx, dy64 @ w64.T, dy @ w.T), (dw, x64.T @ dy64, x.T @ dy)]
      for (got, r64, r32), k in zip(refs, ('fwd', 'dgrad', 'wgrad')):
        e = (got.double() - r64).abs().max().item(); e32 = (r32.double() - r64).abs().max().item()
        rms = ((got.double() - r64) ** 2).mean().sqrt().item(); rms32 = ((r32.double() - r64) ** 2).mean().sqrt().item()
        bias = (got.double() - r64).mean().item()
        line += '\n    %-5s max err %.3e (torch fp32 %.3e)  rms %.3e (%.3e)  mean %.2e  scale %.2e' % (
            k, e, e32, rms, rms32, bias, r64.abs().max().item())
    print(line, flush=True)


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('shapes', nargs='*', default=['atari'])
  ap.add_argument('--child', action='store_true')
  ap.add_argument('--check', action='store_true')
  ap.add_argument('--settings', default='X6=0;X6=7;X6=7,X6_EXP=1')
  a = ap.parse_args()
  if a.child:
    return child(a.shapes, a.check)
  for setting in a.settings.split(';'):
    env = dict(os.environ)
    for kv in setting.split(','):
      k, v = kv.split('=')
      env['SEEDHIP_' + k] = v
    print('== ' + setting, flush=True)
    subprocess.run([sys.executable, os.path.abspath(__file__), '--child'] + (['--check'] if a.check else []) + a.shapes,
                   env=env, check=False)


if __name__ == '__main__':
  main()
