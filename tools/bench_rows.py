#!/usr/bin/env python
"""First conv of one central-inference step (seedhip_conv2d_stack_fwd_rows) alone: time per launch for a few access
patterns of the history rows, to separate the kernel's own time from the cost of reaching the store.

  python tools/bench_rows.py [--n 1024] [--envs 4096] [--L 21]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from seed_rl_amd import ops


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--n', type=int, default=1024)
  ap.add_argument('--envs', type=int, default=4096)
  ap.add_argument('--L', type=int, default=21)
  ap.add_argument('--cout', type=int, default=16)
  ap.add_argument('--iters', type=int, default=200)
  a = ap.parse_args()
  dev = torch.device('cuda')
  n, E, L, HW, C = a.n, a.envs, a.L, 84 * 84, a.cout
  g = torch.Generator(device='cpu').manual_seed(0)
  store = torch.randint(0, 256, (L * E, HW), dtype=torch.uint8, generator=g).to(dev)
  obs = torch.randint(0, 256, (n, HW), dtype=torch.uint8, generator=g).to(dev)
  w = torch.randn(8, 8, 4, C, generator=g).to(dev)
  bias = torch.randn(C, generator=g).to(dev)
  split = torch.zeros(ops.serve_conv0_split_bytes(C) // 4, dtype=torch.int32, device=dev)
  ops.serve_split_conv0(w, C, split)
  out = torch.empty(n, 20, 20, C, device=dev)
  geom = ops.StackConvGeom(1, n, 84, 84, 20, 20, 8, 8, 4, C, C)
  envs = torch.randperm(E, generator=g)[:n]

  def rows(idx, pattern):
    h = torch.zeros(n, 4, dtype=torch.int64)
    for c in range(4):
      sl = (idx - c) % L
      h[:, c] = sl * E + (envs if pattern == 'scattered' else torch.arange(n))
    return h.reshape(-1).to(dev)

  def run(name, hist, append, nvalid):
    for _ in range(5):
      ops.conv2d_stack_fwd_rows(geom, obs, store, hist, nvalid, split, bias, out)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(a.iters):
      ops.conv2d_stack_fwd_rows(geom, obs, store, hist, nvalid, split, bias, out)
    e.record()
    torch.cuda.synchronize()
    print('%-56s %7.1f us' % (name, s.elapsed_time(e) / a.iters * 1e3))

  no_append = torch.full((n,), -1, dtype=torch.int64, device=dev)
  nv4 = torch.full((n,), 4, dtype=torch.uint8, device=dev)
  nv1 = torch.full((n,), 1, dtype=torch.uint8, device=dev)
  for pattern in ('scattered', 'contiguous'):
    h = rows(7, pattern)
    app = h.view(n, 4)[:, 0].contiguous()
    run('%s history, 4 frames' % pattern, h, no_append, nv4)
    run('%s history, 1 frame (request only)' % pattern, h, no_append, nv1)


if __name__ == '__main__':
  main()
