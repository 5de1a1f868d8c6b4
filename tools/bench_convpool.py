#!/usr/bin/env python
"""Times the fused first stage of ImpalaDeep (conv 3x3 on uint8 frames + 3x3/2 max-pool), forward and backward, at the
cfg3 shape (T=20, B=256: 5120 frames of 72x96x3).  HIP events on the launch stream."""
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


def main():
  n, ih, iw = int(os.environ.get('N', 5120)), 72, 96
  dev = torch.device('cuda')
  x = torch.randint(0, 256, (n, ih, iw, 3), dtype=torch.uint8, device=dev)
  w = torch.randn(3, 3, 3, 16, device=dev) / 5
  b = torch.randn(16, device=dev)
  ph, pw = (ih + 1) // 2, (iw + 1) // 2
  pooled = torch.empty((n, ph, pw, 16), device=dev)
  arg = torch.empty((n, ph, pw, 16), dtype=torch.uint8, device=dev)
  dy = torch.randn((n, ph, pw, 16), device=dev)
  dw, db = torch.empty_like(w), torch.empty_like(b)
  ws = torch.empty(ops.conv3x3_u8_pool_bwd_workspace_bytes(n, ih, iw) // 4 + 1, device=dev)
  fl = 2.0 * n * ih * iw * 27 * 16
  byt_f = x.numel() + pooled.numel() * 4 + arg.numel()
  t = timeit(lambda: ops.conv3x3_u8_pool_fwd(x, w, b, pooled, arg))
  print('convpool fwd  %.4f ms  %.1f TF/s  %.0f GB/s' % (t, fl / t / 1e9, byt_f / t / 1e6))
  t = timeit(lambda: ops.conv3x3_u8_pool_bwd(x, dy, arg, dw, db, ws))
  print('convpool bwd  %.4f ms  %.1f TF/s  %.0f GB/s' % (t, fl / t / 1e9, byt_f / t / 1e6))


if __name__ == '__main__':
  main()
