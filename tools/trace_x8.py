#!/usr/bin/env python
"""Timeline of workgroup 0 of xg8_kernel (built with SEEDHIP_X8_EXP & 128): s_memtime stamps of wave 0 (group 0: brings
operand A) and wave 4 (group 1: brings operand B) around their MEM / COMP segments and the hand-over barriers.
  SEEDHIP_X8_EXP=128 python tools/trace_x8.py          (132: also without the split arithmetic, 160: without fragment reads)"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from seed_rl_amd import _lib, ops
from tools.bench_x6 import SHAPES


def main():
  n, cin, cout = SHAPES[sys.argv[1] if len(sys.argv) > 1 else 'atari']
  dev = torch.device('cuda')
  g = ops.dense_geom(n, cin, cout)
  x = torch.randn((n, cin), device=dev); w = torch.randn((cin, cout), device=dev) / cin ** 0.5
  b = torch.randn(cout, device=dev); out = torch.empty((n, cout), device=dev)
  for _ in range(3):
    ops.conv2d_fwd(g, x, w, b, out)
  buf = torch.zeros(2 * 64 * 8, dtype=torch.int64, device=dev)
  lib = _lib.lib()
  lib.seedhip_debug_x6_trace.argtypes = [ctypes.c_void_p]
  lib.seedhip_debug_x6_trace.restype = None
  lib.seedhip_debug_x6_trace(ctypes.c_void_p(buf.data_ptr()))
  ops.conv2d_fwd(g, x, w, b, out)
  torch.cuda.synchronize()
  lib.seedhip_debug_x6_trace(ctypes.c_void_p(0))
  t = buf.cpu().view(2, 64, 8).numpy().astype(np.int64)
  t0 = int(t[0, 0, 0])
  print('k-tile | group 0: MEM start, MEM end, released, COMP end, released | group 1: COMP start, COMP end, released, MEM end, released   (s_memtime ticks)')
  for j in range(1, 24):
    print('%3d | %s | %s' % (j, ' '.join('%7d' % (int(v) - t0) for v in t[0, j, :5]), ' '.join('%7d' % (int(v) - t0) for v in t[1, j, :5])))
  sl = slice(3, 22)
  d = lambda gidx, a, b: float(np.mean(t[gidx, sl, b] - t[gidx, sl, a]))
  print('group 0 (waves 0-3): MEM %.0f  wait at barrier %.0f  COMP %.0f  wait at barrier %.0f   period %.0f' % (
      d(0, 0, 1), d(0, 1, 2), d(0, 2, 3), d(0, 3, 4), float(np.mean(t[0, 4:23, 0] - t[0, 3:22, 0]))))
  print('group 1 (waves 4-7): COMP %.0f  wait at barrier %.0f  MEM %.0f  wait at barrier %.0f   period %.0f' % (
      d(1, 0, 1), d(1, 1, 2), d(1, 2, 3), d(1, 3, 4), float(np.mean(t[1, 4:23, 0] - t[1, 3:22, 0]))))


if __name__ == '__main__':
  main()
