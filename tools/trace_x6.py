#!/usr/bin/env python
"""Timeline of one workgroup of the wave-specialised Dense kernel (xgemm_ws_kernel built with SEEDHIP_X6_WEXP & 128):
s_memtime stamps of the MFMA wave 0 and of one stager wave per operand, per k-tile step.
  SEEDHIP_X6_WEXP=128 python tools/trace_x6.py [atari|r2d2]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from seed_rl_amd import _lib, ops
from tools.bench_x6 import SHAPES


def main():
  nm = sys.argv[1] if len(sys.argv) > 1 else 'atari'
  n, cin, cout = SHAPES[nm]
  dev = torch.device('cuda')
  g = ops.dense_geom(n, cin, cout)
  x = torch.randn((n, cin), device=dev); w = torch.randn((cin, cout), device=dev) / cin ** 0.5
  b = torch.randn(cout, device=dev); out = torch.empty((n, cout), device=dev)
  for _ in range(3):
    ops.conv2d_fwd(g, x, w, b, out)
  buf = torch.zeros(4 * 64 * 8, dtype=torch.int64, device=dev)
  lib = _lib.lib()
  lib.seedhip_debug_x6_trace.argtypes = [ctypes.c_void_p]
  lib.seedhip_debug_x6_trace.restype = None
  lib.seedhip_debug_x6_trace(ctypes.c_void_p(buf.data_ptr()))
  ops.conv2d_fwd(g, x, w, b, out)
  torch.cuda.synchronize()
  lib.seedhip_debug_x6_trace(ctypes.c_void_p(0))
  t = buf.cpu().view(4, 64, 8).numpy()
  t0 = int(t[0, 0, 0])
  print('step | MFMA wave: at-barrier  released  mid(after 24 MFMAs)  end | stager A: top waited split+written loads-issued released | stager B: same')
  for s in range(40):
    m = [int(v) - t0 for v in t[0, s, :4]]
    a = [int(v) - t0 for v in t[2, s, :5]]
    bb = [int(v) - t0 for v in t[3, s, :5]]
    print('%3d | %7d %7d %7d %7d | %7d %7d %7d %7d %7d | %7d %7d %7d %7d %7d' % tuple([s] + m + a + bb))
  # per-phase averages over steps 5..35 (s_memtime ticks = 100 MHz? print raw; compare with 1536-cycle MFMA phases)
  import numpy as np
  sl = slice(5, 36)
  print('MFMA wave: barrier wait %.0f  first half %.0f  second half %.0f  step %.0f' % (
      np.mean(t[0, sl, 1] - t[0, sl, 0]), np.mean(t[0, sl, 2] - t[0, sl, 1]), np.mean(t[0, sl, 3] - t[0, sl, 2]),
      np.mean(t[0, 6:37, 0] - t[0, 5:36, 0])))
  for role, name in ((2, 'stager A'), (3, 'stager B')):
    print('%s: load wait %.0f  split+LDS writes %.0f  load issue %.0f  barrier wait %.0f  step %.0f' % (
        name, np.mean(t[role, sl, 1] - t[role, sl, 0]), np.mean(t[role, sl, 2] - t[role, sl, 1]),
        np.mean(t[role, sl, 3] - t[role, sl, 2]), np.mean(t[role, sl, 4] - t[role, sl, 3]),
        np.mean(t[role, 6:37, 0] - t[role, 5:36, 0])))


if __name__ == '__main__':
  main()
