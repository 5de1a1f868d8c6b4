#!/usr/bin/env python
"""Does a CU-masked stream (ops.cu_mask_stream) really confine kernels?  Times a CU-bound torch kernel and one HIP-graph
replay of it on streams that keep k of every m compute units: time should scale like m / k."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from seed_rl_amd import ops


def main():
  dev = torch.device('cuda')
  a = torch.randn(4096, 4096, device=dev)
  b = torch.randn(4096, 4096, device=dev)
  torch.mm(a, b); torch.cuda.synchronize()
  g = torch.cuda.CUDAGraph()
  with torch.cuda.graph(g):
    c = torch.mm(a, b)
  for name, keep in (('all', lambda i: True), ('i % 16 < 8', lambda i: i % 16 < 8), ('i % 16 < 4', lambda i: i % 16 < 4),
                     ('i % 2 < 1', lambda i: i % 2 < 1), ('i < 128', lambda i: i < 128), ('i < 64', lambda i: i < 64),
                     ('i < 104', lambda i: i < 104), ('i >= 104', lambda i: i >= 104)):
    s, kept = ops.cu_mask_stream(dev, keep)
    for label, fn in (('eager', lambda: torch.mm(a, b)), ('graph', g.replay)):
      with torch.cuda.stream(s):
        for _ in range(3):
          fn()
        s.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
          fn()
        s.synchronize()
      print('%-12s (%3d bits) %s: %.1f us per 4096^3 matmul' % (name, kept, label, (time.perf_counter() - t0) / 20 * 1e6))


if __name__ == '__main__':
  main()
