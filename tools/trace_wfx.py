#!/usr/bin/env python
"""Per-round cycle stamps of wfx.h's forward kernel (second Atari conv on the bf16 matrix pipe).

  SEEDHIP_WFX_TRACE=1 python tools/trace_wfx.py [n_images]

The library launches the TRACE build of the kernel, waits for it and prints, for two workgroups and one wave of each
tap half, how many s_memtime ticks (1.92 GHz on MI355X under this load; calibrated against s_memrealtime in the
output's first line) each part of a round took: address arithmetic + finishing the previous round's outputs, the eight
taps, the two barriers.  DESIGN.md section 7 (round 4) quotes these numbers.  Timing only: run without the variable for
results."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from seed_rl_amd import ops


def main():
  n = int(sys.argv[1]) if len(sys.argv) > 1 else 8448
  g = ops.conv_geom(n, 20, 20, 16, 4, 4, 2, 'valid', 32)
  x = torch.randn((n, 20, 20, 16), device='cuda')
  w = torch.randn((4, 4, 16, 32), device='cuda') / 16
  b = torch.randn(32, device='cuda')
  out = torch.empty((n, 9, 9, 32), device='cuda')
  ops.conv2d_fwd(g, x, w, b, out, out_relu=True)
  torch.cuda.synchronize()


if __name__ == '__main__':
  main()
