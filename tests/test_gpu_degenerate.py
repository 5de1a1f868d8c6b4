"""Degenerate sizes through the host mirror: nothing may crash or hang, and whatever the reference would accept (empty
time axis, a single column, a single image) must give the oracle's answer.  (The reference's own tests do the same for
its ops: tests/vtrace_test.py shapes, grpc ops_test.py:74-117 empty inputs.)"""
import numpy as np
import pytest
import torch

from oracle import nets_torch

pytestmark = pytest.mark.gpu


def test_vtrace_empty_time_and_single_column(device):
  from seed_rl_amd import vtrace
  z = torch.zeros((0, 5), device=device)
  out = vtrace.from_importance_weights(z, z, z, z, z, torch.ones(5, device=device))
  assert tuple(out.vs.shape) == (0, 5) and tuple(out.pg_advantages.shape) == (0, 5)
  one = torch.full((1, 1), 0.5, device=device)
  out = vtrace.from_importance_weights(one * 0, one * 0, one, one, one, torch.ones(1, device=device))
  # rho = 1: delta = r + d * boot - v = 0.5 + 0.5 - 0.5; vs = v + delta
  assert abs(float(out.vs[0, 0]) - 1.0) < 1e-6 and abs(float(out.pg_advantages[0, 0]) - 0.5) < 1e-6


def test_single_image_convs(device):
  """n = 1 on the smallest maps the kernels accept: one workgroup, partial tiles everywhere."""
  from seed_rl_amd import ops
  rng = np.random.default_rng(0)
  for ih, iw, cin, cout in [(1, 1, 4, 4), (2, 2, 4, 8), (3, 3, 16, 16), (4, 5, 32, 32)]:
    x = rng.normal(size=(1, ih, iw, cin)).astype(np.float32)
    w = rng.normal(size=(3, 3, cin, cout)).astype(np.float32) / 10
    b = rng.normal(size=(cout,)).astype(np.float32)
    ref = nets_torch.conv2d(torch.tensor(x), torch.tensor(w), torch.tensor(b), 1, 'same').numpy()
    g = ops.conv_geom(1, ih, iw, cin, 3, 3, 1, 'same', cout)
    out = torch.empty((1, ih, iw, cout), device=device)
    ops.conv2d_fwd(g, torch.tensor(x).to(device), torch.tensor(w).to(device), torch.tensor(b).to(device), out)
    assert np.max(np.abs(out.cpu().numpy() - ref)) <= 2e-5 * max(1.0, np.abs(ref).max()), (ih, iw, cin, cout)


def test_pool_single_pixel(device):
  from seed_rl_amd import ops
  x = torch.arange(8, dtype=torch.float32, device=device).reshape(1, 1, 1, 8)
  y = torch.empty((1, 1, 1, 8), device=device)
  arg = torch.empty((1, 1, 1, 8), dtype=torch.uint8, device=device)
  ops.maxpool_fwd(x, y, arg)
  assert torch.equal(y, x)
  dx = torch.empty_like(x)
  ops.maxpool_bwd(torch.ones_like(y), arg, dx)
  assert torch.equal(dx, torch.ones_like(x))


def test_bad_arguments_fail_loudly(device):
  """Shape errors come back as Python exceptions carrying seedhip_last_error(), never as a device fault."""
  from seed_rl_amd import _lib, ops
  x = torch.zeros((1, 4, 4, 6), device=device)                       # 6 channels: not a multiple of 4
  with pytest.raises(Exception):
    ops.maxpool_fwd(x, torch.empty((1, 2, 2, 6), device=device), torch.empty((1, 2, 2, 6), dtype=torch.uint8, device=device))
  with pytest.raises(Exception):
    ops.conv3x3_u8_pool_fwd(torch.zeros((1, 2, 2, 3), dtype=torch.uint8, device=device), torch.zeros((3, 3, 3, 16), device=device),
                            torch.zeros(16, device=device), torch.empty((1, 1, 1, 16), device=device),
                            torch.empty((1, 1, 1, 16), dtype=torch.uint8, device=device))
  torch.cuda.synchronize()
