"""CPU tests of the implicit-GEMM problem accessors (the index math the GPU kernel
runs) against the torch-CPU oracle.  No GPU, no HIP: tests/host/emul.cpp is
compiled with g++ and includes seed_rl_amd/csrc/conv_problems.h directly."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import frames_np, nets_torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Geom(ctypes.Structure):
  _fields_ = [(n, ctypes.c_int) for n in
              'n_img ih iw cin oh ow kh kw stride pad_t pad_l cout ld_in ld_out'.split()]


class SGeom(ctypes.Structure):
  _fields_ = [(n, ctypes.c_int) for n in 'T B ih iw oh ow kh kw stride cout ld_out'.split()]


@pytest.fixture(scope='module')
def emul():
  out = os.path.join(ROOT, 'build', 'libseedhip_emul.so')
  os.makedirs(os.path.dirname(out), exist_ok=True)
  src = os.path.join(ROOT, 'tests', 'host', 'emul.cpp')
  subprocess.check_call(['g++', '-O2', '-std=c++17', '-shared', '-fPIC', src, '-o', out])
  return ctypes.CDLL(out)


def ptr(a):
  return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def make_geom(n, ih, iw, cin, kh, kw, stride, padding, cout, ld_in=None, ld_out=None):
  if padding == 'same':
    oh, ow = -(-ih // stride), -(-iw // stride)
    pt = max((oh - 1) * stride + kh - ih, 0) // 2
    pl = max((ow - 1) * stride + kw - iw, 0) // 2
  else:
    oh, ow, pt, pl = (ih - kh) // stride + 1, (iw - kw) // stride + 1, 0, 0
  return Geom(n, ih, iw, cin, oh, ow, kh, kw, stride, pt, pl, cout, ld_in or cin, ld_out or cout)


def test_fastdiv(emul):
  emul.emul_fastdiv_check.restype = ctypes.c_int
  for d in [1, 2, 3, 4, 5, 7, 9, 12, 16, 18, 20, 27, 64, 81, 84, 100, 255, 400, 7056, 20736, 65537]:
    assert emul.emul_fastdiv_check(ctypes.c_uint32(d), ctypes.c_uint32(3000000), ctypes.c_uint32(1))
    assert emul.emul_fastdiv_check(ctypes.c_uint32(d), ctypes.c_uint32(2 ** 31 - 1), ctypes.c_uint32(104729))


CONV_CASES = [
    # n, ih, iw, cin, kh, kw, stride, padding, cout
    (2, 7, 6, 16, 3, 3, 1, 'same', 16),      # ImpalaDeep res conv
    (2, 6, 8, 3, 3, 3, 1, 'same', 16),       # ImpalaDeep first conv (cin=3, scalar path)
    (2, 9, 10, 16, 3, 3, 1, 'same', 32),
    (2, 12, 12, 16, 4, 4, 2, 'valid', 32),   # shallow conv2
    (1, 20, 20, 4, 8, 8, 4, 'valid', 32),    # DQN conv1 on fp32 stacked input
    (2, 9, 9, 32, 3, 3, 1, 'valid', 64),     # DQN conv3 -> BN=64 tile
    (5, 1, 1, 40, 1, 1, 1, 'valid', 20),     # Dense 40 -> 20
    (3, 11, 9, 8, 5, 3, 2, 'valid', 12),     # odd shapes, (ih-kh)%s != 0
]


@pytest.mark.parametrize('case', CONV_CASES)
@pytest.mark.parametrize('variant', ['plain', 'relu_res', 'u8'])
def test_conv_fwd_dgrad_wgrad(emul, case, variant):
  n, ih, iw, cin, kh, kw, stride, padding, cout = case
  rng = np.random.default_rng(hash(case) % 1000)
  in_relu = out_relu = 0
  in_dtype = 0
  if variant == 'u8':
    x_raw = rng.integers(0, 256, (n, ih, iw, cin)).astype(np.uint8)
    x = torch.tensor(x_raw).float() / 255
    in_dtype = 1
  else:
    x_raw = rng.normal(size=(n, ih, iw, cin)).astype(np.float32)
    x = torch.tensor(x_raw)
  w = rng.normal(size=(kh, kw, cin, cout)).astype(np.float32) * 0.2
  b = rng.normal(size=(cout,)).astype(np.float32)
  g = make_geom(n, ih, iw, cin, kh, kw, stride, padding, cout)
  res = None
  if variant == 'relu_res':
    in_relu = 1
    if padding == 'same' and cin == cout:
      res = rng.normal(size=(n, g.oh, g.ow, cout)).astype(np.float32)
    else:
      out_relu = 1
  x.requires_grad_(True)
  wt = torch.tensor(w, requires_grad=True)
  bt = torch.tensor(b, requires_grad=True)
  xin = F.relu(x) if in_relu else x
  y = nets_torch.conv2d(xin, wt, bt, stride, padding)
  if res is not None:
    y = y + torch.tensor(res)
  if out_relu:
    y = F.relu(y)
  dy = rng.normal(size=y.shape).astype(np.float32)
  y.backward(torch.tensor(dy))

  out = np.zeros((n, g.oh, g.ow, cout), np.float32)
  emul.emul_conv_fwd(ctypes.byref(g), ptr(x_raw), in_dtype, in_relu, ptr(w), ptr(b), ptr(out), out_relu, ptr(res))
  np.testing.assert_allclose(out, y.detach().numpy(), rtol=1e-4, atol=1e-5)

  # gradient wrt pre-activation output
  dz = dy * (y.detach().numpy() > 0) if out_relu else dy
  dz = np.ascontiguousarray(dz, np.float32)
  dw = np.zeros_like(w); db = np.zeros_like(b)
  emul.emul_conv_wgrad(ctypes.byref(g), ptr(x_raw), in_dtype, in_relu, ptr(dz), ptr(dw), ptr(db), 64)
  np.testing.assert_allclose(dw, wt.grad.numpy(), rtol=1e-4, atol=1e-4)
  np.testing.assert_allclose(db, bt.grad.numpy(), rtol=1e-4, atol=1e-4)

  if variant != 'u8':
    dx = np.full((n, ih, iw, cin), 7.0, np.float32)    # every in-range element must be overwritten
    mask = x_raw if in_relu else None
    add = rng.normal(size=dx.shape).astype(np.float32) if variant == 'relu_res' else None
    emul.emul_conv_dgrad(ctypes.byref(g), ptr(dz), ptr(w), ptr(dx), ptr(mask), ptr(add))
    ref = x.grad.numpy() + (add if add is not None else 0)
    np.testing.assert_allclose(dx, ref, rtol=1e-4, atol=1e-4)


GATHER_CASES = [
    (2, 9, 10, 16, 3, 3, 1, 'same', 32),     # ImpalaDeep residual conv: taps predicated at the border
    (2, 12, 12, 16, 4, 4, 2, 'valid', 32),   # second Atari conv: four parity classes in one data-gradient GEMM
    (2, 9, 9, 32, 3, 3, 1, 'valid', 64),     # DQN conv3: 'valid' stride 1, dY smaller than the input map
    (1, 13, 11, 8, 4, 4, 2, 'valid', 16),    # odd map: super-pixels beyond the dY border and beyond the input
    (2, 7, 8, 4, 3, 3, 1, 'same', 8),        # 4 channels per tap (8 taps per 32-deep k-tile), K = 36
    (1, 12, 12, 16, 6, 6, 3, 'valid', 16),   # stride 3: nine parity classes
]


@pytest.mark.parametrize('case', GATHER_CASES)
def test_gather_gemm_index_math(emul, case):
  """The gathered-operand arithmetic of the GEMM core (seed_rl_amd/csrc/gemm_geom.h: gather_row / gather_tap /
  gather_inside / scatter_addr and the conv_*_setup geometry) executed on the CPU, element by element, against the
  torch oracle: forward, super-pixel data gradient (with ReLU mask and accumulate), transposed-im2col weight gradient."""
  n, ih, iw, cin, kh, kw, stride, padding, cout = case
  rng = np.random.default_rng(abs(hash(case)) % 1000)
  x_raw = rng.normal(size=(n, ih, iw, cin)).astype(np.float32)
  w = rng.normal(size=(kh, kw, cin, cout)).astype(np.float32) * 0.2
  b = rng.normal(size=(cout,)).astype(np.float32)
  g = make_geom(n, ih, iw, cin, kh, kw, stride, padding, cout)
  x = torch.tensor(x_raw, requires_grad=True)
  wt = torch.tensor(w, requires_grad=True); bt = torch.tensor(b, requires_grad=True)
  y = F.relu(nets_torch.conv2d(F.relu(x), wt, bt, stride, padding))
  dy = rng.normal(size=y.shape).astype(np.float32)
  y.backward(torch.tensor(dy))
  for f in (emul.emul_gather_fwd, emul.emul_gather_dgrad, emul.emul_gather_wgrad):
    f.restype = ctypes.c_int

  out = np.zeros((n, g.oh, g.ow, cout), np.float32)
  assert emul.emul_gather_fwd(ctypes.byref(g), ptr(x_raw), 1, ptr(w), ptr(b), ptr(out), 1)
  np.testing.assert_allclose(out, y.detach().numpy(), rtol=1e-4, atol=1e-5)

  dz = np.ascontiguousarray(dy * (y.detach().numpy() > 0), np.float32)
  dw = np.zeros_like(w); db = np.zeros_like(b)
  assert emul.emul_gather_wgrad(ctypes.byref(g), ptr(x_raw), 1, ptr(dz), ptr(dw), ptr(db))
  np.testing.assert_allclose(dw, wt.grad.numpy(), rtol=1e-4, atol=1e-4)
  np.testing.assert_allclose(db, bt.grad.numpy(), rtol=1e-4, atol=1e-4)

  dx = np.full((n, ih, iw, cin), 7.0, np.float32)        # every element must be overwritten
  add = rng.normal(size=dx.shape).astype(np.float32)
  ok = emul.emul_gather_dgrad(ctypes.byref(g), ptr(dz), ptr(w), ptr(dx), ptr(x_raw), ptr(add))
  if kh % stride == 0 and kw % stride == 0 and (stride == 1 or padding == 'valid'):
    assert ok
    np.testing.assert_allclose(dx, x.grad.numpy() + add, rtol=1e-4, atol=1e-4)


def test_conv_strided_ld(emul):
  """Dense with padded row strides (LSTM input concat buffer / head output)."""
  rng = np.random.default_rng(0)
  n, cin, cout, ld_in, ld_out = 6, 8, 20, 12, 24
  xbuf = rng.normal(size=(n, ld_in)).astype(np.float32)
  w = rng.normal(size=(1, 1, cin, cout)).astype(np.float32)
  b = rng.normal(size=(cout,)).astype(np.float32)
  g = make_geom(n, 1, 1, cin, 1, 1, 1, 'valid', cout, ld_in, ld_out)
  out = np.zeros((n, ld_out), np.float32)
  emul.emul_conv_fwd(ctypes.byref(g), ptr(xbuf), 0, 0, ptr(w), ptr(b), ptr(out), 0, None)
  np.testing.assert_allclose(out[:, :cout], xbuf[:, :cin] @ w[0, 0] + b, rtol=1e-5, atol=1e-5)
  assert np.all(out[:, cout:] == 0)
  dy = np.zeros((n, ld_out), np.float32); dy[:, :cout] = rng.normal(size=(n, cout))
  dx = np.zeros((n, ld_in), np.float32)
  emul.emul_conv_dgrad(ctypes.byref(g), ptr(dy), ptr(w), ptr(dx), None, None)
  np.testing.assert_allclose(dx[:, :cin], dy[:, :cout] @ w[0, 0].T, rtol=1e-5, atol=1e-5)
  dw = np.zeros_like(w); db = np.zeros_like(b)
  emul.emul_conv_wgrad(ctypes.byref(g), ptr(xbuf), 0, 0, ptr(dy), ptr(dw), ptr(db), 4)
  np.testing.assert_allclose(dw[0, 0], xbuf[:, :cin].T @ dy[:, :cout], rtol=1e-5, atol=1e-5)
  np.testing.assert_allclose(db, dy[:, :cout].sum(0), rtol=1e-5, atol=1e-5)


def _stack_inputs(rng, T, B, H, W):
  frames = rng.integers(0, 256, (T, B, H, W, 1)).astype(np.uint8)
  done = rng.uniform(size=(T, B)) < 0.25
  state = rng.integers(0, 2 ** 24, (B, H * W)).astype(np.int32)
  return frames, done, state


def host_stack_prepare(frames, done, state):
  """NumPy mirror of seedhip_stack_prepare (frames.hip) for the CPU tests."""
  T, B = done.shape
  HW = state.shape[1]
  ext = np.zeros((T + 3, B, HW), np.uint8)
  ext[0] = state & 0xFF; ext[1] = (state >> 8) & 0xFF; ext[2] = (state >> 16) & 0xFF
  ext[3:] = frames.reshape(T, B, HW)
  nv = np.full((T, B), 4, np.uint8)
  for t in range(T):
    for b in range(B):
      if done[t, b]: nv[t, b] = 1
      elif t >= 1 and done[t - 1, b]: nv[t, b] = 2
      elif t >= 2 and done[t - 2, b]: nv[t, b] = 3
  return ext, nv


@pytest.mark.parametrize('kw_,stride', [(8, 4), (4, 4), (3, 2)])
def test_stack_conv_matches_oracle(emul, kw_, stride):
  """Fused stack_frames + /255 + conv1 == oracle stack_frames -> conv."""
  rng = np.random.default_rng(kw_)
  T, B, H, W, cout = 5, 3, 12 + kw_, 16 + kw_, 16
  H -= (H - kw_) % stride; W -= (W - kw_) % stride
  if kw_ % 4 == 0: W -= W % 4
  frames, done, state = _stack_inputs(rng, T, B, H, W)
  stacked, _ = frames_np.stack_frames(frames, state, done, 4)
  x = torch.tensor(stacked / np.float32(255)).reshape(T * B, H, W, 4)
  w = (rng.normal(size=(kw_, kw_, 4, cout)) * 0.1).astype(np.float32)
  b = rng.normal(size=(cout,)).astype(np.float32)
  wt = torch.tensor(w, requires_grad=True); bt = torch.tensor(b, requires_grad=True)
  y = F.relu(nets_torch.conv2d(x, wt, bt, stride, 'valid'))
  dy = rng.normal(size=y.shape).astype(np.float32)
  y.backward(torch.tensor(dy))
  ext, nv = host_stack_prepare(frames, done, state)
  oh, ow = (H - kw_) // stride + 1, (W - kw_) // stride + 1
  g = SGeom(T, B, H, W, oh, ow, kw_, kw_, stride, cout, cout)
  out = np.zeros((T * B, oh, ow, cout), np.float32)
  emul.emul_stack_fwd(ctypes.byref(g), ptr(ext), ptr(nv), ptr(w), ptr(b), ptr(out), 1)
  np.testing.assert_allclose(out, y.detach().numpy(), rtol=1e-4, atol=1e-5)
  dz = np.ascontiguousarray(dy * (out > 0), np.float32)
  dw = np.zeros_like(w); db = np.zeros_like(b)
  emul.emul_stack_wgrad(ctypes.byref(g), ptr(ext), ptr(nv), ptr(dz), ptr(dw), ptr(db), 128)
  np.testing.assert_allclose(dw, wt.grad.numpy(), rtol=1e-4, atol=1e-4)
  np.testing.assert_allclose(db, bt.grad.numpy(), rtol=1e-4, atol=1e-4)


WS_CASES = [
    (3, 20, 20, 16, 4, 4, 2, 'valid', 32),   # the second Atari conv (cfg2): the shape the specialised kernel runs
    (2, 20, 22, 16, 4, 4, 2, 'valid', 32),   # non-square map
    (2, 19, 21, 16, 4, 4, 2, 'valid', 32),   # odd extents: super-pixels hang over the input map and over dY
    (2, 20, 20, 8, 4, 4, 2, 'valid', 64),    # 64 output channels (NR = 4 forward), 32-column data gradient
]


@pytest.mark.parametrize('case', WS_CASES)
def test_ws_kernel_index_math(emul, case):
  """The index arithmetic of the weight-stationary conv kernels (seed_rl_amd/csrc/wsgemm_geom.h: row decode by
  division + stepping, A byte offsets, tap validity, W' re-indexing, dX scatter offsets -- the helpers
  ws_fast_kernel calls) executed on the CPU in the kernel's own tile order, against the torch oracle."""
  n, ih, iw, cin, kh, kw, stride, padding, cout = case
  rng = np.random.default_rng(abs(hash(case)) % 1000)
  x_raw = np.abs(rng.normal(size=(n, ih, iw, cin))).astype(np.float32)
  x_raw[rng.uniform(size=x_raw.shape) < 0.3] = 0.0                         # post-ReLU input: also the dgrad's mask
  w = rng.normal(size=(kh, kw, cin, cout)).astype(np.float32) * 0.2
  b = rng.normal(size=(cout,)).astype(np.float32)
  g = make_geom(n, ih, iw, cin, kh, kw, stride, padding, cout)
  x = torch.tensor(x_raw, requires_grad=True)
  wt = torch.tensor(w); bt = torch.tensor(b)
  y = F.relu(nets_torch.conv2d(x, wt, bt, stride, padding))
  dy = rng.normal(size=y.shape).astype(np.float32)
  y.backward(torch.tensor(dy))
  emul.emul_ws_fwd.restype = ctypes.c_int; emul.emul_ws_dgrad.restype = ctypes.c_int
  out = np.zeros((n, g.oh, g.ow, cout), np.float32)
  ok = emul.emul_ws_fwd(ctypes.byref(g), ptr(x_raw), ptr(w), ptr(b), ptr(out), 1)
  assert ok                                                                 # all WS_CASES are inside the kernel's range
  np.testing.assert_allclose(out, y.detach().numpy(), rtol=1e-4, atol=1e-5)
  dz = np.ascontiguousarray(dy * (y.detach().numpy() > 0), np.float32)
  dx = np.full((n, ih, iw, cin), 7.0, np.float32)                           # every element must be overwritten
  add = rng.normal(size=dx.shape).astype(np.float32)
  ok = emul.emul_ws_dgrad(ctypes.byref(g), ptr(dz), ptr(w), ptr(dx), ptr(x_raw), ptr(add))
  assert ok
  np.testing.assert_allclose(dx, x.grad.numpy() * (x_raw > 0) + add, rtol=1e-4, atol=1e-4)


def test_batch_gate_bounds_without_device_reads():
  """learner_server.BatchGate (r6): the learner's lower bound and admission's upper bound of the device's column count
  from per-batch mirror words and the numbered submissions, against a simulated device that completes batches late and
  in order -- the lower bound never exceeds the true count at the dequeue's place in the stream, admission never lets
  the true count pass the capacity, and both bounds are exact when nothing is in flight."""
  import collections
  import threading
  import numpy as np
  import torch
  from seed_rl_amd import learner_server

  class State(object):
    cap = 40
    batch_count = torch.zeros(1, dtype=torch.int32)

  rng = np.random.default_rng(0)
  st, lock = State(), threading.Lock()
  n, B = 8, 16
  gate = learner_server.BatchGate(st, n, lock, ring=64, mirrors=torch.zeros(64, dtype=torch.int32))
  # the "stream": submissions execute in order when the simulated device gets to them
  stream = collections.deque()
  true_count = 0
  done_tokens = collections.deque()
  peak = 0
  for step in range(4000):
    r = rng.uniform()
    if r < 0.45 and not gate.would_block():
      with lock:
        st.batch_count[0] = -12345                      # the host must never look at the live count
        tok = gate.submitted()
        stream.append(('batch', tok, int(rng.integers(0, n + 1))))
    elif r < 0.6 and gate.fill >= B:
      with lock:
        gate.dequeued(B)
        stream.append(('deq', None, B))
    elif r < 0.9 and stream:
      kind, tok, amount = stream.popleft()              # the device executes the oldest submission
      if kind == 'batch':
        true_count += amount
        gate.mirrors[tok % 64] = true_count             # the batch's own mirror word, written behind it
        done_tokens.append(tok)
      else:
        assert true_count >= amount                     # the lower bound held at the dequeue's place in the stream
        true_count -= amount
      peak = max(peak, true_count)
      assert true_count <= st.cap
    elif done_tokens:
      gate.completed(done_tokens.popleft())             # the completion thread notices, possibly much later
    assert gate.up >= true_count + sum(a for k, _, a in stream if k == 'batch') - sum(a for k, _, a in stream if k == 'deq')
  while stream:
    kind, tok, amount = stream.popleft()
    if kind == 'batch':
      true_count += amount; gate.mirrors[tok % 64] = true_count; done_tokens.append(tok)
    else:
      assert true_count >= amount; true_count -= amount
  while done_tokens:
    gate.completed(done_tokens.popleft())
  assert gate.low == gate.up == true_count and gate.inflight == 0
  assert peak > B                                        # the run really filled the batch
