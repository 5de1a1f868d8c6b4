"""The torch-CPU oracle's Keras layers (oracle/nets_torch.py) against a SECOND, independent restatement of the
documented TF 2.4.1 / Keras semantics written as direct NumPy loops.

Why: these layers are "parity unpinned" -- TensorFlow is not in this image and no reference test holds numbers for
Conv2D / MaxPool2D / LSTMCell / Adam.  Two restatements that were written separately and agree do not
replace a TF-produced vector, but they do catch the mistakes a single restatement can hide: NHWC / kernel layout,
'same' padding on even sizes (TF pads AFTER, PyTorch-style pooling pads symmetrically), gate order, where epsilon sits
in Adam.  The HIP path is then compared with nets_torch (tests/test_gpu_*.py); call sites: dmlab/networks.py:31-44,
77, 84-89; atari/networks.py:233-252; dmlab/vtrace_main.py:46-51."""
import math

import numpy as np
import torch

from oracle import nets_torch


def _conv2d_loops(x, k, b, stride, padding):
  """Keras Conv2D, NHWC input, kernel [kh, kw, cin, cout]; 'same' (stride 1): pad_total = k - 1, before = total // 2."""
  n, h, w, cin = x.shape
  kh, kw, _, cout = k.shape
  if padding == 'same':
    assert stride == 1
    pt, pl = (kh - 1) // 2, (kw - 1) // 2
    xp = np.zeros((n, h + kh - 1, w + kw - 1, cin), x.dtype)
    xp[:, pt:pt + h, pl:pl + w] = x
    oh, ow = h, w
  else:
    xp = x
    oh, ow = (h - kh) // stride + 1, (w - kw) // stride + 1
  y = np.zeros((n, oh, ow, cout), np.float64)
  for oy in range(oh):
    for ox in range(ow):
      patch = xp[:, oy * stride:oy * stride + kh, ox * stride:ox * stride + kw, :].astype(np.float64)   # [n,kh,kw,cin]
      y[:, oy, ox, :] = np.tensordot(patch, k.astype(np.float64), axes=([1, 2, 3], [0, 1, 2])) + b
  return y


def test_conv2d_same_and_valid():
  rng = np.random.default_rng(0)
  for (h, w, cin, cout, kk, s, pad) in ((6, 8, 3, 4, 3, 1, 'same'), (5, 7, 2, 3, 3, 1, 'same'), (20, 20, 4, 5, 8, 4, 'valid'),
                                       (9, 9, 3, 2, 4, 2, 'valid'), (7, 7, 2, 2, 3, 1, 'valid')):
    x = rng.normal(size=(2, h, w, cin)).astype(np.float32)
    k = rng.normal(size=(kk, kk, cin, cout)).astype(np.float32)
    b = rng.normal(size=cout).astype(np.float32)
    got = nets_torch.conv2d(torch.tensor(x), torch.tensor(k), torch.tensor(b), s, pad).numpy()
    want = _conv2d_loops(x, k, b, s, pad)
    assert got.shape == want.shape
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-5)


def _maxpool_loops(x):
  """MaxPool2D(3, strides=2, 'same'): out = ceil(in / 2); pad_total = (out - 1) * 2 + 3 - in; before = total // 2."""
  n, h, w, c = x.shape
  oh, ow = -(-h // 2), -(-w // 2)
  pth, ptw = max((oh - 1) * 2 + 3 - h, 0), max((ow - 1) * 2 + 3 - w, 0)
  bh, bw = pth // 2, ptw // 2
  y = np.full((n, oh, ow, c), -np.inf, np.float32)
  for oy in range(oh):
    for ox in range(ow):
      for dy in range(3):
        for dx in range(3):
          iy, ix = oy * 2 + dy - bh, ox * 2 + dx - bw
          if 0 <= iy < h and 0 <= ix < w:
            y[:, oy, ox] = np.maximum(y[:, oy, ox], x[:, iy, ix])
  return y


def test_maxpool_3x3_s2_same_even_and_odd():
  rng = np.random.default_rng(1)
  for h, w in ((72, 96), (36, 48), (18, 24), (6, 8), (7, 9), (5, 5), (1, 2)):
    x = rng.normal(size=(2, h, w, 3)).astype(np.float32)
    got = nets_torch.max_pool_3x3_s2_same(torch.tensor(x)).numpy()
    want = _maxpool_loops(x)
    assert got.shape == want.shape == (2, -(-h // 2), -(-w // 2), 3)
    np.testing.assert_array_equal(got, want)
  # even sizes: window i covers rows [2i, 2i + 2] (0 before, 1 after) -- NOT [2i - 1, 2i + 1]
  x = np.zeros((1, 4, 4, 1), np.float32)
  x[0, 0, 0, 0] = 5.0
  out = nets_torch.max_pool_3x3_s2_same(torch.tensor(x)).numpy()[0, :, :, 0]
  assert out[0, 0] == 5.0 and out[0, 1] == 0.0 and out[1, 0] == 0.0


def test_lstm_cell_gate_order_and_single_bias():
  rng = np.random.default_rng(2)
  B, I, H = 3, 5, 4
  x, h, c = (rng.normal(size=s).astype(np.float32) for s in ((B, I), (B, H), (B, H)))
  W, U, b = (rng.normal(size=s).astype(np.float32) for s in ((I, 4 * H), (H, 4 * H), (4 * H,)))
  h2, c2 = nets_torch.lstm_cell(*[torch.tensor(a) for a in (x, h, c, W, U, b)])
  sig = lambda z: 1.0 / (1.0 + np.exp(-z))
  z = x.astype(np.float64) @ W + h.astype(np.float64) @ U + b          # ONE bias vector; split order i, f, c~, o
  i, f, g, o = sig(z[:, :H]), sig(z[:, H:2 * H]), np.tanh(z[:, 2 * H:3 * H]), sig(z[:, 3 * H:])
  c_ref = f * c + i * g
  h_ref = o * np.tanh(c_ref)
  np.testing.assert_allclose(c2.numpy(), c_ref, atol=2e-6)
  np.testing.assert_allclose(h2.numpy(), h_ref, atol=2e-6)
  # the done-reset happens BEFORE the cell (dmlab/networks.py:162-166)
  p = {'core/kernel': torch.tensor(W), 'core/recurrent_kernel': torch.tensor(U), 'core/bias': torch.tensor(b)}
  xs = torch.tensor(np.stack([x, x]))
  done = torch.tensor(np.array([[False, True, False], [True, False, False]]))
  outs, _ = nets_torch.unroll_lstm(p, 'core', xs, done, (torch.tensor(h), torch.tensor(c)))
  h0 = np.where(done.numpy()[0][:, None], 0.0, h); c0 = np.where(done.numpy()[0][:, None], 0.0, c)
  z = x.astype(np.float64) @ W + h0 @ U + b
  c1 = sig(z[:, H:2 * H]) * c0 + sig(z[:, :H]) * np.tanh(z[:, 2 * H:3 * H])
  np.testing.assert_allclose(outs[0].numpy(), sig(z[:, 3 * H:]) * np.tanh(c1), atol=2e-6)


def test_keras_adam_and_schedule():
  """OptimizerV2 Adam: t = iterations + 1; lr_t = lr(t - 1) sqrt(1 - b2^t) / (1 - b1^t); m += (g - m)(1 - b1);
  v += (g^2 - v)(1 - b2); theta -= lr_t m / (sqrt(v) + eps) -- epsilon OUTSIDE the square root, not bias-corrected."""
  rng = np.random.default_rng(3)
  for b1, eps in ((0.9, 1e-7), (0.0, 3.125e-7)):
    p0 = rng.normal(size=7)
    grads = [rng.normal(size=7) for _ in range(3)]
    p = torch.tensor(p0.copy())
    opt = nets_torch.KerasAdam([p], nets_torch.polynomial_decay(4.8e-4, 10), beta_1=b1, beta_2=0.999, epsilon=eps)
    ref, m, v = p0.copy(), np.zeros(7), np.zeros(7)
    for it, g in enumerate(grads):
      opt.apply_gradients([torch.tensor(g)])
      t = it + 1
      lr = 4.8e-4 * (1 - min(it, 10) / 10)                              # PolynomialDecay(lr0, 10, end 0, power 1) at step t-1
      lr_t = lr * math.sqrt(1 - 0.999 ** t) / (1 - b1 ** t)
      m = m + (g - m) * (1 - b1)
      v = v + (g * g - v) * (1 - 0.999)
      ref = ref - lr_t * m / (np.sqrt(v) + eps)
      np.testing.assert_allclose(p.numpy(), ref, rtol=1e-12, atol=1e-15)
  f = nets_torch.polynomial_decay(1.0, 4)
  assert [f(s) for s in (0, 1, 4, 9)] == [1.0, 0.75, 0.0, 0.0]


def test_flatten_is_nhwc_and_dense_is_in_out():
  """Flatten on NHWC: index (h W + w) C + c; Dense kernel [in, out], y = x W + b (Appendix A) -- checked through the
  oracle's ImpalaDeep torso against the same layers applied by hand."""
  rng = np.random.default_rng(4)
  A, obs = 5, (8, 8, 3)
  spec = nets_torch.param_spec('impala_deep', A, obs)
  params = nets_torch.init_params(spec, seed=1)
  p = nets_torch.to_torch(params)
  frames = rng.integers(0, 256, (2, 1) + obs).astype(np.uint8)
  flat_ref = None
  x = frames.reshape((2,) + obs).astype(np.float32) / 255.0
  for i, ch in enumerate((16, 32, 32)):
    x = _conv2d_loops(x.astype(np.float32), params['stack%d/conv/kernel' % i], params['stack%d/conv/bias' % i], 1, 'same').astype(np.float32)
    x = _maxpool_loops(x)
    for blk in range(2):
      y = np.maximum(x, 0)
      y = _conv2d_loops(y, params['stack%d/res_%d/conv2d_0/kernel' % (i, blk)], params['stack%d/res_%d/conv2d_0/bias' % (i, blk)], 1, 'same').astype(np.float32)
      y = np.maximum(y, 0)
      y = _conv2d_loops(y, params['stack%d/res_%d/conv2d_1/kernel' % (i, blk)], params['stack%d/res_%d/conv2d_1/bias' % (i, blk)], 1, 'same').astype(np.float32)
      x = x + y
  x = np.maximum(x, 0)
  flat_ref = x.reshape(2, -1)                                           # NHWC flatten
  dense = np.maximum(flat_ref.astype(np.float64) @ params['conv_to_linear/kernel'] + params['conv_to_linear/bias'], 0)
  got = nets_torch.impala_deep_torso(p, torch.tensor(frames.reshape((2,) + obs)))
  np.testing.assert_allclose(got.numpy(), dense, rtol=0, atol=5e-5)
