"""The exact three-way bf16 split behind the first Atari conv's forward (seed_rl_amd/csrc/stackconv.hip, "bf16x3"):
every fp32 weight must equal hi + mid + lo bit for bit, with each part a bf16 number (numpy restatement of the
device code; CPU only)."""
import numpy as np


def bf16_trunc_bits(x):
  return (x.astype(np.float32).view(np.uint32) >> 16).astype(np.uint32)


def as_float(bits16):
  return (bits16.astype(np.uint32) << 16).view(np.float32)


def split3(w):
  hi = bf16_trunc_bits(w)
  r1 = (w - as_float(hi)).astype(np.float32)
  mid = bf16_trunc_bits(r1)
  r2 = (r1 - as_float(mid)).astype(np.float32)
  lo = bf16_trunc_bits(r2)                                     # r2 has <= 8 significant bits: nothing is cut
  return hi, mid, lo, r2


def test_split_is_exact():
  rng = np.random.default_rng(0)
  cases = [
      (rng.normal(size=200000) / 16).astype(np.float32),                      # conv kernels at init scale
      ((rng.normal(size=200000) / 16).astype(np.float32) / np.float32(255.0)),  # with the folded 1/255
      rng.uniform(-4, 4, 200000).astype(np.float32),
      (rng.normal(size=200000) * 1e-6).astype(np.float32),
      np.array([0.0, -0.0, 1.0, -1.0, 255.0, 1 / 255.0, 3.0e38, 1.2e-30, np.float32(1) + np.float32(2) ** -23],
               np.float32),
  ]
  for w in cases:
    hi, mid, lo, r2 = split3(w)
    assert np.array_equal(as_float(lo), r2)                                    # third part takes the rest exactly
    total = (as_float(hi).astype(np.float64) + as_float(mid).astype(np.float64) + as_float(lo).astype(np.float64))
    assert np.array_equal(total.astype(np.float32), w) and np.array_equal(total, w.astype(np.float64))


def test_pixels_are_exact_in_bf16():
  n = np.arange(256, dtype=np.float32)
  assert np.array_equal(as_float(bf16_trunc_bits(n)), n)
  assert np.all((n.view(np.uint32) & 0xFFFF) == 0)                             # the device code just takes the high half
