"""A minimal NumPy stand-in for the handful of TensorFlow ops used by the reference's pure-math functions
(common/vtrace.py:from_importance_weights; agents/r2d2/learner.py value rescaling, n_step_bellman_target,
compute_loss_and_priorities_from_agent_outputs).  It exists ONLY so that tests/golden/make_golden.py can execute
the REFERENCE'S OWN CODE in this container (TensorFlow is not installable here) and record golden vectors.
Every op computes in float32 like the TF CPU kernels (transcendentals may differ from Eigen's by an ulp)."""
import contextlib

import numpy as np

float32, int32, int64, bool_ = np.float32, np.int32, np.int64, np.bool_


class Shape(tuple):
  @property
  def ndims(self):
    return len(self)

  @property
  def dims(self):
    return list(self)

  def assert_has_rank(self, rank):
    if len(self) != rank:
      raise ValueError('Shape %s must have rank %d' % (tuple(self), rank))


def _a(x):
  return x.a if isinstance(x, Tensor) else x


class Tensor(object):

  def __init__(self, a):
    self.a = np.asarray(a)

  @property
  def shape(self):
    return Shape(self.a.shape)

  @property
  def dtype(self):
    return self.a.dtype

  def __getitem__(self, idx):
    return Tensor(self.a[idx])

  def _bin(self, other, fn, swap=False):
    o = _a(other)
    if not isinstance(o, np.ndarray) or o.dtype != self.a.dtype:
      o = np.asarray(o, self.a.dtype)
    return Tensor(fn(o, self.a) if swap else fn(self.a, o))

  def __add__(self, o): return self._bin(o, np.add)
  def __radd__(self, o): return self._bin(o, np.add, True)
  def __sub__(self, o): return self._bin(o, np.subtract)
  def __rsub__(self, o): return self._bin(o, np.subtract, True)
  def __mul__(self, o): return self._bin(o, np.multiply)
  def __rmul__(self, o): return self._bin(o, np.multiply, True)
  def __truediv__(self, o): return self._bin(o, np.divide)
  def __rtruediv__(self, o): return self._bin(o, np.divide, True)
  def __neg__(self): return Tensor(-self.a)
  def numpy(self): return self.a


def convert_to_tensor(x, dtype=None, name=None):
  return Tensor(np.asarray(_a(x), dtype))


def _stack_if_list(x):
  if isinstance(x, (list, tuple)):
    return np.stack([_a(v) for v in x])
  return _a(x)


def exp(x): return Tensor(np.exp(_a(x)))
def minimum(a, b, name=None):
  a, b = _a(a), _a(b)
  dt = a.dtype if isinstance(a, np.ndarray) else b.dtype
  return Tensor(np.minimum(np.asarray(a, dt), np.asarray(b, dt)))
def concat(xs, axis=0): return Tensor(np.concatenate([_a(x) for x in xs], axis=axis))
def expand_dims(x, axis): return Tensor(np.expand_dims(_a(x), axis))
def zeros_like(x): return Tensor(np.zeros_like(_a(x)))
def add(a, b, name=None): return Tensor(_stack_if_list(a) + _stack_if_list(b))
def stop_gradient(x): return x
def cast(x, dtype): return Tensor(_a(x).astype(dtype))
def abs(x): return Tensor(np.abs(_a(x)))                     # pylint: disable=redefined-builtin
def shape(x): return Shape(_a(x).shape)
def one_hot(idx, depth, on=1., off=0.):
  i = _a(idx)
  return Tensor(np.where(np.arange(int(depth))[None, ...] == i[..., None], np.float32(on), np.float32(off)))
def reduce_sum(x, axis=None): return Tensor(np.sum(_a(x), axis=axis, dtype=np.float32))
def reduce_max(x, axis=None): return Tensor(np.max(_a(x), axis=axis))
def reduce_mean(x, axis=None): return Tensor(np.mean(_a(x), axis=axis, dtype=np.float32))


@contextlib.contextmanager
def name_scope(name):
  yield


class _Math(object):
  sign = staticmethod(lambda x: Tensor(np.sign(_a(x))))
  sqrt = staticmethod(lambda x: Tensor(np.sqrt(_a(x))))
  abs = staticmethod(lambda x: Tensor(np.abs(_a(x))))
  square = staticmethod(lambda x: Tensor(np.square(_a(x))))


math = _Math()
