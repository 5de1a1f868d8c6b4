"""Flat fp32 parameter / gradient / Adam-state buffers.

All trainable tensors of an agent live in ONE contiguous HBM buffer (each tensor
at a 16-byte aligned offset), so the optimizer is one kernel launch
(csrc/adam.hip) and the data-parallel exchange is ONE RCCL all-reduce of one
bucket instead of the reference's per-variable updates
(agents/vtrace/learner.py:265-275).
"""
import collections

import numpy as np
import torch


class FlatParams(object):

  def __init__(self, spec, device):
    """spec: ordered [(name, shape)]."""
    self.spec = [(n, tuple(int(d) for d in s)) for n, s in spec]
    self.offsets = collections.OrderedDict()
    off = 0
    for n, s in self.spec:
      self.offsets[n] = off
      off += (int(np.prod(s)) + 3) // 4 * 4
    self.size = off
    self.device = device
    self.constraint = None      # (flat index, lo, hi): one element with a Keras variable constraint (optimizers.Adam)
    self.params = torch.zeros(off, dtype=torch.float32, device=device)
    self.grads = torch.zeros(off, dtype=torch.float32, device=device)
    self._views = {}
    self._gviews = {}
    for n, s in self.spec:
      o, k = self.offsets[n], int(np.prod(s))
      self._views[n] = self.params[o:o + k].view(s)
      self._gviews[n] = self.grads[o:o + k].view(s)

  def p(self, name):
    return self._views[name]

  def g(self, name):
    return self._gviews[name]

  def names(self):
    return [n for n, _ in self.spec]

  def num_params(self):
    return sum(int(np.prod(s)) for _, s in self.spec)
