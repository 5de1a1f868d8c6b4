"""NumPy restatement of the categorical branch of
/root/reference/common/parametric_distribution.py:69-74,83-97 (oracle only).

The arithmetic lives in tensorflow_probability==0.11.0 `tfd.Categorical`
(un-vendored third party, docker/Dockerfile.dmlab:76):
  log_prob(k) = logits[k] - logsumexp(logits)       (max-subtracted)
  entropy     = -sum softmax * log_softmax
The reference pins log_prob with tests/vtrace_test.py:88-115.
"""
import numpy as np


def log_softmax(logits, dtype=np.float32):
  x = np.asarray(logits, dtype=dtype)
  m = np.max(x, axis=-1, keepdims=True)
  z = (x - m).astype(dtype)
  lse = np.log(np.sum(np.exp(z).astype(dtype), axis=-1, keepdims=True,
                      dtype=dtype)).astype(dtype)
  return (z - lse).astype(dtype)


def log_prob(logits, actions, dtype=np.float32):
  ls = log_softmax(logits, dtype)
  a = np.asarray(actions).astype(np.int64)
  return np.take_along_axis(ls, a[..., None], axis=-1)[..., 0]


def entropy(logits, dtype=np.float32):
  ls = log_softmax(logits, dtype)
  p = np.exp(ls).astype(dtype)
  return (-np.sum(p * ls, axis=-1, dtype=dtype)).astype(dtype)
