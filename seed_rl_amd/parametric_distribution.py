"""Categorical action distribution on MI355X.

Mirrors the categorical branch of
/root/reference/common/parametric_distribution.py (ParametricDistribution
:31-80, categorical_distribution :83-97, get_parametric_distribution_for_
action_space :293-332): `log_prob(parameters, actions)` and
`entropy(parameters)` run as one HIP kernel (csrc/loss.hip,
`seedhip_categorical_log_prob_entropy`).  Continuous-control distributions of
the reference (normal/tanh, :100-290) are out of scope.
"""
import torch

from seed_rl_amd import _lib


class ParametricDistribution(object):
  """Categorical distribution over `param_size` actions (logits parametrisation)."""

  def __init__(self, param_size, dtype=torch.int64):
    self._param_size = param_size
    self._dtype = dtype
    self._rng = {}

  @property
  def param_size(self):
    return self._param_size

  @property
  def reparametrizable(self):
    return False

  def _rows(self, parameters):
    if parameters.shape[-1] != self._param_size:
      raise ValueError('expected last dim %d, got %s' % (self._param_size, tuple(parameters.shape)))
    return parameters.reshape(-1, self._param_size).to(torch.float32).contiguous()

  def _run(self, parameters, actions, want_lp, want_ent):
    _lib.require_cuda(parameters)
    with torch.no_grad():
      logits = self._rows(parameters)
      rows = logits.shape[0]
      lp = torch.empty(rows, device=logits.device, dtype=torch.float32) if want_lp else None
      ent = torch.empty(rows, device=logits.device, dtype=torch.float32) if want_ent else None
      act = None
      esz = 0
      if actions is not None:
        if tuple(actions.shape) != tuple(parameters.shape[:-1]):
          raise ValueError('actions shape %s != %s' % (tuple(actions.shape), tuple(parameters.shape[:-1])))
        act = actions.reshape(-1)
        if act.dtype not in (torch.int32, torch.int64):
          act = act.to(torch.int64)
        act = act.contiguous()
        esz = act.element_size()
      with torch.cuda.device(logits.device):
        rc = _lib.lib().seedhip_categorical_log_prob_entropy(
            _lib.ptr(logits), _lib.ptr(act), esz, rows, self._param_size, _lib.ptr(lp), _lib.ptr(ent),
            _lib.stream())
      _lib.check(rc, 'seedhip_categorical_log_prob_entropy')
    shp = parameters.shape[:-1]
    return (lp.reshape(shp) if want_lp else None, ent.reshape(shp) if want_ent else None)

  def log_prob(self, parameters, actions):
    """parametric_distribution.py:69-70."""
    return self._run(parameters, actions, True, False)[0]

  def entropy(self, parameters):
    """parametric_distribution.py:72-74."""
    return self._run(parameters, None, False, True)[1]

  def sample(self, parameters, seed=None):
    """parametric_distribution.py:66-67 (tfd.Categorical.sample; used by the agents' _head,
    dmlab/networks.py:120-122): Gumbel-max over counter-based randoms in one kernel (csrc/inference.hip:
    seedhip_categorical_sample); the generator state is a device (seed, counter) pair per device."""
    _lib.require_cuda(parameters)
    from seed_rl_amd import ops
    logits = self._rows(parameters)
    key = logits.device
    rng = self._rng.get(key)
    if rng is None or seed is not None:
      rng = torch.tensor([0x5EED if seed is None else int(seed), 0], dtype=torch.int64, device=logits.device)
      self._rng[key] = rng
    out = torch.empty(logits.shape[0], dtype=torch.int64, device=logits.device)
    ops.categorical_sample(logits, self._param_size, logits.shape[0], self._param_size, rng, out)
    return out.reshape(parameters.shape[:-1]).to(self._dtype)


def categorical_distribution(n_actions, dtype=torch.int64):
  """parametric_distribution.py:83-97."""
  return ParametricDistribution(n_actions, dtype)


class Discrete(object):
  """Minimal stand-in for gym.spaces.Discrete (gym is not part of the hot path)."""

  def __init__(self, n, dtype=torch.int64):
    self.n = n
    self.dtype = dtype


def get_parametric_distribution_for_action_space(action_space):
  """parametric_distribution.py:293-332 (Discrete branch :306-307)."""
  if isinstance(action_space, Discrete) or hasattr(action_space, 'n'):
    return categorical_distribution(action_space.n, getattr(action_space, 'dtype', torch.int64))
  raise NotImplementedError('only Discrete action spaces are on the MI355X hot path')
