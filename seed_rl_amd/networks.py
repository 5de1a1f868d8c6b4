"""Agents of the SEED-RL learner on MI355X: explicit forward / backward over the
HIP kernels (no autograd tape, no tracing compiler).

Mirrors the reference's agent duck-type (SURVEY.md 8(b) B1;
/root/reference/dmlab/networks.py:91-150): `initial_state(batch_size)`,
`__call__(prev_actions, env_outputs, core_state, unroll=False, is_training=False)
-> (AgentOutput, new_state)`, `get_action(...)`.  In addition each agent has
`backward()` (the TF tape's job in the reference, agents/vtrace/learner.py:261-266)
which consumes the head gradient written by the fused loss kernel and fills the
flat gradient buffer.

AtariShallow: the reference has NO Atari V-trace network.
It is built from reference parts: learner-side frame stacking
(atari/networks.py:57-173,330) -> conv torso -> policy/baseline heads
(dmlab/networks.py:116-124); torso 'shallow' = IMPALA-paper shallow net
(Conv 8x8/4x16, Conv 4x4/2x32, Dense 256), torso 'dqn' = the reference's Atari conv
body (atari/networks.py:233-242).
"""
import collections
import math

import numpy as np
import torch

from seed_rl_amd import _lib, ops
from seed_rl_amd.flat import FlatParams

AgentOutput = collections.namedtuple('AgentOutput', 'action policy_logits baseline')
AgentState = collections.namedtuple('AgentState', 'core_state frame_stacking_state')


def _round4(n):
  return (n + 3) // 4 * 4


def keras_init(spec, seed=0):
  """Keras default initialisers (glorot_uniform kernels, zero biases, orthogonal
  recurrent kernel, unit_forget_bias) from numpy.random.default_rng(seed).
  spec entries: (name, shape, init)."""
  rng = np.random.default_rng(seed)
  out = collections.OrderedDict()
  for name, shape, init in spec:
    if init == 'glorot':
      rf = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
      lim = math.sqrt(6.0 / (shape[-2] * rf + shape[-1] * rf))
      v = rng.uniform(-lim, lim, size=shape)
    elif init == 'orthogonal':
      a = rng.standard_normal(size=(max(shape), min(shape)))
      q, r = np.linalg.qr(a)
      q = q * np.sign(np.diag(r))
      v = (q if shape[0] >= shape[1] else q.T).reshape(shape)
    elif init == 'lstm_bias':
      h = shape[0] // 4
      v = np.zeros(shape)
      v[h:2 * h] = 1.0
    else:
      v = np.zeros(shape)
    out[name] = v.astype(np.float32)
  return out


class _Agent(object):
  """Shared machinery: flat parameters, per-shape workspaces, packed policy/baseline head."""

  def __init__(self, num_actions, device):
    self._num_actions = num_actions
    self.device = torch.device(device)
    if self.device.type != 'cuda':
      raise _lib.SeedHipError('agents run on a HIP device only (no CPU fallback)')
    self._ws = {}
    self._ldh = _round4(num_actions + 1)     # head row: [logits(A) | baseline | zero pad]

  # -- parameters ------------------------------------------------------------- #
  def _build_params(self, ref_spec, seed):
    """ref_spec: reference-structured [(name, shape, init)].  The policy and baseline
    Dense layers are stored packed as ONE [feat, ldh] matrix so that both heads are one
    GEMM whose output row feeds the fused loss kernel directly."""
    self._ref_spec = ref_spec
    internal = []
    for name, shape, _ in ref_spec:
      if name.startswith('policy_logits/') or name.startswith('baseline/'):
        continue
      internal.append((name, shape))
    feat = dict((n, s) for n, s, _ in ref_spec)['policy_logits/kernel'][0]
    internal.append(('heads/kernel', (feat, self._ldh)))
    internal.append(('heads/bias', (self._ldh,)))
    self.flat = FlatParams(internal, self.device)
    self.load_reference_params(keras_init(ref_spec, seed))

  def load_reference_params(self, values):
    """values: {reference variable name: numpy array} (Keras layouts)."""
    A = self._num_actions
    with torch.no_grad():
      for name, shape, _ in self._ref_spec:
        v = torch.as_tensor(np.asarray(values[name], np.float32)).to(self.device)
        if name == 'policy_logits/kernel':
          self.flat.p('heads/kernel')[:, :A] = v
        elif name == 'policy_logits/bias':
          self.flat.p('heads/bias')[:A] = v
        elif name == 'baseline/kernel':
          self.flat.p('heads/kernel')[:, A] = v[:, 0]
        elif name == 'baseline/bias':
          self.flat.p('heads/bias')[A] = v[0]
        else:
          self.flat.p(name).copy_(v.reshape(self.flat.p(name).shape))

  def _ref_view(self, getter, name):
    A = self._num_actions
    if name == 'policy_logits/kernel':
      return getter('heads/kernel')[:, :A]
    if name == 'policy_logits/bias':
      return getter('heads/bias')[:A]
    if name == 'baseline/kernel':
      return getter('heads/kernel')[:, A:A + 1]
    if name == 'baseline/bias':
      return getter('heads/bias')[A:A + 1]
    return getter(name)

  @property
  def trainable_variables(self):
    """Reference-structured list [(name, view)] (e.g. 39 tensors for ImpalaDeep,
    tests/agents_test.py:45)."""
    return [(n, self._ref_view(self.flat.p, n)) for n, _, _ in self._ref_spec]

  def reference_gradients(self):
    return collections.OrderedDict((n, self._ref_view(self.flat.g, n)) for n, _, _ in self._ref_spec)

  # -- workspaces ---------------------------------------------------------------- #
  def _buf(self, key, shape, dtype=torch.float32, zero=False):
    k = (key, tuple(shape), dtype)
    t = self._ws.get(k)
    if t is None:
      t = (torch.zeros if zero else torch.empty)(tuple(shape), dtype=dtype, device=self.device)
      self._ws[k] = t
    return t

  def entropy_cost(self):
    raise NotImplementedError

  def get_action(self, *args, **kwargs):
    return self.__call__(*args, **kwargs)

  # -- head -------------------------------------------------------------------- #
  def _head_fwd(self, feat, rows, feat_dim, ld_feat=None):
    ldh = self._ldh
    head = self._buf('head', (rows, ldh))
    g = ops.dense_geom(rows, feat_dim, ldh, ld_in=ld_feat or feat_dim)
    ops.conv2d_fwd(g, feat, self.flat.p('heads/kernel'), self.flat.p('heads/bias'), head)
    return head

  def _sample(self, logits):
    gmb = -torch.log(-torch.log(torch.rand_like(logits).clamp_min(1e-20)).clamp_min(1e-20))
    return torch.argmax(logits + gmb, dim=-1)

  def _agent_output(self, head, T1, B, sample):
    A = self._num_actions
    h3 = head.view(T1, B, self._ldh)
    logits, baseline = h3[..., :A], h3[..., A]
    action = self._sample(logits) if sample else None
    return AgentOutput(action, logits, baseline)


class AtariShallow(_Agent):
  """Frame-stacked Atari policy/value agent (see module docstring, D1)."""

  def __init__(self, num_actions, observation_shape=(84, 84, 1), torso='shallow', device='cuda', seed=0,
               entropy_cost=0.00025):
    super(AtariShallow, self).__init__(num_actions, device)
    h, w, c = observation_shape
    if c != 1:
      raise ValueError('frame stacking needs a single-channel observation (atari/networks.py:86-88)')
    self._obs = (h, w)
    self._convs = [(8, 4, 16), (4, 2, 32)] if torso == 'shallow' else [(8, 4, 32), (4, 2, 64), (3, 1, 64)]
    self._fc = 256 if torso == 'shallow' else 512
    self._entropy_cost = entropy_cost
    spec, cin = [], 4
    self._shapes = []                       # (ih, iw, cin, k, s, cout, oh, ow)
    for i, (k, s, ch) in enumerate(self._convs):
      spec += [('conv%d/kernel' % i, (k, k, cin, ch), 'glorot'), ('conv%d/bias' % i, (ch,), 'zeros')]
      oh, ow = (h - k) // s + 1, (w - k) // s + 1
      self._shapes.append((h, w, cin, k, s, ch, oh, ow))
      h, w, cin = oh, ow, ch
    self._flat_dim = h * w * cin
    spec += [('fc/kernel', (self._flat_dim, self._fc), 'glorot'), ('fc/bias', (self._fc,), 'zeros'),
             ('policy_logits/kernel', (self._fc, num_actions), 'glorot'), ('policy_logits/bias', (num_actions,), 'zeros'),
             ('baseline/kernel', (self._fc, 1), 'glorot'), ('baseline/bias', (1,), 'zeros')]
    self._build_params(spec, seed)
    self._last = None

  def entropy_cost(self):
    return self._entropy_cost

  def initial_state(self, batch_size):
    hw = self._obs[0] * self._obs[1]
    return AgentState(core_state=(), frame_stacking_state=torch.zeros((batch_size, hw), dtype=torch.int32,
                                                                      device=self.device))

  def frames_buffer(self, T1, B):
    """uint8 [3+T1, B, H*W] trajectory buffer; rows 3.. are the unroll's frames.  A data
    pipeline can write observations straight into `frames_buffer(T1,B)[3:]` (time-major)
    and pass that view as env_outputs.observation: no copy is made then."""
    return self._buf('frames_ext', (T1 + 3, B, self._obs[0] * self._obs[1]), torch.uint8, zero=True)

  def __call__(self, prev_actions, env_outputs, agent_state, unroll=False, is_training=False):
    del prev_actions   # the feed-forward agent does not consume it
    obs, done = env_outputs.observation, env_outputs.done
    if not unroll:
      obs, done = obs[None], done[None]
    T1, B = done.shape[0], done.shape[1]
    H, W = self._obs
    HW = H * W
    N = T1 * B
    ext = self.frames_buffer(T1, B)
    fr = obs.reshape(T1, B, HW)
    if fr.dtype != torch.uint8:
      raise ValueError('observations must be uint8 frames')
    if fr.data_ptr() != ext[3:].data_ptr():
      ext[3:].copy_(fr)
    nvalid = self._buf('nvalid', (T1, B), torch.uint8)
    done_u8 = done.to(torch.uint8).contiguous()
    ops.stack_prepare(agent_state.frame_stacking_state.contiguous(), done_u8, T1, B, HW, ext, nvalid)

    acts = []
    ih, iw, cin, k, s, ch, oh, ow = self._shapes[0]
    g0 = ops.StackConvGeom(T1, B, ih, iw, oh, ow, k, k, s, ch, ch)
    a = self._buf('act0', (N, oh, ow, ch))
    ops.conv2d_stack_fwd(g0, ext, nvalid, self.flat.p('conv0/kernel'), self.flat.p('conv0/bias'), a, out_relu=True)
    acts.append(a)
    geoms = [g0]
    for i in range(1, len(self._shapes)):
      ih, iw, cin, k, s, ch, oh, ow = self._shapes[i]
      g = ops.conv_geom(N, ih, iw, cin, k, k, s, 'valid', ch)
      a2 = self._buf('act%d' % i, (N, oh, ow, ch))
      ops.conv2d_fwd(g, a, self.flat.p('conv%d/kernel' % i), self.flat.p('conv%d/bias' % i), a2, out_relu=True)
      acts.append(a2); geoms.append(g); a = a2
    gfc = ops.dense_geom(N, self._flat_dim, self._fc)
    hfc = self._buf('fc_out', (N, self._fc))
    ops.conv2d_fwd(gfc, a, self.flat.p('fc/kernel'), self.flat.p('fc/bias'), hfc, out_relu=True)
    head = self._head_fwd(hfc, N, self._fc)

    new_fs = torch.empty_like(agent_state.frame_stacking_state)
    ops.stack_pack_state(ext, nvalid, T1, B, HW, new_fs)
    self._last = dict(T1=T1, B=B, N=N, ext=ext, nvalid=nvalid, acts=acts, geoms=geoms, gfc=gfc, hfc=hfc, head=head)
    out = self._agent_output(head, T1, B, sample=not is_training)
    if not unroll:
      out = AgentOutput(*[None if t is None else t[0] for t in out])
    return out, AgentState(core_state=(), frame_stacking_state=new_fs)

  # head-gradient buffer the fused loss kernel writes into (same layout as `head`)
  def head_buffers(self):
    L = self._last
    d_head = self._buf('d_head', (L['N'], self._ldh), zero=True)
    return L['head'], d_head, self._ldh

  def backward(self):
    """Gradients of the loss wrt all parameters, given d_head (written by the loss kernel)."""
    L = self._last
    N = L['N']
    fl = self.flat
    d_head = self._buf('d_head', (N, self._ldh), zero=True)
    wsb = self._wgrad_ws()
    # heads
    gh = ops.dense_geom(N, self._fc, self._ldh)
    ops.conv2d_bwd_weight(gh, L['hfc'], d_head, fl.g('heads/kernel'), fl.g('heads/bias'), wsb)
    dz = self._buf('d_fc', (N, self._fc))
    ops.conv2d_bwd_data(gh, d_head, fl.p('heads/kernel'), dz, relu_mask=L['hfc'])
    # fc
    a_last = L['acts'][-1]
    ops.conv2d_bwd_weight(L['gfc'], a_last, dz, fl.g('fc/kernel'), fl.g('fc/bias'), wsb)
    da = self._buf('d_act%d' % (len(L['acts']) - 1), tuple(a_last.shape))
    ops.conv2d_bwd_data(L['gfc'], dz, fl.p('fc/kernel'), da, relu_mask=a_last)
    # convs, last to second
    for i in range(len(L['acts']) - 1, 0, -1):
      g = L['geoms'][i]
      a_in = L['acts'][i - 1]
      ops.conv2d_bwd_weight(g, a_in, da, fl.g('conv%d/kernel' % i), fl.g('conv%d/bias' % i), wsb)
      d_in = self._buf('d_act%d' % (i - 1), tuple(a_in.shape))
      ops.conv2d_bwd_data(g, da, fl.p('conv%d/kernel' % i), d_in, relu_mask=a_in)
      da = d_in
    # first conv: weight gradient straight from the uint8 frames
    ops.conv2d_stack_bwd_weight(L['geoms'][0], L['ext'], L['nvalid'], da, fl.g('conv0/kernel'), fl.g('conv0/bias'),
                                self._stack_ws(L['geoms'][0]))

  def _wgrad_ws(self):
    L = self._last
    need = ops.conv2d_bwd_weight_workspace_bytes(ops.dense_geom(L['N'], self._fc, self._ldh))
    need = max(need, ops.conv2d_bwd_weight_workspace_bytes(L['gfc']))
    for g in L['geoms'][1:]:
      need = max(need, ops.conv2d_bwd_weight_workspace_bytes(g))
    return self._buf('wgrad_ws', (need // 4 + 4,))

  def _stack_ws(self, g0):
    return self._buf('stack_ws', (ops.conv2d_stack_bwd_weight_workspace_bytes(g0) // 4 + 4,))
