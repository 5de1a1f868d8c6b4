"""PyTorch-CPU fp32 restatement of the reference agents' graphs (oracle; test
infrastructure only — also the `cpu_baseline` "port" timed by bench.py).

Graphs restated (op order as in the reference, eager, unfused):
  * ImpalaDeep          /root/reference/dmlab/networks.py:26-171
  * DuelingLSTMDQNNet   /root/reference/atari/networks.py:221-340
  * AtariShallow        NO reference definition exists:
      stack_frames (atari/networks.py:57-173) -> /255 -> IMPALA-paper shallow
      torso (Conv 8x8/4x16, Conv 4x4/2x32, Dense 256; arXiv 1802.01561 fig. 3)
      or the reference's Atari conv body (atari/networks.py:233-242) ->
      policy/baseline heads of dmlab/networks.py:116-124.
Layer arithmetic is Keras / TF 2.4.1 (un-vendored third party) restated from
SURVEY.md Appendix A: NHWC Conv2D with [kh,kw,cin,cout] kernels, TF 'SAME'
max-pool padding (0 before / 1 after for even sizes), NHWC Flatten, Dense
[in,out], LSTMCell gate order i,f,c~,o with a single bias, Keras Adam.
PARITY UNPINNED for these layers (no reference test pins their numbers).
"""
import collections
import math

import numpy as np
import torch
import torch.nn.functional as F

AgentOutput = collections.namedtuple('AgentOutput', 'action policy_logits baseline')
R2D2Output = collections.namedtuple('R2D2Output', 'action q_values')


# --------------------------------------------------------------------------- #
# Parameter specs + Keras-style initialisation.
# --------------------------------------------------------------------------- #
def _conv_out(n, k, s, padding):
  return (n + s - 1) // s if padding == 'same' else (n - k) // s + 1


def param_spec(kind, num_actions, obs_shape=None, core=None):
  """Ordered [(name, shape, init)] in the reference's variable-creation order."""
  spec = []
  def conv(name, kh, kw, cin, cout):
    spec.append((name + '/kernel', (kh, kw, cin, cout), 'glorot'))
    spec.append((name + '/bias', (cout,), 'zeros'))
  def dense(name, cin, cout, bias=True):
    spec.append((name + '/kernel', (cin, cout), 'glorot'))
    if bias:
      spec.append((name + '/bias', (cout,), 'zeros'))
  def lstm(name, cin, h):
    spec.append((name + '/kernel', (cin, 4 * h), 'glorot'))
    spec.append((name + '/recurrent_kernel', (h, 4 * h), 'orthogonal'))
    spec.append((name + '/bias', (4 * h,), 'lstm_bias'))

  if kind == 'impala_deep':           # dmlab/networks.py:63-89
    h, w, c = obs_shape or (72, 96, 3)
    cin = c
    for i, ch in enumerate([16, 32, 32]):
      conv('stack%d/conv' % i, 3, 3, cin, ch)
      for b in range(2):
        conv('stack%d/res_%d/conv2d_0' % (i, b), 3, 3, ch, ch)
        conv('stack%d/res_%d/conv2d_1' % (i, b), 3, 3, ch, ch)
      cin = ch
      h, w = (h + 1) // 2, (w + 1) // 2
    dense('conv_to_linear', h * w * cin, 256)
    lstm('core', 256 + 1 + num_actions, 256)
    dense('policy_logits', 256, num_actions)
    dense('baseline', 256, 1)
  elif kind in ('atari_shallow', 'atari_dqn_body'):
    h, w, _ = obs_shape or (84, 84, 1)
    if kind == 'atari_shallow':
      convs, fc = [(8, 4, 16), (4, 2, 32)], 256
    else:
      convs, fc = [(8, 4, 32), (4, 2, 64), (3, 1, 64)], 512
    cin = 4
    for i, (k, s, ch) in enumerate(convs):
      conv('conv%d' % i, k, k, cin, ch)
      h, w, cin = _conv_out(h, k, s, 'valid'), _conv_out(w, k, s, 'valid'), ch
    dense('fc', h * w * cin, fc)
    feat = fc
    if core == 'lstm':
      lstm('core', fc + 1 + num_actions, 256)
      feat = 256
    dense('policy_logits', feat, num_actions)
    dense('baseline', feat, 1)
  elif kind == 'r2d2':                # atari/networks.py:221-254
    h, w, _ = obs_shape or (84, 84, 1)
    cin = 4
    for i, (k, s, ch) in enumerate([(8, 4, 32), (4, 2, 64), (3, 1, 64)]):
      conv('body/conv%d' % i, k, k, cin, ch)
      h, w, cin = _conv_out(h, k, s, 'valid'), _conv_out(w, k, s, 'valid'), ch
    dense('body/fc', h * w * cin, 512)
    dense('value/hidden', 512, 512)
    dense('value/head', 512, 1)
    dense('advantage/hidden', 512, 512)
    dense('advantage/head', 512, num_actions, bias=False)
    lstm('core', 512 + 1 + num_actions, 512)
  elif kind == 'gfootball':           # football/networks.py:68-96 (lecun_normal kernels, four stacks, no LSTM)
    h, w, c = obs_shape or (72, 96, 1)
    cin = c * 16
    def conv_ln(name, cin_, cout_):
      spec.append((name + '/kernel', (3, 3, cin_, cout_), 'lecun_normal'))
      spec.append((name + '/bias', (cout_,), 'zeros'))
    for i, ch in enumerate([16, 32, 32, 32]):
      conv_ln('stack%d/conv' % i, cin, ch)
      for b in range(2):
        conv_ln('stack%d/res_%d/conv2d_0' % (i, b), ch, ch)
        conv_ln('stack%d/res_%d/conv2d_1' % (i, b), ch, ch)
      cin = ch
      h, w = (h + 1) // 2, (w + 1) // 2
    for name, ci, co in (('conv_to_linear', h * w * cin, 256), ('policy_logits', 256, num_actions), ('baseline', 256, 1)):
      spec.append((name + '/kernel', (ci, co), 'lecun_normal'))
      spec.append((name + '/bias', (co,), 'zeros'))
  elif kind == 'mlp_lstm':            # agents/vtrace/networks.py:25-52; core = (observation_size, mlp_sizes, lstm_sizes)
    obs_size, mlp_sizes, lstm_sizes = core
    cin = obs_size
    for i, m in enumerate(mlp_sizes):
      dense('mlp/dense_%d' % i, cin, m)
      cin = m
    for l, hh in enumerate(lstm_sizes):
      lstm('core/cell_%d' % l, cin, hh)
      cin = hh
    dense('policy_logits', cin, num_actions)
    dense('baseline', cin, 1)
  else:
    raise ValueError(kind)
  return spec


def init_params(spec, seed=0):
  """Keras default initialisers (glorot_uniform / orthogonal / zeros /
  unit_forget_bias) driven by numpy.random.default_rng(seed)."""
  rng = np.random.default_rng(seed)
  out = collections.OrderedDict()
  for name, shape, init in spec:
    if init == 'glorot':
      rf = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
      fan_in, fan_out = shape[-2] * rf, shape[-1] * rf
      lim = math.sqrt(6.0 / (fan_in + fan_out))
      v = rng.uniform(-lim, lim, size=shape)
    elif init == 'orthogonal':
      a = rng.standard_normal(size=(max(shape), min(shape)))
      q, r = np.linalg.qr(a)
      q = q * np.sign(np.diag(r))
      v = q if shape[0] >= shape[1] else q.T
      v = v.reshape(shape)
    elif init == 'lstm_bias':
      h = shape[0] // 4
      v = np.zeros(shape)
      v[h:2 * h] = 1.0
    elif init == 'lecun_normal':
      # Keras lecun_normal = VarianceScaling(1, 'fan_in', 'truncated_normal'): stddev sqrt(1 / fan_in) / .87962566...,
      # truncated at two standard deviations
      rf = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
      std = math.sqrt(1.0 / (shape[-2] * rf)) / .87962566103423978
      v = rng.standard_normal(size=shape)
      bad = np.abs(v) > 2.0
      while bad.any():
        v[bad] = rng.standard_normal(size=int(bad.sum()))
        bad = np.abs(v) > 2.0
      v = v * std
    else:
      v = np.zeros(shape)
    out[name] = v.astype(np.float32)
  return out


def to_torch(params, requires_grad=False, dtype=torch.float32):
  return collections.OrderedDict(
      (k, torch.tensor(v, dtype=dtype, requires_grad=requires_grad))
      for k, v in params.items())


class float64_truth(object):
  """`with float64_truth(): ...` evaluates the SAME graphs in fp64 (pass to_torch(..., dtype=torch.float64)
  parameters): every cast in this module follows torch's default dtype.  Used by tests/parity.py to tell fp32
  re-association noise (HIP and this oracle are both ~eps * sqrt(terms) away from the fp64 value, in different
  directions) from a real difference."""

  def __enter__(self):
    self._old = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)

  def __exit__(self, *exc):
    torch.set_default_dtype(self._old)


# --------------------------------------------------------------------------- #
# Keras-semantics layers.
# --------------------------------------------------------------------------- #
def conv2d(x_nhwc, kernel, bias, stride, padding):
  kh, kw = kernel.shape[0], kernel.shape[1]
  x = x_nhwc.permute(0, 3, 1, 2)
  if padding == 'same':
    assert stride == 1
    ph, pw = kh - 1, kw - 1
    x = F.pad(x, (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2))
  y = F.conv2d(x, kernel.permute(3, 2, 0, 1), bias, stride=stride)
  return y.permute(0, 2, 3, 1)


def max_pool_3x3_s2_same(x_nhwc):
  """tf.keras MaxPool2D(3, strides=2, padding='same') (Appendix A)."""
  n, h, w, c = x_nhwc.shape
  oh, ow = (h + 1) // 2, (w + 1) // 2
  ph = max((oh - 1) * 2 + 3 - h, 0)
  pw = max((ow - 1) * 2 + 3 - w, 0)
  x = x_nhwc.permute(0, 3, 1, 2)
  x = F.pad(x, (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2),
            value=float('-inf'))
  return F.max_pool2d(x, 3, 2).permute(0, 2, 3, 1)


def lstm_cell(x, h, c, kernel, recurrent_kernel, bias):
  z = x @ kernel + h @ recurrent_kernel + bias
  hdim = h.shape[1]
  i = torch.sigmoid(z[:, 0 * hdim:1 * hdim])
  f = torch.sigmoid(z[:, 1 * hdim:2 * hdim])
  g = torch.tanh(z[:, 2 * hdim:3 * hdim])
  o = torch.sigmoid(z[:, 3 * hdim:4 * hdim])
  c2 = f * c + i * g
  h2 = o * torch.tanh(c2)
  return h2, c2


def unroll_lstm(p, prefix, xs, done, state):
  """dmlab/networks.py:157-169 == atari/networks.py:176-218."""
  h, c = state
  outs = []
  for t in range(xs.shape[0]):
    keep = (~done[t]).to(torch.get_default_dtype())[:, None]
    h, c = h * keep, c * keep                      # where(done, zeros, state)
    h, c = lstm_cell(xs[t], h, c, p[prefix + '/kernel'],
                     p[prefix + '/recurrent_kernel'], p[prefix + '/bias'])
    outs.append(h)
  return torch.stack(outs), (h, c)


def stack_frames_torch(frames_u8, frame_state, done, stack_size=4):
  """atari/networks.py:57-173 in torch (frames uint8 [T,B,H,W,1])."""
  T, B = frames_u8.shape[:2]
  obs = frames_u8.shape[2:-1]
  st = frame_state.reshape((B,) + tuple(obs))
  prev = [((st >> (8 * i)) & 0xFF).to(torch.get_default_dtype())[None, ..., None]
          for i in range(stack_size - 1)]
  ext = torch.cat(prev + [frames_u8.to(torch.get_default_dtype())], 0)
  stacked = torch.cat(
      [ext[stack_size - 1 - i:ext.shape[0] - i] for i in range(stack_size)], -1)
  row = (T, B) + (1,) * (frames_u8.dim() - 2)
  masks = [torch.zeros(row, dtype=torch.bool), done.reshape(row)]
  while len(masks) < stack_size:
    pr = masks[-1]
    sh = torch.cat([torch.zeros_like(pr[:1]), pr[:-1]], 0)
    masks.append(pr | sh)
  stacked = torch.where(torch.cat(masks, -1), torch.zeros(()), stacked)
  last = stacked[-1, ..., :-1].to(torch.int32)
  shifts = torch.tensor([8 * i for i in range(stack_size - 2, -1, -1)],
                        dtype=torch.int32)
  new_state = (last << shifts).sum(-1).to(torch.int32).reshape(B, -1)
  return stacked, new_state


# --------------------------------------------------------------------------- #
# Agents.
# --------------------------------------------------------------------------- #
def unpackbits(frame_u16):
  """football/observation.py:48-63: every 16-bit word -> 16 channels of 0 / 255 in the order 2^7..2^0, 2^15..2^8."""
  pats = torch.tensor([2 ** 7, 2 ** 6, 2 ** 5, 2 ** 4, 2 ** 3, 2 ** 2, 2 ** 1, 2 ** 0, 2 ** 15, 2 ** 14, 2 ** 13, 2 ** 12,
                       2 ** 11, 2 ** 10, 2 ** 9, 2 ** 8], dtype=torch.int32)
  f = frame_u16.to(torch.int32) & 0xFFFF
  bits = (f[..., None] & pats) != 0
  out = bits.to(torch.get_default_dtype()) * 255
  return out.reshape(tuple(frame_u16.shape[:-1]) + (frame_u16.shape[-1] * 16,))


def impala_deep_torso(p, frames_u8, num_stacks=3):
  """dmlab/networks.py:94-109 on [N,H,W,C] uint8 (or already-float 0..255 planes: football/networks.py:98-116)."""
  x = frames_u8.to(torch.get_default_dtype()) / 255
  for i in range(num_stacks):
    x = conv2d(x, p['stack%d/conv/kernel' % i], p['stack%d/conv/bias' % i], 1, 'same')
    x = max_pool_3x3_s2_same(x)
    for b in range(2):
      blk = x
      x = F.relu(x)
      x = conv2d(x, p['stack%d/res_%d/conv2d_0/kernel' % (i, b)],
                 p['stack%d/res_%d/conv2d_0/bias' % (i, b)], 1, 'same')
      x = F.relu(x)
      x = conv2d(x, p['stack%d/res_%d/conv2d_1/kernel' % (i, b)],
                 p['stack%d/res_%d/conv2d_1/bias' % (i, b)], 1, 'same')
      x = x + blk
  x = F.relu(x)
  x = x.reshape(x.shape[0], -1)
  x = F.relu(x @ p['conv_to_linear/kernel'] + p['conv_to_linear/bias'])
  return x


def impala_deep_unroll(p, num_actions, prev_actions, reward, done, frames_u8,
                       core_state):
  """dmlab/networks.py:135-171 with unroll=True.  Time-major [T1,B,...]."""
  T1, B = done.shape
  feat = impala_deep_torso(p, frames_u8.reshape((T1 * B,) + frames_u8.shape[2:]))
  clipped = torch.clamp(reward.reshape(-1), -1, 1)[:, None]
  onehot = F.one_hot(prev_actions.reshape(-1).long(), num_actions).to(torch.get_default_dtype())
  xs = torch.cat([feat, clipped, onehot], 1).reshape(T1, B, -1)
  core, state = unroll_lstm(p, 'core', xs, done, core_state)
  flat = core.reshape(T1 * B, -1)
  logits = (flat @ p['policy_logits/kernel'] + p['policy_logits/bias'])
  baseline = (flat @ p['baseline/kernel'] + p['baseline/bias'])[:, 0]
  return logits.reshape(T1, B, -1), baseline.reshape(T1, B), state


def gfootball_unroll(p, num_actions, frames_packed):
  """football/networks.py:98-150 with unroll=True: frames int16/uint16 [T1, B, H, W, planes] (packed bits)."""
  T1, B = frames_packed.shape[:2]
  x = unpackbits(frames_packed.reshape((T1 * B,) + tuple(frames_packed.shape[2:])))
  feat = impala_deep_torso(p, x, num_stacks=4)
  logits = feat @ p['policy_logits/kernel'] + p['policy_logits/bias']
  baseline = (feat @ p['baseline/kernel'] + p['baseline/bias'])[:, 0]
  return logits.reshape(T1, B, -1), baseline.reshape(T1, B)


def mlp_lstm_unroll(p, num_mlp, num_lstm, observation, done, core_state):
  """agents/vtrace/networks.py:99-121: Dense+relu stack, then StackedRNNCells stepped over time with the done-reset of
  EVERY cell's state before each step; core_state = ((h, c), ...) per cell."""
  T1, B = done.shape
  x = observation.reshape(T1 * B, -1).to(torch.get_default_dtype())
  for i in range(num_mlp):
    x = F.relu(x @ p['mlp/dense_%d/kernel' % i] + p['mlp/dense_%d/bias' % i])
  xs = x.reshape(T1, B, -1)
  state = [tuple(s) for s in core_state]
  outs = []
  for t in range(T1):
    keep = (~done[t]).to(torch.get_default_dtype())[:, None]
    inp = xs[t]
    for l in range(num_lstm):
      h, c = state[l]
      h, c = lstm_cell(inp, h * keep, c * keep, p['core/cell_%d/kernel' % l], p['core/cell_%d/recurrent_kernel' % l],
                       p['core/cell_%d/bias' % l])
      state[l] = (h, c)
      inp = h
    outs.append(inp)
  flat = torch.stack(outs).reshape(T1 * B, -1)
  logits = flat @ p['policy_logits/kernel'] + p['policy_logits/bias']
  baseline = (flat @ p['baseline/kernel'] + p['baseline/bias'])[:, 0]
  return logits.reshape(T1, B, -1), baseline.reshape(T1, B), tuple(state)


def atari_body(p, x, prefix, convs):
  for i, (s,) in enumerate(convs):
    x = F.relu(conv2d(x, p['%sconv%d/kernel' % (prefix, i)],
                      p['%sconv%d/bias' % (prefix, i)], s, 'valid'))
  x = x.reshape(x.shape[0], -1)
  return F.relu(x @ p[prefix + 'fc/kernel'] + p[prefix + 'fc/bias'])


def atari_shallow_unroll(p, kind, num_actions, prev_actions, reward, done,
                         frames_u8, frame_state, core_state=None):
  """AtariShallow / atari_dqn_body agent (see module docstring, D1)."""
  T1, B = done.shape
  stacked, new_fs = stack_frames_torch(frames_u8, frame_state, done, 4)
  x = (stacked / 255).reshape((T1 * B,) + stacked.shape[2:])
  convs = [(4,), (2,)] if kind == 'atari_shallow' else [(4,), (2,), (1,)]
  feat = atari_body(p, x, '', convs)
  state = None
  if 'core/kernel' in p:
    clipped = torch.clamp(reward.reshape(-1), -1, 1)[:, None]
    onehot = F.one_hot(prev_actions.reshape(-1).long(), num_actions).to(torch.get_default_dtype())
    xs = torch.cat([feat, clipped, onehot], 1).reshape(T1, B, -1)
    core, state = unroll_lstm(p, 'core', xs, done, core_state)
    feat = core.reshape(T1 * B, -1)
  logits = feat @ p['policy_logits/kernel'] + p['policy_logits/bias']
  baseline = (feat @ p['baseline/kernel'] + p['baseline/bias'])[:, 0]
  return logits.reshape(T1, B, -1), baseline.reshape(T1, B), new_fs, state


def r2d2_unroll(p, num_actions, prev_actions, reward, done, frames_u8,
                frame_state, core_state):
  """atari/networks.py:256-340 with unroll=True, stack_size=4."""
  T1, B = done.shape
  stacked, new_fs = stack_frames_torch(frames_u8, frame_state, done, 4)
  x = (stacked / 255).reshape((T1 * B,) + stacked.shape[2:])
  feat = atari_body(p, x, 'body/', [(4,), (2,), (1,)])
  onehot = F.one_hot(prev_actions.reshape(-1).long(), num_actions).to(torch.get_default_dtype())
  xs = torch.cat([feat, reward.reshape(-1)[:, None], onehot], 1).reshape(T1, B, -1)
  core, state = unroll_lstm(p, 'core', xs, done, core_state)
  flat = core.reshape(T1 * B, -1)
  v = F.relu(flat @ p['value/hidden/kernel'] + p['value/hidden/bias'])
  v = v @ p['value/head/kernel'] + p['value/head/bias']
  a = F.relu(flat @ p['advantage/hidden/kernel'] + p['advantage/hidden/bias'])
  a = a @ p['advantage/head/kernel']
  a = a - a.mean(-1, keepdim=True)
  q = (v + a).reshape(T1, B, -1)
  return R2D2Output(q.argmax(-1).to(torch.int32), q), new_fs, state


# --------------------------------------------------------------------------- #
# Loss (torch, differentiable) + Keras Adam — the eager CPU learner step.
# --------------------------------------------------------------------------- #
def vtrace_torch(tgt_lp, beh_lp, discounts, rewards, values, bootstrap,
                 clip_rho=1.0, clip_pg_rho=1.0, lambda_=1.0):
  """common/vtrace.py:84-148 (same op order)."""
  with torch.no_grad():
    rhos = torch.exp(tgt_lp - beh_lp)
    crho = torch.clamp(rhos, max=clip_rho) if clip_rho is not None else rhos
    cs = torch.clamp(rhos, max=1.0) * lambda_
    v1 = torch.cat([values[1:], bootstrap[None]], 0)
    deltas = crho * (rewards + discounts * v1 - values)
    acc = torch.zeros_like(bootstrap)
    out = []
    for i in range(discounts.shape[0] - 1, -1, -1):
      acc = deltas[i] + discounts[i] * cs[i] * acc
      out.append(acc)
    vs = torch.stack(out[::-1]) + values
    vs1 = torch.cat([vs[1:], bootstrap[None]], 0)
    cpg = torch.clamp(rhos, max=clip_pg_rho) if clip_pg_rho is not None else rhos
    pg = cpg * (rewards + discounts * vs1 - values)
  return vs, pg


def impala_loss_torch(logits, baseline, beh_logits, actions, rewards, done,
                      entropy_cost=0.00025, baseline_cost=0.5, kl_cost=0.0,
                      discounting=0.99, lambda_=1.0, max_abs_reward=0.0,
                      entropy_cost_param=None, entropy_cost_adjustment_speed=10.0, target_entropy=None):
  """agents/vtrace/learner.py:82-135, unfused eager ops.  entropy_cost_param (a 0-d tensor, usually with
  requires_grad): the learner's learnable entropy cost exp(speed * param) (learner.py:225-234) with the
  Lagrange-style adjustment loss of :127-132 when target_entropy is set."""
  bootstrap = baseline[-1]
  tl, bl, vals = logits[:-1], beh_logits[:-1], baseline[:-1]
  act = actions[:-1].long()
  rew, dn = rewards[1:], done[1:]
  if max_abs_reward:
    rew = torch.clamp(rew, -max_abs_reward, max_abs_reward)
  disc = (~dn).to(torch.get_default_dtype()) * discounting
  tls = F.log_softmax(tl, -1)
  tgt_lp = tls.gather(-1, act[..., None])[..., 0]
  beh_lp = F.log_softmax(bl, -1).gather(-1, act[..., None])[..., 0]
  vs, pg = vtrace_torch(tgt_lp, beh_lp, disc, rew, vals, bootstrap,
                        lambda_=lambda_)
  policy_loss = -(tgt_lp * pg).mean()
  v_loss = baseline_cost * 0.5 * ((vs - vals) ** 2).mean()
  entropy = (-(tls.exp() * tls).sum(-1)).mean()
  kl_loss = kl_cost * (beh_lp - tgt_lp).mean()
  adjustment = 0.0
  if entropy_cost_param is not None:
    cost = torch.exp(entropy_cost_adjustment_speed * entropy_cost_param)          # learner.py:233
    entropy_cost = cost.detach()                                                  # :121 stop_gradient
    if target_entropy:
      adjustment = cost * (entropy.detach() - target_entropy)                     # :128-130
    else:
      adjustment = 0. * cost                                                      # :131-132
  total = policy_loss + v_loss + entropy_cost * -entropy + kl_loss + adjustment
  return total, dict(policy_loss=policy_loss, v_loss=v_loss, entropy=entropy,
                     kl_loss=kl_loss, vs=vs, pg_advantages=pg, entropy_cost=entropy_cost,
                     entropy_adjustment_loss=adjustment)


class KerasAdam:
  """tf.keras.optimizers.Adam (OptimizerV2) dense update, per variable
  (Appendix A): lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m += (g-m)(1-b1);
  v += (g^2-v)(1-b2); p -= lr_t*m/(sqrt(v)+eps)."""

  def __init__(self, params, lr_fn, beta_1=0.9, beta_2=0.999, epsilon=1e-7):
    self.params = list(params)
    self.lr_fn, self.b1, self.b2, self.eps = lr_fn, beta_1, beta_2, epsilon
    self.m = [torch.zeros_like(p) for p in self.params]
    self.v = [torch.zeros_like(p) for p in self.params]
    self.iterations = 0

  def apply_gradients(self, grads):
    t = self.iterations + 1
    lr = self.lr_fn(self.iterations)
    lr_t = lr * math.sqrt(1 - self.b2 ** t) / (1 - self.b1 ** t)
    with torch.no_grad():
      for p, g, m, v in zip(self.params, grads, self.m, self.v):
        m.add_((g - m) * (1 - self.b1))
        v.add_((g * g - v) * (1 - self.b2))
        p.sub_(lr_t * m / (v.sqrt() + self.eps))
    self.iterations = t


def polynomial_decay(lr0, decay_steps, end_lr=0.0, power=1.0):
  """tf.keras.optimizers.schedules.PolynomialDecay (dmlab/vtrace_main.py:47-48)."""
  def fn(step):
    s = min(step, decay_steps)
    return (lr0 - end_lr) * (1 - s / decay_steps) ** power + end_lr
  return fn


def r2d2_loss_torch(training_q, target_q, actions, rewards, done, importance_weights=None, gamma=0.997,
                    n_steps=5, eta=0.9, eps=1e-3):
  """agents/r2d2/learner.py:258-330 + :604 in torch (differentiable wrt training_q)."""
  def h(x):
    return torch.sign(x) * (torch.sqrt(torch.abs(x) + 1.) - 1.) + eps * x
  def h_inv(x):
    return torch.sign(x) * (torch.square((torch.sqrt(1. + 4. * eps * (torch.abs(x) + 1. + eps)) - 1.) / (2. * eps)) - 1.)
  T, B, A = training_q.shape
  act = actions.long()
  replay_q = training_q.gather(-1, act[..., None])[..., 0]
  best = training_q.argmax(-1)
  qmax = h_inv(target_q.gather(-1, best[..., None])[..., 0])
  bt = torch.cat([torch.zeros_like(qmax[0:1]), qmax] + [qmax[-1:] / gamma ** k for k in range(1, n_steps)], 0)
  d = torch.cat([done] + [torch.zeros_like(done[0:1])] * n_steps, 0)
  r = torch.cat([rewards] + [torch.zeros_like(rewards[0:1])] * n_steps, 0)
  for _ in range(n_steps):
    r, d = r[:-1], d[:-1]
    bt = r + gamma * (1. - d.to(torch.get_default_dtype())) * bt[1:]
  bt = h(bt[1:].detach())
  abs_td = torch.abs(bt - replay_q[:-1])
  prio = eta * abs_td.max(0)[0] + (1 - eta) * abs_td.mean(0)
  loss = 0.5 * torch.sum(abs_td ** 2, 0)
  w = torch.ones(B) if importance_weights is None else importance_weights
  return (loss * w).mean(), loss, prio
