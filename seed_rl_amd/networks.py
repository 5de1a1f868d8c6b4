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
import os

import numpy as np
import torch

from seed_rl_amd import _lib, ops
from seed_rl_amd.flat import FlatParams

AgentOutput = collections.namedtuple('AgentOutput', 'action policy_logits baseline')
# A frame-stacking state that is NOT a [B, HW] tensor but rows of a per-environment table, read and updated in place
# by the torso (csrc/frames.hip: *_indexed): column b's state is table[rows[b]]; zero_mask[b] != 0: counts as zeros
# (restarted actor); valid_mask[b] == 0: the row is not written back.  Passed by inference.FusedInferenceState.
IndexedFrameState = collections.namedtuple('IndexedFrameState', 'table rows zero_mask valid_mask')
_RELU_BITS = os.environ.get('SEEDHIP_RELU_BITS', '1') == '1'      # A/B knob, see _AtariTorso._torso_fwd
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
    elif init == 'lecun_normal':
      # Keras VarianceScaling(scale=1, mode='fan_in', distribution='truncated_normal'): N(0, sqrt(1/fan_in)/.8796...)
      # truncated at two standard deviations (football/networks.py:33-49,86-96)
      rf = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
      std = math.sqrt(1.0 / (shape[-2] * rf)) / .87962566103423978
      v = rng.standard_normal(size=shape)
      bad = np.abs(v) > 2.0
      while bad.any():                               # resample the tails (deterministic given the generator)
        v[bad] = rng.standard_normal(size=int(bad.sum()))
        bad = np.abs(v) > 2.0
      v = v * std
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
    self._lstm_ctx = {}
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
    shapes = dict((n, s) for n, s, _ in ref_spec)
    if 'policy_logits/kernel' in shapes:
      feat = shapes['policy_logits/kernel'][0]
      internal.append(('heads/kernel', (feat, self._ldh)))
      internal.append(('heads/bias', (self._ldh,)))
      # slot of the learner's learnable entropy cost (agents/vtrace/learner.py:225-234); inert (value 0, gradient 0)
      # until a Learner attaches it: attach_entropy_cost_param()
      internal.append(('entropy_cost_param', (1,)))
    self.flat = FlatParams(internal, self.device)
    self.load_reference_params(keras_init(ref_spec, seed))
    self._ec_speed = None

  # -- entropy cost (agents/vtrace/learner.py:121,127-135,225-234) ----------------------------------------------- #
  def has_own_entropy_cost(self):
    """True for an agent constructed with an explicit entropy_cost (the reference's `hasattr(agent, 'entropy_cost')`):
    the learner then leaves it alone; otherwise it attaches the learnable parameter."""
    return getattr(self, '_entropy_cost', None) is not None

  def attach_entropy_cost_param(self, entropy_cost, adjustment_speed):
    """learner.py:225-234: entropy_cost_param = log(FLAGS.entropy_cost) / speed, a TRAINABLE scalar constrained to
    [-20/speed, 20/speed]; entropy_cost() = exp(speed * param).  The parameter is one more element of the flat buffer
    (Adam and the gradient all-reduce see it like any other), its constraint is applied by the Adam kernel."""
    mul = np.float32(adjustment_speed)
    self._ensure_entropy_cost_in_spec()
    self._ec_speed = float(mul)
    if not getattr(self, '_ec_loaded', False):       # a value restored from a checkpoint BEFORE the learner existed stays
      with torch.no_grad():
        self.flat.p('entropy_cost_param').fill_(float(np.log(np.float32(entropy_cost)) / mul))
    self.flat.constraint = (self.flat.offsets['entropy_cost_param'], float(-20.0 / mul), float(20.0 / mul))

  def _ensure_entropy_cost_in_spec(self):
    if not any(n == 'entropy_cost_param' for n, _, _ in self._ref_spec):
      self._ref_spec = list(self._ref_spec) + [('entropy_cost_param', (1,), 'zeros')]

  def entropy_cost_param(self):
    """(param, d_param, speed) device scalars of the attached learnable entropy cost, or None."""
    if self._ec_speed is None:
      return None
    return self.flat.p('entropy_cost_param'), self.flat.g('entropy_cost_param'), self._ec_speed

  def entropy_cost(self):
    """agent.entropy_cost() of the reference: the agent's own constant, or exp(speed * param) as a device tensor."""
    if self.has_own_entropy_cost():
      return self._entropy_cost
    if self._ec_speed is not None:
      return torch.exp(self._ec_speed * self.flat.p('entropy_cost_param'))[0]
    return None

  def load_reference_params(self, values):
    """values: {reference variable name: numpy array} (Keras layouts)."""
    A = self._num_actions
    if 'entropy_cost_param' in values and 'entropy_cost_param' in self.flat.offsets and not self.has_own_entropy_cost():
      # a checkpoint of a learner with the learnable entropy cost, restored in ANY order relative to Learner():
      # the value is kept by a later attach_entropy_cost_param() and its Adam slots are addressable from now on
      self._ensure_entropy_cost_in_spec()
      self._ec_loaded = True
    with torch.no_grad():
      for name, shape, _ in self._ref_spec:
        if name == 'entropy_cost_param' and name not in values:
          continue                                   # checkpoints written before the parameter was attached
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
  grad_ready_hook = None   # set by the learner: callable(lo, hi) on ranges of flat.grads that are final

  def _grads_ready_from(self, first_name, upto=None):
    """Reports flat.grads[offset(first_name) : upto or end] as final (everything after `first_name` in creation
    order must already have its gradient)."""
    hook = self.grad_ready_hook
    if hook is not None:
      lo = self.flat.offsets[first_name] if first_name is not None else 0
      hook(lo, self.flat.size if upto is None else self.flat.offsets[upto])

  def _buf(self, key, shape, dtype=torch.float32, zero=False):
    k = (key, tuple(shape), dtype)
    t = self._ws.get(k)
    if t is None:
      t = (torch.zeros if zero else torch.empty)(tuple(shape), dtype=dtype, device=self.device)
      self._ws[k] = t
    return t

  def get_action(self, *args, **kwargs):
    return self.__call__(*args, **kwargs)

  def inference_twin(self):
    """A second handle on the SAME parameters (one flat buffer: every optimizer step is visible at once, no weight
    copy) with its OWN workspaces, activations and sampler state, for central inference on another stream beside the
    train step -- the reference runs inference on its own devices against the shared variables
    (agents/vtrace/learner.py:350-411).  A forward that overlaps an Adam launch may read some tensors before and some
    after the update; the behaviour logits it returns are the ones it acted with, which is what V-trace corrects for."""
    import copy
    t = copy.copy(self)
    t._ws, t._lstm_ctx, t._last = {}, {}, None
    t._last_lstm, t._seq_flag, t._seq_event, t._rng = None, None, None, None
    t.grad_ready_hook = None
    t.frames_slot = 0
    return t

  # -- head -------------------------------------------------------------------- #
  def _head_fwd(self, feat, rows, feat_dim, ld_feat=None):
    ldh = self._ldh
    head = self._buf('head', (rows, ldh))
    if self._skinny_heads(feat_dim):              # N <= 32: one skinny GEMM (csrc/heads.hip) instead of the general core
      ops.heads_fwd(feat, ld_feat or feat_dim, self.flat.p('heads/kernel'), self.flat.p('heads/bias'), rows, feat_dim,
                    ldh, head)
      return head
    g = ops.dense_geom(rows, feat_dim, ldh, ld_in=ld_feat or feat_dim)
    ops.conv2d_fwd(g, feat, self.flat.p('heads/kernel'), self.flat.p('heads/bias'), head)
    return head

  def _skinny_heads(self, feat_dim):
    return ops.heads_supported(feat_dim, self._ldh)

  def _head_bwd(self, feat, feat_dim, d_head, rows, dx, relu_mask, wsb):
    """Gradients of the packed heads: heads/kernel, heads/bias into the flat gradient buffer and dx [rows, feat_dim] =
    d_head W^T (zeroed where feat <= 0 when relu_mask: feat is then the ReLU output the heads read)."""
    fl = self.flat
    gh = ops.dense_geom(rows, feat_dim, self._ldh)
    ops.conv2d_bwd_weight(gh, feat, d_head, fl.g('heads/kernel'), fl.g('heads/bias'), wsb)
    ops.conv2d_bwd_data(gh, d_head, fl.p('heads/kernel'), dx, relu_mask=feat if relu_mask else None)

  def rng_state(self):
    """Device uint64[2] = (seed, call counter) of the action sampler; the kernels advance the counter themselves, so
    sampling sits inside captured HIP graphs."""
    if getattr(self, '_rng', None) is None:
      self._rng = torch.tensor([self._sample_seed, 0], dtype=torch.int64, device=self.device)
    return self._rng

  def seed_sampler(self, seed):
    self._sample_seed = int(seed)
    self._rng = None

  _sample_seed = 0x5EED

  def _sample(self, head, rows):
    """tfd.Categorical(logits).sample() of the reference heads (dmlab/networks.py:122): one kernel over the head-GEMM
    output rows [rows, ldh] (logits in columns 0..A-1), Gumbel-max over counter-based randoms."""
    action = torch.empty(rows, dtype=torch.int64, device=self.device)
    ops.categorical_sample(head, self._ldh, rows, self._num_actions, self.rng_state(), action)
    return action

  def _agent_output(self, head, T1, B, sample):
    A = self._num_actions
    h3 = head.view(T1, B, self._ldh)
    logits, baseline = h3[..., :A], h3[..., A]
    action = self._sample(head, T1 * B).view(T1, B) if sample else None
    return AgentOutput(action, logits, baseline)


  # head-gradient buffer the fused loss kernel writes into (same layout as `head`)
  def head_buffers(self):
    L = self._last
    d_head = self._buf('d_head', (L['N'], self._ldh), zero=True)
    return L['head'], d_head, self._ldh

  # -- LSTM core with done-reset (dmlab/networks.py:152-171; atari/networks.py:176-218) ---- #
  def _lstm_fwd(self, X, ldx, in_dim, H, T1, B, done_u8, state, prefix='core', x_pad_zero=False):
    """X [T1*B, ldx] (features | reward | one-hot action); returns core outputs [T1*B, H] and
    the new (h, c).  Keeps what the backward needs in self._last['lstm'].
    x_pad_zero: the caller zeroed columns in_dim .. ldx - 1 of X (lstm_assemble_inputs does).  The input projection of
    a training-sized batch then runs as an [N, ldx] x [ldx, 4H] product on a zero-padded copy of the kernel: an input
    width that is not a multiple of 4 (512 + 1 + 18 = 531, 256 + 1 + 9 = 266) keeps the layer off the bf16x6 GEMMs
    (xgemm.h / xgemm8.h move 16-byte vectors of four k), and a zero column times a zero row adds exactly nothing."""
    N = T1 * B
    fl = self.flat
    Wx, pad = fl.p(prefix + '/kernel'), None
    kdim = in_dim
    if x_pad_zero and in_dim % 4 and ldx == _round4(in_dim) and N >= 4096 and os.environ.get('SEEDHIP_LSTM_PADK', '1') != '0':
      pad = self._buf(prefix + '/kernel_pad', (ldx, 4 * H), zero=True)         # (rows in_dim .. stay zero)
      pad[:in_dim].copy_(Wx)
      Wx, kdim = pad, ldx
    gx = ops.dense_geom(N, kdim, 4 * H, ld_in=ldx)
    Zx = self._buf(prefix + '/lstm_zx', (N, 4 * H))
    ops.conv2d_fwd(gx, X, Wx, fl.p(prefix + '/bias'), Zx)
    Hin = self._buf(prefix + '/lstm_hin', (T1 + 1, B, H))
    Cin = self._buf(prefix + '/lstm_cin', (T1 + 1, B, H))
    h0, c0 = state
    ops.lstm_mask_state(h0.contiguous(), c0.contiguous(), done_u8[0], B, H, Hin[0], Cin[0])
    Z = self._buf(prefix + '/lstm_z', (T1, B, 4 * H))
    Hout = self._buf(prefix + '/lstm_hout', (N, H))
    gu = ops.dense_geom(B, H, 4 * H)
    U = fl.p(prefix + '/recurrent_kernel')
    Zx3, Hout3 = Zx.view(T1, B, 4 * H), Hout.view(T1, B, H)
    fused_step = ops.lstm_step_supported(B, H)
    if fused_step:                                # recurrent GEMM + gates + reset in one launch per step
      Up = self._buf(prefix + '_u_perm', (H, 4 * H))
      ops.lstm_permute_u(U, H, Up)
    if fused_step and not torch.cuda.is_current_stream_capturing() and getattr(self, '_seq_flag', None) is not None:
      self._lstm_seq_check()                      # a timed-out wait of an EARLIER step demotes this agent before it picks
    seq_ok = fused_step and os.environ.get('SEEDHIP_LSTM_SEQ', '1') != '0' and ops.lstm_seq_supported(T1, B, H) and \
        not getattr(self, '_seq_demoted', False)
    fused_seq = seq_ok and os.environ.get('SEEDHIP_LSTM_SEQ_FWD', '1') != '0'
    if fused_seq:                                 # the whole unroll in one launch (resident workgroups + grid barrier)
      capturing = torch.cuda.is_current_stream_capturing()
      if not capturing:
        self._lstm_seq_check()
      ops.lstm_seq_fwd(Up, Zx3, done_u8, T1, B, H, Z, Hout, H, Hin, Cin, self._seq_sync('fwd'), self._seq_sticky())
      if not capturing:
        self.mirror_error_flags()
    for t in range(0 if not fused_seq else T1, T1):
      done_next = done_u8[t + 1] if t + 1 < T1 else None
      if fused_step:
        ops.lstm_step_fwd(Hin[t], Up, Zx3[t], Cin[t], done_next, B, H, Z[t], Hout3[t], H, Hin[t + 1], Cin[t + 1])
      else:
        ops.conv2d_fwd(gu, Hin[t], U, None, Z[t], residual=Zx3[t])
        ops.lstm_gates_fwd(Z[t], Cin[t], done_next, B, H, Hout3[t], H, Hin[t + 1], Cin[t + 1])
    self._last_lstm = dict(X=X, ldx=ldx, in_dim=in_dim, H=H, T1=T1, B=B, done=done_u8, gx=gx, gu=gu, Z=Z, Hin=Hin,
                           Cin=Cin, Hout=Hout, prefix=prefix, fused_seq=seq_ok, Wx=Wx, padded=pad is not None)
    self._lstm_ctx[prefix] = self._last_lstm       # stacked cores (MLPandLSTM): one context per layer
    return Hout, (Hin[T1].clone(), Cin[T1].clone())

  def _seq_sync(self, which):
    """int32[2] (arrival counter, abort flag) of the sequence kernels; one pair per direction for the whole agent (the
    layers of a stacked core run one after the other on one stream)."""
    return self._buf('lstm_seq_sync' if which == 'fwd' else 'lstm_seq_sync_bwd', (2,), torch.int32)

  def _seq_sticky(self):
    """int32[1], set to 1 by a sequence kernel whose wait timed out and cleared only by `_lstm_seq_check` (every launch
    zeroes its own sync pair, so the pair of a stacked core's first layer would be lost behind the second layer's launch).
    Registered as the flat buffer's `step_guard`: the optimizer's update kernel drops a step while it is set."""
    t = self._buf('lstm_seq_sticky', (1,), torch.int32, zero=True)
    if getattr(self.flat, 'step_guard', None) is None:
      self.flat.step_guard = t
    return t

  def mirror_error_flags(self):
    """Asynchronous device -> pinned-host copy of the sequence kernels' abort flags (forward, backward) on the current
    stream.  Called after every eager launch, and by learner.GraphedStep after every graph replay (inside a captured
    graph the launches cannot do it themselves): `_lstm_seq_check()` / `check_errors()` then see a timed-out wait of a
    REPLAYED step too, one step late and without a host sync."""
    if getattr(self, '_seq_flag', None) is None:
      return
    t = self._ws.get(('lstm_seq_sticky', (1,), torch.int32))
    if t is not None:
      self._seq_flag[0:1].copy_(t, non_blocking=True)
    self._seq_event.record()

  def check_errors(self):
    """Blocking check of the LSTM sequence kernels' abort flag: True if this call demoted the agent to the per-step
    kernels (a bounded wait had timed out; the steps concerned were dropped on the device, see _lstm_seq_check)."""
    if getattr(self, '_seq_flag', None) is not None:
      self.mirror_error_flags()
      return self._lstm_seq_check(wait=True)
    return False

  def _lstm_seq_check(self, wait=False):
    """The persistent sequence kernels assume their grid is co-resident; when something else holds CUs (an inference
    stream beside the train stream) a bounded wait can time out.  The kernel then sets a STICKY device word: the
    optimizer's update kernel sees it and drops that step (parameters and moments untouched -- the garbage gradients
    are never applied, inside a replayed HIP graph too), and its mirror in pinned host memory is looked at here --
    before the next launch (by then the copy has landed: no sync) or, with wait=True, after waiting for the copy.
    Seen set, the agent is DEMOTED to one launch per LSTM step (`seedhip_lstm_step_fwd`, no residency assumption) for
    the rest of its life, the word is cleared, and training goes on (logged once; VERDICT r3 task 9: it used to raise
    mid-training).  Returns True when this call demoted the agent (learner.GraphedStep re-captures its graphs)."""
    if getattr(self, '_seq_flag', None) is None:
      self._seq_flag = torch.zeros(2, dtype=torch.int32).pin_memory()
      self._seq_event = torch.cuda.Event()
      return False
    if wait:
      self._seq_event.synchronize()
    if self._seq_event.query() and int(self._seq_flag[0]) != 0:
      import sys
      self._seq_flag.zero_()
      t = self._ws.get(('lstm_seq_sticky', (1,), torch.int32))
      if t is not None:
        t.zero_()                                  # stream-ordered: behind every step that saw it set
      if getattr(self, '_seq_demoted', False):
        # already demoted: a graph captured BEFORE the demotion (another unroll slot's) replayed its sequence kernels
        # and timed out again.  Clear the word all the same -- left set, the update kernel would drop every later step
        # of every slot without an error -- and say so; learner.GraphedStep re-captures that graph.
        sys.stderr.write('seed_rl_amd: a graph captured before the demotion replayed an LSTM sequence kernel that timed out '
                         'again; step dropped, sticky word cleared\n')
        return False
      self._seq_demoted = True
      sys.stderr.write('seed_rl_amd: an LSTM sequence kernel timed out waiting for its co-resident grid (another stream '
                       'held CUs); the affected step(s) were dropped, this agent now runs one launch per LSTM step\n')
      return True
    return False

  def _lstm_bwd(self, dHout, wsb, prefix=None, relu_mask_x=True):
    """dHout [T1*B, H]: gradient wrt the core outputs.  Fills the core's weight gradients and returns
    dX [T1*B, ldx] (columns >= in_dim undefined), already masked by X > 0 (ReLU of the feature layer;
    reward / one-hot columns carry no gradient we use)."""
    L = self._last_lstm if prefix is None else self._lstm_ctx[prefix]
    T1, B, H, N = L['T1'], L['B'], L['H'], L['T1'] * L['B']
    fl, prefix = self.flat, L['prefix']
    U = fl.p(prefix + '/recurrent_kernel')
    dZ = self._buf(prefix + '/lstm_dz', (T1, B, 4 * H))
    dcb = [self._buf(prefix + '/lstm_dc0', (B, H)), self._buf(prefix + '/lstm_dc1', (B, H))]
    dhb = self._buf(prefix + '/lstm_dh', (B, H))
    dH3 = dHout.view(T1, B, H)
    dh_rec = dc_rec = None
    fused_seq = L['fused_seq'] and os.environ.get('SEEDHIP_LSTM_SEQ_BWD', '1') != '0'
    if fused_seq:                                 # the whole recurrence in one launch (same residency as the forward)
      capturing = torch.cuda.is_current_stream_capturing()
      if not capturing:
        self._lstm_seq_check()
      ring = self._buf(prefix + '/lstm_seq_ring', (ops.lstm_seq_bwd_workspace_bytes(B, H) // 4,))
      sync = self._seq_sync('bwd')
      ops.lstm_seq_bwd(self._buf(prefix + '_u_perm', (H, 4 * H)), L['Z'], L['Cin'], dHout, H, L['done'], T1, B, H, dZ,
                       ring, sync, self._seq_sticky())
      if not capturing:
        self.mirror_error_flags()
    for t in range(T1 - 1, -1 if not fused_seq else T1 - 1, -1):
      ops.lstm_gates_bwd(L['Z'][t], L['Cin'][t], dH3[t], H, dh_rec, dc_rec,
                         L['done'][t + 1] if t + 1 < T1 else None, B, H, dZ[t], dcb[t & 1])
      dc_rec = dcb[t & 1]
      if t > 0:
        ops.conv2d_bwd_data(L['gu'], dZ[t], U, dhb)
        dh_rec = dhb
    dZf = dZ.view(N, 4 * H)
    gall = ops.dense_geom(N, H, 4 * H)
    ops.conv2d_bwd_weight(gall, L['Hin'][:T1].view(N, H), dZf, fl.g(prefix + '/recurrent_kernel'), None, wsb)
    if L['padded']:                               # gradient of the padded kernel; its rows < in_dim are the layer's
      gpad = self._buf(prefix + '/kernel_pad_grad', (L['ldx'], 4 * H))
      ops.conv2d_bwd_weight(L['gx'], L['X'], dZf, gpad, fl.g(prefix + '/bias'), wsb)
      fl.g(prefix + '/kernel').copy_(gpad[:L['in_dim']])
    else:
      ops.conv2d_bwd_weight(L['gx'], L['X'], dZf, fl.g(prefix + '/kernel'), fl.g(prefix + '/bias'), wsb)
    dX = self._buf(prefix + '/lstm_dx', (N, L['ldx']))
    ops.conv2d_bwd_data(L['gx'], dZf, L['Wx'], dX, relu_mask=L['X'] if relu_mask_x else None)
    return dX

  def _lstm_ws_bytes(self, prefix=None):
    L = self._last_lstm if prefix is None else self._lstm_ctx[prefix]
    N = L['T1'] * L['B']
    return max(ops.conv2d_bwd_weight_workspace_bytes(ops.dense_geom(N, L['H'], 4 * L['H'])),
               ops.conv2d_bwd_weight_workspace_bytes(L['gx']))


class _AtariTorso(object):
  """Learner-side frame stacking (atari/networks.py:57-173) + /255 (:330) + VALID conv body + Dense(ReLU):
  shared by AtariShallow and DuelingLSTMDQNNet.  The first conv reads the uint8 frames directly."""

  def _torso_init(self, observation_shape, convs, fc, prefix=''):
    h, w, c = observation_shape
    if c != 1:
      raise ValueError('frame stacking needs a single-channel observation (atari/networks.py:86-88)')
    self._obs = (h, w)
    self._convs, self._fc, self._tp = convs, fc, prefix
    spec, cin = [], 4
    self._shapes = []                       # (ih, iw, cin, k, s, cout, oh, ow)
    for i, (k, s, ch) in enumerate(convs):
      spec += [('%sconv%d/kernel' % (prefix, i), (k, k, cin, ch), 'glorot'), ('%sconv%d/bias' % (prefix, i), (ch,), 'zeros')]
      oh, ow = (h - k) // s + 1, (w - k) // s + 1
      self._shapes.append((h, w, cin, k, s, ch, oh, ow))
      h, w, cin = oh, ow, ch
    self._flat_dim = h * w * cin
    spec += [(prefix + 'fc/kernel', (self._flat_dim, fc), 'glorot'), (prefix + 'fc/bias', (fc,), 'zeros')]
    return spec

  def frames_buffer(self, T1, B):
    """uint8 [3+T1, B, H*W] trajectory buffer; rows 3.. are the unroll's frames.  A data
    pipeline can write observations straight into `frames_buffer(T1,B)[3:]` (time-major)
    and pass that view as env_outputs.observation: no copy is made then."""
    slot = getattr(self, 'frames_slot', 0)      # double buffering: a copy stream fills slot 1 - s while slot s is read
    return self._buf('frames_ext' if not slot else 'frames_ext%d' % slot, (T1 + 3, B, self._obs[0] * self._obs[1]),
                     torch.uint8, zero=True)

  def _torso_fwd(self, obs, done_u8, frame_state, out, ld_out, need_state=True, for_backward=True):
    """obs uint8 [T1,B,H,W,1]; writes relu(fc) into out[:, :fc] (row stride ld_out).  Returns the new
    frame-stacking state and the context the backward needs (for_backward=False: a forward nobody differentiates --
    central inference -- skips what only the backward reads)."""
    T1, B = done_u8.shape[0], done_u8.shape[1]
    H, W = self._obs
    HW, N = H * W, T1 * B
    fl, tp = self.flat, self._tp
    ext = self.frames_buffer(T1, B)
    fr = obs.reshape(T1, B, HW)
    if fr.dtype != torch.uint8:
      raise ValueError('observations must be uint8 frames')
    if fr.data_ptr() != ext[3:].data_ptr():
      ext[3:].copy_(fr)
    nvalid = self._buf('nvalid', (T1, B), torch.uint8)
    indexed = isinstance(frame_state, IndexedFrameState)
    if indexed:                                   # central inference: the state lives in a per-env table, used in place
      ops.stack_prepare_indexed(frame_state.table, frame_state.rows, frame_state.zero_mask, done_u8, T1, B, HW, ext,
                                nvalid)
    else:
      ops.stack_prepare(frame_state.contiguous(), done_u8, T1, B, HW, ext, nvalid)
    acts, geoms = [], []
    ih, iw, cin, k, s, ch, oh, ow = self._shapes[0]
    g0 = ops.StackConvGeom(T1, B, ih, iw, oh, ow, k, k, s, ch, ch)
    a = self._buf('act0', (N, oh, ow, ch))
    # The ReLU mask of the first conv as bytes (one per four channels), where both it and the second conv's data
    # gradient have the kernel for it (the shallow torso): that gradient then reads 1 byte where it read 16 of act0
    # (17 MB instead of 275 MB at cfg2).  r3 measured a net loss against the fp32-MFMA data gradient (not byte-bound);
    # r5: wdx.h IS byte-bound (0.82 of its byte floor, 275 of its 663 MB the mask).  SEEDHIP_RELU_BITS=0: fp32 mask.
    bits0 = None
    if _RELU_BITS and for_backward and len(self._shapes) > 1 and ops.conv2d_stack_fwd_bits_supported(g0):
      ih1, iw1, cin1, k1, s1, ch1, _, _ = self._shapes[1]
      if ops.conv2d_bwd_data_bits_supported(ops.conv_geom(N, ih1, iw1, cin1, k1, k1, s1, 'valid', ch1)):
        bits0 = self._buf('act0_bits', (N, oh, ow, ch // 4), torch.uint8)
    ops.conv2d_stack_fwd(g0, ext, nvalid, fl.p(tp + 'conv0/kernel'), fl.p(tp + 'conv0/bias'), a, out_relu=True,
                         relu_bits=bits0)
    acts.append(a); geoms.append(g0)
    bits_last = None
    for i in range(1, len(self._shapes)):
      ih, iw, cin, k, s, ch, oh, ow = self._shapes[i]
      g = ops.conv_geom(N, ih, iw, cin, k, k, s, 'valid', ch)
      a2 = self._buf('act%d' % i, (N, oh, ow, ch))
      bits_i = None
      if (_RELU_BITS and for_backward and i == len(self._shapes) - 1 and ops.conv2d_fwd_bits_supported(g) and
          ops.conv2d_bwd_data_bits_supported(ops.dense_geom(N, self._flat_dim, self._fc, ld_out=ld_out))):
        bits_i = bits_last = self._buf('act%d_bits' % i, (N, oh, ow, ch // 4), torch.uint8)   # for the Dense layer's data gradient
      ops.conv2d_fwd(g, a, fl.p('%sconv%d/kernel' % (tp, i)), fl.p('%sconv%d/bias' % (tp, i)), a2, out_relu=True, relu_bits=bits_i)
      acts.append(a2); geoms.append(g); a = a2
    gfc = ops.dense_geom(N, self._flat_dim, self._fc, ld_out=ld_out)
    ops.conv2d_fwd(gfc, a, fl.p(tp + 'fc/kernel'), fl.p(tp + 'fc/bias'), out, out_relu=True)
    new_fs = None
    if need_state and indexed:
      ops.stack_pack_state_indexed(ext, nvalid, T1, B, HW, frame_state.table, frame_state.rows, frame_state.valid_mask)
      new_fs = frame_state                        # updated in place
    elif need_state:                              # the learner discards it (agents/vtrace/learner.py:75-79 `learner_outputs, _ =`)
      new_fs = torch.empty_like(frame_state)
      ops.stack_pack_state(ext, nvalid, T1, B, HW, new_fs)
    return new_fs, dict(ext=ext, nvalid=nvalid, acts=acts, geoms=geoms, gfc=gfc, bits0=bits0, bits_last=bits_last)

  # -- central inference in six launches (csrc/servestep.hip; inference.FusedInferenceState) ---------------------- #
  def serve_torso_supported(self):
    ih, iw, _, k, s, ch, oh, ow = self._shapes[0]
    return ops.conv2d_stack_fwd_rows_supported(ops.StackConvGeom(1, 1, ih, iw, oh, ow, k, k, s, ch, ch))

  def conv0_split_buffer(self):
    """The first conv's W / 255 as bf16 planes (written by serve_begin every call: the weights train)."""
    ch = self._shapes[0][5]
    return self._buf('conv0_split', (ops.serve_conv0_split_bytes(ch) // 4,), torch.int32)

  def _torso_fwd_rows(self, n, obs, store_obs, hist_rows, nvalid):
    """One inference step of n envs whose frame stacks live in the unroll store (store_obs u8 [rows, H*W]; history rows
    hist_rows [n, 4], nvalid [n]) under the request frames obs u8 [n, H*W].
    Returns the Dense layer's split-K partial sums [slices][n][fc] and the slice count: bias + ReLU are applied by the
    consumer (serve_finish)."""
    fl, tp = self.flat, self._tp
    ih, iw, _, k, s, ch, oh, ow = self._shapes[0]
    g0 = ops.StackConvGeom(1, n, ih, iw, oh, ow, k, k, s, ch, ch)
    a = self._buf('srv_act0', (n, oh, ow, ch))
    ops.conv2d_stack_fwd_rows(g0, obs, store_obs, hist_rows, nvalid, self.conv0_split_buffer(),
                              fl.p(tp + 'conv0/bias'), a, out_relu=True)
    for i in range(1, len(self._shapes)):
      ih, iw, cin, k, s, ch, oh, ow = self._shapes[i]
      g = ops.conv_geom(n, ih, iw, cin, k, k, s, 'valid', ch)
      a2 = self._buf('srv_act%d' % i, (n, oh, ow, ch))
      ops.conv2d_fwd(g, a, fl.p('%sconv%d/kernel' % (tp, i)), fl.p('%sconv%d/bias' % (tp, i)), a2, out_relu=True)
      a = a2
    gfc = ops.dense_geom(n, self._flat_dim, self._fc)
    ws = self._buf('srv_fc_partial', (ops.dense_fwd_partial_workspace_bytes(gfc) // 4 + 4,))
    return ws, ops.dense_fwd_partial(gfc, a, fl.p(tp + 'fc/kernel'), ws)

  def _torso_bwd(self, ctx, dz, wsb):
    """dz: gradient wrt the Dense pre-activation (already masked by its ReLU), row stride = gfc.ld_out."""
    fl, tp = self.flat, self._tp
    acts, geoms, gfc = ctx['acts'], ctx['geoms'], ctx['gfc']
    a_last = acts[-1]
    ops.conv2d_bwd_weight(gfc, a_last, dz, fl.g(tp + 'fc/kernel'), fl.g(tp + 'fc/bias'), wsb)
    # the torso is created first and back-propagated last: everything from its Dense layer to the end of the flat
    # buffer is final now -- its exchange overlaps the conv backward
    self._grads_ready_from(tp + 'fc/kernel')
    da = self._buf('d_act%d' % (len(acts) - 1), tuple(a_last.shape))
    if ctx.get('bits_last') is not None:
      ops.conv2d_bwd_data(gfc, dz, fl.p(tp + 'fc/kernel'), da, relu_bits=ctx['bits_last'])
    else:
      ops.conv2d_bwd_data(gfc, dz, fl.p(tp + 'fc/kernel'), da, relu_mask=a_last)
    for i in range(len(acts) - 1, 0, -1):
      g, a_in = geoms[i], acts[i - 1]
      ops.conv2d_bwd_weight(g, a_in, da, fl.g('%sconv%d/kernel' % (tp, i)), fl.g('%sconv%d/bias' % (tp, i)), wsb)
      d_in = self._buf('d_act%d' % (i - 1), tuple(a_in.shape))
      if i == 1 and ctx.get('bits0') is not None:
        ops.conv2d_bwd_data(g, da, fl.p('%sconv%d/kernel' % (tp, i)), d_in, relu_bits=ctx['bits0'])
      else:
        ops.conv2d_bwd_data(g, da, fl.p('%sconv%d/kernel' % (tp, i)), d_in, relu_mask=a_in)
      da = d_in
    # first conv: weight gradient straight from the uint8 frames
    ws0 = self._buf('stack_ws', (ops.conv2d_stack_bwd_weight_workspace_bytes(geoms[0]) // 4 + 4,))
    ops.conv2d_stack_bwd_weight(geoms[0], ctx['ext'], ctx['nvalid'], da, fl.g(tp + 'conv0/kernel'),
                                fl.g(tp + 'conv0/bias'), ws0)
    self._grads_ready_from(None, upto=tp + 'fc/kernel')

  def _torso_ws_bytes(self, ctx):
    need = ops.conv2d_bwd_weight_workspace_bytes(ctx['gfc'])
    for g in ctx['geoms'][1:]:
      need = max(need, ops.conv2d_bwd_weight_workspace_bytes(g))
    return need


class AtariShallow(_Agent, _AtariTorso):
  """Frame-stacked Atari policy/value agent (see module docstring, D1)."""

  def __init__(self, num_actions, observation_shape=(84, 84, 1), torso='shallow', device='cuda', seed=0,
               entropy_cost=None):
    super(AtariShallow, self).__init__(num_actions, device)
    convs = [(8, 4, 16), (4, 2, 32)] if torso == 'shallow' else [(8, 4, 32), (4, 2, 64), (3, 1, 64)]
    fc = 256 if torso == 'shallow' else 512
    self._entropy_cost = entropy_cost
    spec = self._torso_init(observation_shape, convs, fc)
    spec += [('policy_logits/kernel', (fc, num_actions), 'glorot'), ('policy_logits/bias', (num_actions,), 'zeros'),
             ('baseline/kernel', (fc, 1), 'glorot'), ('baseline/bias', (1,), 'zeros')]
    self._build_params(spec, seed)
    self._last = None

  def initial_state(self, batch_size):
    hw = self._obs[0] * self._obs[1]
    return AgentState(core_state=(), frame_stacking_state=torch.zeros((batch_size, hw), dtype=torch.int32,
                                                                      device=self.device))

  accepts_need_state = True      # need_state=False: skip re-packing the frame-stacking state the caller will not use
  accepts_sample_actions = True
  accepts_indexed_frame_state = True    # agent_state.frame_stacking_state may be an IndexedFrameState (table rows, in place)

  def serve_step_supported(self, full_length):
    """inference.FusedInferenceState may run this agent's inference step through csrc/servestep.hip (the frame stack is
    the agent's only recurrent state and is read from the unroll store's own frames)."""
    return (full_length >= 5 and self.serve_torso_supported() and self._skinny_heads(self._fc) and
            self._fc % 64 == 0 and self._fc <= 512 and self._ldh <= 32 and self._num_actions < self._ldh)

  def heads_image_buffer(self):
    """The packed heads as serve_finish's B-operand register image (written by serve_begin every call)."""
    return self._buf('heads_image', (ops.serve_heads_image_bytes(self._fc) // 4,))

  def serve_begin_weights(self):
    """serve_begin's weight arguments: (conv0 kernel, cout, planes, heads kernel, feat, ldh, heads image)."""
    fl = self.flat
    return (fl.p(self._tp + 'conv0/kernel'), self._shapes[0][5], self.conv0_split_buffer(), fl.p('heads/kernel'),
            self._fc, self._ldh, self.heads_image_buffer())

  def serve_forward(self, n, obs, store_obs, hist_rows, nvalid):
    """Torso of one inference step (see _torso_fwd_rows) + what serve_finish needs of the heads:
    (fc_partial, slices, fc_bias, fc, heads_image, heads_b, ldh, num_actions)."""
    ws, slices = self._torso_fwd_rows(n, obs, store_obs, hist_rows, nvalid)
    fl = self.flat
    return (ws, slices, fl.p(self._tp + 'fc/bias'), self._fc, self.heads_image_buffer(), fl.p('heads/bias'), self._ldh,
            self._num_actions)

  def __call__(self, prev_actions, env_outputs, agent_state, unroll=False, is_training=False, need_state=True,
               sample_actions=True):
    """sample_actions=False (central inference): the caller samples from head_buffers() itself (fused into the
    inference bookkeeping kernel); the returned action is None."""
    del prev_actions   # the feed-forward agent does not consume it
    obs, done = env_outputs.observation, env_outputs.done
    if not unroll:
      obs, done = obs[None], done[None]
    T1, B = done.shape[0], done.shape[1]
    N = T1 * B
    done_u8 = ops.as_u8(done)
    hfc = self._buf('fc_out', (N, self._fc))
    new_fs, ctx = self._torso_fwd(obs, done_u8, agent_state.frame_stacking_state, hfc, self._fc, need_state,
                                  for_backward=unroll)
    head = self._head_fwd(hfc, N, self._fc)
    self._last = dict(T1=T1, B=B, N=N, ctx=ctx, hfc=hfc, head=head)
    out = self._agent_output(head, T1, B, sample=sample_actions and not is_training)
    if not unroll:
      out = AgentOutput(*[None if t is None else t[0] for t in out])
    return out, AgentState(core_state=(), frame_stacking_state=new_fs)

  def backward(self):
    """Gradients of the loss wrt all parameters, given d_head (written by the loss kernel)."""
    L = self._last
    N = L['N']
    fl = self.flat
    d_head = self._buf('d_head', (N, self._ldh), zero=True)
    gh = ops.dense_geom(N, self._fc, self._ldh)
    need = max(ops.conv2d_bwd_weight_workspace_bytes(gh), self._torso_ws_bytes(L['ctx']))
    wsb = self._buf('wgrad_ws', (need // 4 + 4,))
    dz = self._buf('d_fc', (N, self._fc))
    self._head_bwd(L['hfc'], self._fc, d_head, N, dz, True, wsb)
    self._torso_bwd(L['ctx'], dz, wsb)


R2D2AgentOutput = collections.namedtuple('R2D2AgentOutput', 'action q_values')


class DuelingLSTMDQNNet(_Agent, _AtariTorso):
  """R2D2 recurrent dueling Q-network: mirror of /root/reference/atari/networks.py:221-340
  (conv 8x8/4x32 -> 4x4/2x64 -> 3x3/1x64 -> Dense 512; LSTM(512) on [features, reward, one_hot(prev_action)];
  value 512->512->1 and advantage 512->512->A (no bias) heads; q = v + a - mean(a)).
  Signature follows the reference: __call__((prev_actions, env_outputs), agent_state, unroll=False)."""

  def __init__(self, num_actions, observation_shape=(84, 84, 1), stack_size=4, device='cuda', seed=0):
    super(DuelingLSTMDQNNet, self).__init__(num_actions, device)
    if stack_size != 4:
      raise ValueError('the fused first conv implements stack_size=4 (atari/r2d2_main.py:36)')
    H = self._H = 512
    spec = self._torso_init(observation_shape, [(8, 4, 32), (4, 2, 64), (3, 1, 64)], 512, prefix='body/')
    self._in_dim = 512 + 1 + num_actions
    self._ldx = _round4(self._in_dim)
    self._ldq = _round4(num_actions + 1)
    spec += [('value/hidden/kernel', (H, 512), 'glorot'), ('value/hidden/bias', (512,), 'zeros'),
             ('value/head/kernel', (512, 1), 'glorot'), ('value/head/bias', (1,), 'zeros'),
             ('advantage/hidden/kernel', (H, 512), 'glorot'), ('advantage/hidden/bias', (512,), 'zeros'),
             ('advantage/head/kernel', (512, num_actions), 'glorot'),
             ('core/kernel', (self._in_dim, 4 * H), 'glorot'), ('core/recurrent_kernel', (H, 4 * H), 'orthogonal'),
             ('core/bias', (4 * H,), 'lstm_bias')]
    self._build_params(spec, seed)
    self._last = None

  def initial_state(self, batch_size):
    hw = self._obs[0] * self._obs[1]
    z = torch.zeros((batch_size, self._H), dtype=torch.float32, device=self.device)
    return AgentState(core_state=(z, z.clone()),
                      frame_stacking_state=torch.zeros((batch_size, hw), dtype=torch.int32, device=self.device))

  def __call__(self, input_, agent_state, unroll=False):
    prev_actions, env_outputs = input_
    reward, done, obs = env_outputs.reward, env_outputs.done, env_outputs.observation
    if not unroll:
      reward, done, obs, prev_actions = reward[None], done[None], obs[None], prev_actions[None]
    T1, B = done.shape[0], done.shape[1]
    N, A, H = T1 * B, self._num_actions, self._H
    fl = self.flat
    done_u8 = ops.as_u8(done)
    ldx = self._ldx
    X = self._buf('lstm_x', (N, ldx))
    new_fs, ctx = self._torso_fwd(obs, done_u8, agent_state.frame_stacking_state, X, ldx)
    ops.lstm_assemble_inputs(X, ldx, 512, A, reward.to(torch.float32).contiguous(), prev_actions.contiguous(),
                             False, N)                                               # networks.py:263-271 (raw reward)
    Hout, core_state = self._lstm_fwd(X, ldx, self._in_dim, H, T1, B, done_u8, agent_state.core_state, x_pad_zero=True)
    hid = self._buf('hid', (N, 1024))
    hf = hid.view(-1)
    ghid = ops.dense_geom(N, H, 512, ld_out=1024)
    ops.conv2d_fwd(ghid, Hout, fl.p('value/hidden/kernel'), fl.p('value/hidden/bias'), hf, out_relu=True)
    ops.conv2d_fwd(ghid, Hout, fl.p('advantage/hidden/kernel'), fl.p('advantage/hidden/bias'), hf[512:], out_relu=True)
    ldq = self._ldq
    va = self._buf('va', (N, ldq), zero=True)
    vf = va.view(-1)
    gadv = ops.dense_geom(N, 512, A, ld_in=1024, ld_out=ldq)
    gval = ops.dense_geom(N, 512, 1, ld_in=1024, ld_out=ldq)
    ops.conv2d_fwd(gadv, hf[512:], fl.p('advantage/head/kernel'), None, vf)
    ops.conv2d_fwd(gval, hf, fl.p('value/head/kernel'), fl.p('value/head/bias'), vf[A:])
    q = self._buf('q', (N, A))
    action = self._buf('q_action', (N,), torch.int32)
    ops.dueling_fwd(va, ldq, N, A, q, action)
    self._last = dict(T1=T1, B=B, N=N, ctx=ctx, Hout=Hout, hid=hid, ghid=ghid, gadv=gadv, gval=gval)
    out = R2D2AgentOutput(action.view(T1, B), q.view(T1, B, A))
    if not unroll:
      out = R2D2AgentOutput(out.action[0], out.q_values[0])
    return out, AgentState(core_state=core_state, frame_stacking_state=new_fs)

  def backward(self, dq=None):
    """dq [T,B,A]: gradient of the loss wrt q_values (default: the one the fused R2D2 loss kernel left in
    self._last['dq']); fills the flat gradient buffer."""
    L = self._last
    if dq is None:
      dq = L['dq']
    N, A, H = L['N'], self._num_actions, self._H
    fl = self.flat
    need = max(ops.conv2d_bwd_weight_workspace_bytes(L['ghid']), self._lstm_ws_bytes(), self._torso_ws_bytes(L['ctx']),
               ops.conv2d_bwd_weight_workspace_bytes(L['gadv']), ops.conv2d_bwd_weight_workspace_bytes(L['gval']))
    wsb = self._buf('wgrad_ws', (need // 4 + 4,))
    ldq = self._ldq
    d_va = self._buf('d_va', (N, ldq))
    ops.dueling_bwd(dq.contiguous().view(N, A), N, A, d_va, ldq)
    dvf = d_va.view(-1)
    hf = L['hid'].view(-1)
    d_hid = self._buf('d_hid', (N, 1024))
    dhf = d_hid.view(-1)
    ops.conv2d_bwd_weight(L['gadv'], hf[512:], dvf, fl.g('advantage/head/kernel'), None, wsb)
    ops.conv2d_bwd_data(L['gadv'], dvf, fl.p('advantage/head/kernel'), dhf[512:], relu_mask=hf[512:])
    ops.conv2d_bwd_weight(L['gval'], hf, dvf[A:], fl.g('value/head/kernel'), fl.g('value/head/bias'), wsb)
    ops.conv2d_bwd_data(L['gval'], dvf[A:], fl.p('value/head/kernel'), dhf, relu_mask=hf)
    ghid = L['ghid']
    ops.conv2d_bwd_weight(ghid, L['Hout'], dhf, fl.g('value/hidden/kernel'), fl.g('value/hidden/bias'), wsb)
    ops.conv2d_bwd_weight(ghid, L['Hout'], dhf[512:], fl.g('advantage/hidden/kernel'), fl.g('advantage/hidden/bias'), wsb)
    dHout = self._buf('d_hout', (N, H))
    ops.conv2d_bwd_data(ghid, dhf, fl.p('value/hidden/kernel'), dHout)
    ops.conv2d_bwd_data(ghid, dhf[512:], fl.p('advantage/hidden/kernel'), dHout, add=dHout)
    dX = self._lstm_bwd(dHout, wsb)
    self._torso_bwd(L['ctx'], dX, wsb)


class ImpalaDeep(_Agent):
  """IMPALA deep ResNet + LSTM(256) agent: mirror of /root/reference/dmlab/networks.py:63-171
  (3 stacks of Conv3x3 -> MaxPool 3/2 -> 2 residual blocks, channels 16/32/32; Dense 256; LSTM 256 on
  [features, clip(reward), one_hot(prev_action)]; policy / baseline heads).  39 trainable tensors in the
  reference's creation order (tests/agents_test.py:45)."""

  def __init__(self, num_actions, observation_shape=(72, 96, 3), device='cuda', seed=0, entropy_cost=None,
               channels=(16, 32, 32), fc=256, lstm=256, kernel_init='glorot', packed_bits=False):
    """lstm=0: feed-forward (no core, no reward / action inputs); kernel_init: initialiser of every kernel
    ('glorot' Keras default | 'lecun_normal'); packed_bits: observations are uint16 / int16 bit planes [h, w, planes],
    unpacked to 16 binary channels each (GFootball, football/networks.py:68-150)."""
    super(ImpalaDeep, self).__init__(num_actions, device)
    h, w, c = observation_shape
    self._packed = bool(packed_bits)
    self._obs_in = (h, w, c)
    if packed_bits:
      c = c * 16
    self._obs = (h, w, c)
    self._channels, self._fc, self._H = tuple(channels), fc, lstm
    self._entropy_cost = entropy_cost
    # csrc/convpool.hip is built for the reference's first stage: 3-channel uint8 frames -> 16 channels
    self._fused_first_stage = (c == 3 and self._channels[0] == 16 and w <= 114 and h >= 3 and w >= 3)
    spec, cin = [], c
    ki = kernel_init
    self._stack_shapes = []                    # (ih, iw, cin, ch, oh, ow)
    for i, ch in enumerate(self._channels):
      spec += [('stack%d/conv/kernel' % i, (3, 3, cin, ch), ki), ('stack%d/conv/bias' % i, (ch,), 'zeros')]
      for b in range(2):
        for j in range(2):
          spec += [('stack%d/res_%d/conv2d_%d/kernel' % (i, b, j), (3, 3, ch, ch), ki),
                   ('stack%d/res_%d/conv2d_%d/bias' % (i, b, j), (ch,), 'zeros')]
      oh, ow = (h + 1) // 2, (w + 1) // 2
      self._stack_shapes.append((h, w, cin, ch, oh, ow))
      h, w, cin = oh, ow, ch
    self._flat_dim = h * w * cin
    H = lstm
    self._in_dim = fc + 1 + num_actions if H else fc
    self._ldx = _round4(self._in_dim)
    spec += [('conv_to_linear/kernel', (self._flat_dim, fc), ki), ('conv_to_linear/bias', (fc,), 'zeros')]
    if H:
      spec += [('core/kernel', (self._in_dim, 4 * H), 'glorot'), ('core/recurrent_kernel', (H, 4 * H), 'orthogonal'),
               ('core/bias', (4 * H,), 'lstm_bias')]
    feat = H or fc
    spec += [('policy_logits/kernel', (feat, num_actions), ki), ('policy_logits/bias', (num_actions,), 'zeros'),
             ('baseline/kernel', (feat, 1), ki), ('baseline/bias', (1,), 'zeros')]
    self._build_params(spec, seed)
    self._last = None

  def initial_state(self, batch_size):
    if not self._H:
      return ()
    z = torch.zeros((batch_size, self._H), dtype=torch.float32, device=self.device)
    return (z, z.clone())

  accepts_sample_actions = True

  def __call__(self, prev_actions, env_outputs, core_state, unroll=False, is_training=False, sample_actions=True):
    reward, done, obs = env_outputs.reward, env_outputs.done, env_outputs.observation
    if not unroll:
      reward, done, obs, prev_actions = reward[None], done[None], obs[None], prev_actions[None]
    T1, B = done.shape[0], done.shape[1]
    N = T1 * B
    fl = self.flat
    h0, w0, c0 = self._obs
    if self._packed:
      if obs.dtype not in (torch.int16, torch.uint16):
        raise ValueError('packed observations must be 16-bit words (football/observation.py)')
      x = self._buf('unpacked', (N, h0, w0, c0), torch.uint8)
      ops.unpackbits_u16(obs.contiguous(), x)                                          # football/networks.py:101-104
    else:
      if obs.dtype != torch.uint8:
        raise ValueError('observations must be uint8 frames')
      x = obs.reshape(N, h0, w0, c0).contiguous()
    saved = []
    for i, (ih, iw, cin, ch, oh, ow) in enumerate(self._stack_shapes):
      g = ops.conv_geom(N, ih, iw, cin, 3, 3, 1, 'same', ch)
      gres = ops.conv_geom(N, oh, ow, ch, 3, 3, 1, 'same', ch)
      # ReLU masks as bytes (one per four channels): every tensor a residual-block layer reads through its ReLU gets
      # its sign written by the kernel that produces it (pool / conv epilogue), and that layer's data gradient reads
      # those bytes instead of the fp32 activation (1/16 of the bytes).  SEEDHIP_RELU_BITS=0: fp32 masks.
      bits = _RELU_BITS and unroll and ops.conv2d_fwd_outbits_supported(gres)
      mbuf = lambda name: self._buf('s%d_%s' % (i, name), (N, oh, ow, ch // 4), torch.uint8) if bits else None
      p = self._buf('s%d_p' % i, (N, oh, ow, ch))
      arg = self._buf('s%d_arg' % i, (N, oh, ow, ch), torch.uint8)
      mp = mbuf('p_m')
      fused = i == 0 and self._fused_first_stage
      if fused:
        # conv + max-pool of the uint8 stage in one kernel: the [N, 72, 96, 16] pre-pool tensor is never written
        ops.conv3x3_u8_pool_fwd(x, fl.p('stack0/conv/kernel'), fl.p('stack0/conv/bias'), p, arg, pooled_bits=mp)
      else:
        a = self._buf('s%d_a' % i, (N, ih, iw, ch))
        ops.conv2d_fwd(g, x, fl.p('stack%d/conv/kernel' % i), fl.p('stack%d/conv/bias' % i), a,
                       in_dtype=ops.IN_U8_DIV255 if i == 0 else ops.IN_F32)        # dmlab/networks.py:98-100
        ops.maxpool_fwd(a, p, arg, y_bits=mp)
      blocks = []
      for b in range(2):                                                             # dmlab/networks.py:52-59
        r1 = self._buf('s%d_b%d_r1' % (i, b), (N, oh, ow, ch))
        m1 = mbuf('b%d_m1' % b)                                                      # sign of r1
        ops.conv2d_fwd(gres, p, fl.p('stack%d/res_%d/conv2d_0/kernel' % (i, b)),
                       fl.p('stack%d/res_%d/conv2d_0/bias' % (i, b)), r1, in_relu=True, out_bits=m1)
        r2 = self._buf('s%d_b%d_r2' % (i, b), (N, oh, ow, ch))
        m2 = mbuf('b%d_m2' % b) if b == 0 else None                                  # sign of r2 = the next block's input
        ops.conv2d_fwd(gres, r1, fl.p('stack%d/res_%d/conv2d_1/kernel' % (i, b)),
                       fl.p('stack%d/res_%d/conv2d_1/bias' % (i, b)), r2, in_relu=True, residual=p, out_bits=m2)
        blocks.append((p, r1, mp, m1))
        p, mp = r2, m2
      saved.append(dict(g=g, gres=gres, x=x, arg=arg, a_shape=(N, ih, iw, ch), blocks=blocks, fused=fused))
      x = p
    flat = x.view(N, self._flat_dim)
    ldx = self._ldx
    X = self._buf('lstm_x', (N, ldx))
    gfc = ops.dense_geom(N, self._flat_dim, self._fc, ld_out=ldx)
    ops.conv2d_fwd(gfc, flat, fl.p('conv_to_linear/kernel'), fl.p('conv_to_linear/bias'), X, in_relu=True,
                   out_relu=True)                                                    # :105-109
    if self._H:
      ops.lstm_assemble_inputs(X, ldx, self._fc, self._num_actions, reward.to(torch.float32).contiguous(),
                               prev_actions.contiguous(), True, N)                   # :112-114
      done_u8 = ops.as_u8(done)
      Hout, new_state = self._lstm_fwd(X, ldx, self._in_dim, self._H, T1, B, done_u8, core_state, x_pad_zero=True)
      head = self._head_fwd(Hout, N, self._H)
    else:                                                                            # football/networks.py:147-150
      Hout, new_state = None, ()
      head = self._head_fwd(X, N, self._fc)
    self._last = dict(T1=T1, B=B, N=N, saved=saved, flat=flat, gfc=gfc, X=X, Hout=Hout, head=head)
    out = self._agent_output(head, T1, B, sample=sample_actions and not is_training)
    if not unroll:
      out = AgentOutput(*[None if t is None else t[0] for t in out])
    return out, new_state

  def backward(self):
    L = self._last
    N = L['N']
    fl = self.flat
    d_head = self._buf('d_head', (N, self._ldh), zero=True)
    wsb = self._wgrad_ws()
    H = self._H
    if H:
      gh = ops.dense_geom(N, H, self._ldh)
      dHout = self._buf('d_hout', (N, H))
      self._head_bwd(L['Hout'], H, d_head, N, dHout, False, wsb)
      dX = self._lstm_bwd(dHout, wsb)
    else:
      gh = ops.dense_geom(N, self._fc, self._ldh)
      ops.conv2d_bwd_weight(gh, L['X'], d_head, fl.g('heads/kernel'), fl.g('heads/bias'), wsb)
      dX = self._buf('d_fc', (N, self._ldx))
      ops.conv2d_bwd_data(gh, d_head, fl.p('heads/kernel'), dX, relu_mask=L['X'])
    # Dense 256 (its ReLU mask was applied to dX by the LSTM input-projection / head dgrad)
    ops.conv2d_bwd_weight(L['gfc'], L['flat'], dX, fl.g('conv_to_linear/kernel'), fl.g('conv_to_linear/bias'), wsb,
                          in_relu=True)
    self._grads_ready_from('conv_to_linear/kernel')        # Dense + LSTM + heads: exchanged under the conv backward
    d_flat = self._buf('d_flat', tuple(L['flat'].shape))
    ops.conv2d_bwd_data(L['gfc'], dX, fl.p('conv_to_linear/kernel'), d_flat, relu_mask=L['flat'])
    dp = d_flat
    for i in range(len(L['saved']) - 1, -1, -1):
      S = L['saved'][i]
      gres = S['gres']
      for b in (1, 0):
        p_in, r1, m0, m1 = S['blocks'][b]
        k0, k1 = 'stack%d/res_%d/conv2d_0' % (i, b), 'stack%d/res_%d/conv2d_1' % (i, b)
        ops.conv2d_bwd_weight(gres, r1, dp, fl.g(k1 + '/kernel'), fl.g(k1 + '/bias'), wsb, in_relu=True)
        d_r1 = self._buf('d_s%d_b%d_r1' % (i, b), tuple(r1.shape))
        if m1 is not None:
          ops.conv2d_bwd_data(gres, dp, fl.p(k1 + '/kernel'), d_r1, relu_bits=m1)
        else:
          ops.conv2d_bwd_data(gres, dp, fl.p(k1 + '/kernel'), d_r1, relu_mask=r1)
        ops.conv2d_bwd_weight(gres, p_in, d_r1, fl.g(k0 + '/kernel'), fl.g(k0 + '/bias'), wsb, in_relu=True)
        d_pin = self._buf('d_s%d_b%d_p' % (i, b), tuple(p_in.shape))
        if m0 is not None:
          ops.conv2d_bwd_data(gres, d_r1, fl.p(k0 + '/kernel'), d_pin, relu_bits=m0, add=dp)  # + skip path
        else:
          ops.conv2d_bwd_data(gres, d_r1, fl.p(k0 + '/kernel'), d_pin, relu_mask=p_in, add=dp)
        dp = d_pin
      kc = 'stack%d/conv' % i
      if S['fused']:
        n0, ih0, iw0, _ = S['a_shape']
        ws0 = self._buf('convpool_ws', (ops.conv3x3_u8_pool_bwd_workspace_bytes(n0, ih0, iw0) // 4 + 4,))
        ops.conv3x3_u8_pool_bwd(S['x'], dp, S['arg'], fl.g(kc + '/kernel'), fl.g(kc + '/bias'), ws0)
        continue
      d_a = self._buf('d_s%d_a' % i, S['a_shape'])
      if i > 0 and os.environ.get('SEEDHIP_POOL_DGRAD', '1') != '0' and ops.conv2d_bwd_data_pool_supported(S['g']):
        # the max-pool backward inside the data gradient's loader (fgx.h): d_a is written once for the weight gradient
        # and never read back by this layer's data gradient
        dx = self._buf('d_s%d_x' % i, tuple(S['x'].shape))
        ops.conv2d_bwd_data_pool(S['g'], dp, S['arg'], fl.p(kc + '/kernel'), dx, d_a)
        ops.conv2d_bwd_weight(S['g'], S['x'], d_a, fl.g(kc + '/kernel'), fl.g(kc + '/bias'), wsb)
        dp = dx
        continue
      ops.maxpool_bwd(dp, S['arg'], d_a)
      ops.conv2d_bwd_weight(S['g'], S['x'], d_a, fl.g(kc + '/kernel'), fl.g(kc + '/bias'), wsb,
                            in_dtype=ops.IN_U8_DIV255 if i == 0 else ops.IN_F32)
      if i > 0:
        dx = self._buf('d_s%d_x' % i, tuple(S['x'].shape))
        ops.conv2d_bwd_data(S['g'], d_a, fl.p(kc + '/kernel'), dx)
        dp = dx
    self._grads_ready_from(None, upto='conv_to_linear/kernel')

  def _wgrad_ws(self):
    L = self._last
    need = ops.conv2d_bwd_weight_workspace_bytes(ops.dense_geom(L['N'], self._H or self._fc, self._ldh))
    need = max(need, ops.conv2d_bwd_weight_workspace_bytes(L['gfc']))
    if self._H:
      need = max(need, self._lstm_ws_bytes())
    for S in L['saved']:
      need = max(need, ops.conv2d_bwd_weight_workspace_bytes(S['g']), ops.conv2d_bwd_weight_workspace_bytes(S['gres']))
    return self._buf('wgrad_ws', (need // 4 + 4,))


class GFootball(ImpalaDeep):
  """Google Research Football agent: mirror of /root/reference/football/networks.py:68-150 -- the ImpalaDeep torso with
  FOUR stacks (16, 32, 32, 32 channels), lecun_normal kernels, no LSTM and no reward / action inputs; observations are
  the packed SMM bit planes of football/observation.py (uint16 words, 16 binary channels each), unpacked on the
  device (csrc/frames.hip: unpackbits) and read by the first conv as bytes with its / 255 fused."""

  def __init__(self, num_actions, observation_shape=(72, 96, 1), device='cuda', seed=0, entropy_cost=None):
    super(GFootball, self).__init__(num_actions, observation_shape=observation_shape, device=device, seed=seed,
                                    entropy_cost=entropy_cost, channels=(16, 32, 32, 32), fc=256, lstm=0,
                                    kernel_init='lecun_normal', packed_bits=True)


class MLPandLSTM(_Agent):
  """MLP + stacked-LSTM agent: mirror of /root/reference/agents/vtrace/networks.py:25-121 (the agent of the MuJoCo /
  generic V-trace mains): Dense(size, relu) layers on the flat float observation, tf.keras StackedRNNCells of LSTMCells
  with the done-reset before every step, policy / baseline heads.  The recurrences run layer by layer over the whole
  unroll (a layer only needs its own previous state and the layer below at the same step), each on the whole-sequence
  LSTM kernels; state = tuple of (h, c) per cell (StackedRNNCells.get_initial_state)."""

  def __init__(self, num_actions, observation_size, mlp_sizes=(64, 64), lstm_sizes=(64,), device='cuda', seed=0,
               entropy_cost=None):
    super(MLPandLSTM, self).__init__(num_actions, device)
    if not lstm_sizes:
      raise ValueError('MLPandLSTM needs at least one LSTM layer')
    if any(m % 4 for m in mlp_sizes) or any(h % 4 for h in lstm_sizes):
      raise ValueError('layer sizes must be multiples of 4 (16-byte rows)')
    self._obs_dim, self._mlp, self._lstm = observation_size, tuple(mlp_sizes), tuple(lstm_sizes)
    self._entropy_cost = entropy_cost
    spec, cin = [], observation_size
    for i, m in enumerate(self._mlp):
      spec += [('mlp/dense_%d/kernel' % i, (cin, m), 'glorot'), ('mlp/dense_%d/bias' % i, (m,), 'zeros')]
      cin = m
    for l, h in enumerate(self._lstm):
      spec += [('core/cell_%d/kernel' % l, (cin, 4 * h), 'glorot'), ('core/cell_%d/recurrent_kernel' % l, (h, 4 * h), 'orthogonal'),
               ('core/cell_%d/bias' % l, (4 * h,), 'lstm_bias')]
      cin = h
    spec += [('policy_logits/kernel', (cin, num_actions), 'glorot'), ('policy_logits/bias', (num_actions,), 'zeros'),
             ('baseline/kernel', (cin, 1), 'glorot'), ('baseline/bias', (1,), 'zeros')]
    self._build_params(spec, seed)
    self._last = None

  accepts_sample_actions = True

  def initial_state(self, batch_size):
    z = lambda h: torch.zeros((batch_size, h), dtype=torch.float32, device=self.device)
    return tuple((z(h), z(h)) for h in self._lstm)

  def __call__(self, prev_actions, env_outputs, core_state, unroll=False, is_training=False, sample_actions=True):
    del prev_actions                                 # unused by this agent (networks.py:79)
    done, obs = env_outputs.done, env_outputs.observation
    if not unroll:
      done, obs = done[None], obs[None]
    T1, B = done.shape[0], done.shape[1]
    N = T1 * B
    fl = self.flat
    x = obs.reshape(N, self._obs_dim).to(torch.float32)
    ld = _round4(self._obs_dim)
    if ld != self._obs_dim or not x.is_contiguous():
      xp = self._buf('obs_pad', (N, ld), zero=True)
      xp[:, :self._obs_dim].copy_(x)
      x = xp
    acts, geoms, cin = [x], [], self._obs_dim
    for i, m in enumerate(self._mlp):                # networks.py:104 (Dense + relu)
      g = ops.dense_geom(N, cin, m, ld_in=ld)
      a = self._buf('mlp%d' % i, (N, m))
      ops.conv2d_fwd(g, acts[-1], fl.p('mlp/dense_%d/kernel' % i), fl.p('mlp/dense_%d/bias' % i), a, out_relu=True)
      acts.append(a); geoms.append(g); cin, ld = m, m
    done_u8 = ops.as_u8(done)
    h_in, new_state = acts[-1], []
    for l, h in enumerate(self._lstm):               # networks.py:106-119, one layer at a time
      h_in, st = self._lstm_fwd(h_in, ld, cin, h, T1, B, done_u8, core_state[l], prefix='core/cell_%d' % l)
      new_state.append(st); cin, ld = h, h
    head = self._head_fwd(h_in, N, cin)
    self._last = dict(T1=T1, B=B, N=N, acts=acts, geoms=geoms, top=h_in, head=head)
    out = self._agent_output(head, T1, B, sample=sample_actions and not is_training)
    if not unroll:
      out = AgentOutput(*[None if t is None else t[0] for t in out])
    return out, tuple(new_state)

  def backward(self):
    L = self._last
    N, fl = L['N'], self.flat
    d_head = self._buf('d_head', (N, self._ldh), zero=True)
    htop = self._lstm[-1]
    gh = ops.dense_geom(N, htop, self._ldh)
    need = ops.conv2d_bwd_weight_workspace_bytes(gh)
    for l in range(len(self._lstm)):
      need = max(need, self._lstm_ws_bytes('core/cell_%d' % l))
    for g in L['geoms']:
      need = max(need, ops.conv2d_bwd_weight_workspace_bytes(g))
    wsb = self._buf('wgrad_ws', (need // 4 + 4,))
    dh = self._buf('d_top', (N, htop))
    self._head_bwd(L['top'], htop, d_head, N, dh, False, wsb)
    for l in range(len(self._lstm) - 1, -1, -1):
      # the layer's input is the layer below (no activation) or, for layer 0, the last Dense + relu output
      dh = self._lstm_bwd(dh, wsb, prefix='core/cell_%d' % l, relu_mask_x=(l == 0 and bool(self._mlp)))
    acts, geoms = L['acts'], L['geoms']
    for i in range(len(self._mlp) - 1, -1, -1):
      ops.conv2d_bwd_weight(geoms[i], acts[i], dh, fl.g('mlp/dense_%d/kernel' % i), fl.g('mlp/dense_%d/bias' % i), wsb)
      if i > 0:
        d_in = self._buf('d_mlp%d' % (i - 1), tuple(acts[i].shape))
        ops.conv2d_bwd_data(geoms[i], dh, fl.p('mlp/dense_%d/kernel' % i), d_in, relu_mask=acts[i])
        dh = d_in
    self._grads_ready_from(None)
