"""Thin torch-tensor wrappers over the C ABI (pointers + sizes only cross it)."""
import collections
import os
import contextlib
import ctypes

import torch

from seed_rl_amd import _lib
from seed_rl_amd._lib import ConvGeom, StackConvGeom

IN_F32, IN_U8_DIV255 = 0, 1


# --------------------------------------------------------------------------- #
# Optional per-kernel timing with HIP events on the launch stream (bench.py).
# --------------------------------------------------------------------------- #
class Profiler(object):
  """Records (start, end) HIP events around C-ABI calls on torch's current stream.
  `only`: restrict to a set of region names (cheap enough for the timed region)."""

  def __init__(self, only=None, serialize=False):
    """serialize=True drains the device before every region (per-kernel attribution pass: a region then
    measures its own kernel(s) only, not the tail of the previous launch); never used in a timed region."""
    self.serialize = serialize
    self.only = set(only) if only is not None else None
    self.events = collections.OrderedDict()
    self.costs = {}                                  # name -> [(flops, bytes) per call]
    self.pipes = {}

  @staticmethod
  def event_overhead_ms(n=50, serialize=False):
    """What a (start, end) HIP event pair reports around a TRIVIAL kernel (a 1-element fill): the marker /
    dispatch overhead that every instrumented region carries.  Subtracted from every region so that kernels
    launched hundreds of times per step (LSTM steps, a few microseconds each) are not ranked by it."""
    x = torch.zeros(1, device='cuda')
    pairs = []
    for _ in range(n):
      s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      if serialize:
        torch.cuda.synchronize()                     # same idle-device dispatch latency as a serialized region
      s.record(); x.fill_(1.0); e.record()
      pairs.append((s, e))
    torch.cuda.synchronize()
    v = sorted(s.elapsed_time(e) for s, e in pairs)
    return v[len(v) // 2]

  def summary(self):
    torch.cuda.synchronize()
    ovh = self.event_overhead_ms(serialize=self.serialize)
    durations = collections.OrderedDict(
        (name, [max(s.elapsed_time(e) - ovh, 0.0) for s, e in evs]) for name, evs in self.events.items())
    return aggregate(durations, self.costs, self.pipes)


def _median(ms):
  ms = sorted(ms)
  return ms[len(ms) // 2] if len(ms) % 2 else 0.5 * (ms[len(ms) // 2 - 1] + ms[len(ms) // 2])


def aggregate(durations, costs, pipes=None):
  """Per-region totals from per-CALL records (pure bookkeeping, no GPU: tests/test_bench_accounting.py).

  durations[name] = [ms of call 0, ...]; costs[name] = [(flops, bytes) of call 0, ...].  One region name can cover calls
  of different sizes (R2D2 runs each torso layer on 40 x B and on 81 x B frames): the calls are grouped by their
  (flops, bytes), each group is priced at its own MEDIAN duration (one launch that catches a clock ramp, a page fault or
  a first-use code-object load -- seen: 0.34 ms for a 15 us kernel -- must not re-rank the kernels), and the region
  reports sums over the groups: total_ms, flops_total, bytes_total, and per-call AVERAGES avg_ms / flops / bytes, so that
  flops / avg_ms == sum(flops) / sum(time).  (r4 kept the LAST call's flops and the median over ALL calls: cfg5's
  conv layers were priced 1.34x too fast.)"""
  out = collections.OrderedDict()
  for name, ms in durations.items():
    cs = costs[name]
    assert len(cs) == len(ms), name
    groups = collections.OrderedDict()
    for d, c in zip(ms, cs):
      groups.setdefault(tuple(c), []).append(d)
    glist = [dict(calls=len(v), flops=k[0], bytes=k[1], med_ms=_median(v)) for k, v in groups.items()]
    n = len(ms)
    total = sum(g['med_ms'] * g['calls'] for g in glist)
    fl, by = sum(g['flops'] * g['calls'] for g in glist), sum(g['bytes'] * g['calls'] for g in glist)
    out[name] = dict(calls=n, total_ms=total, avg_ms=total / n, mean_ms=sum(ms) / n, flops=fl / n, bytes=by / n,
                     flops_total=fl, bytes_total=by, groups=glist, pipe=(pipes or {}).get(name, 'f32'))
  return out


_PROFILER = None


def set_profiler(p):
  global _PROFILER
  _PROFILER = p


@contextlib.contextmanager
def _region(name, flops=0, nbytes=0, pipe='f32'):
  """pipe: what bounds the region's flops -- 'f32' (fp32 MFMA), 'bf16x3' / 'bf16x6' (bf16 MFMA through the exact
  operand splits: 3 / 6 bf16 MACs per algorithmic MAC); bench.py prices rooflines with it."""
  p = _PROFILER
  if p is None or (p.only is not None and name not in p.only):
    yield
    return
  s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  if p.serialize:
    torch.cuda.synchronize()
  s.record()
  yield
  e.record()
  p.events.setdefault(name, []).append((s, e))
  p.costs.setdefault(name, []).append((flops, nbytes))
  p.pipes[name] = pipe() if callable(pipe) else pipe


def _conv_name(kind, g):
  return '%s[%dx%d/%d %d->%d @%dx%d]' % (kind, g.kh, g.kw, g.stride, g.cin, g.cout, g.ih, g.iw)


def conv2d_pipe(g, which):
  """seedhip_conv2d_pipe: 1 = fp32 MFMA, 6 = bf16 MFMA through the exact three-way split; which: 0 fwd, 1 dgrad, 2 wgrad."""
  return int(_lib.lib().seedhip_conv2d_pipe(ctypes.byref(g), which))


def _conv_pipe(g, which):
  return 'bf16x6' if conv2d_pipe(g, which) == 6 else 'f32'


def _stack_pipe():
  return 'f32' if os.environ.get('SEEDHIP_STACK_BF16', '1') == '0' else 'bf16x3'


def _conv_cost(g, in_bytes_per_el=4):
  """Algorithmic flops and bytes of one conv/dense op."""
  flops = 2.0 * g.n_img * g.oh * g.ow * g.cout * g.kh * g.kw * g.cin
  nbytes = (g.n_img * g.ih * g.iw * g.cin * in_bytes_per_el + g.kh * g.kw * g.cin * g.cout * 4 +
            g.n_img * g.oh * g.ow * g.cout * 4)
  return flops, nbytes


def conv_geom(n_img, ih, iw, cin, kh, kw, stride, padding, cout, ld_in=None, ld_out=None):
  """Keras Conv2D geometry; padding in {'same','valid'}."""
  if padding == 'same':
    oh, ow = -(-ih // stride), -(-iw // stride)
    pt = max((oh - 1) * stride + kh - ih, 0) // 2
    pl = max((ow - 1) * stride + kw - iw, 0) // 2
  elif padding == 'valid':
    oh, ow, pt, pl = (ih - kh) // stride + 1, (iw - kw) // stride + 1, 0, 0
  else:
    raise ValueError(padding)
  return ConvGeom(n_img, ih, iw, cin, oh, ow, kh, kw, stride, pt, pl, cout,
                  ld_in or cin, ld_out or cout)


def dense_geom(rows, cin, cout, ld_in=None, ld_out=None):
  return conv_geom(rows, 1, 1, cin, 1, 1, 1, 'valid', cout, ld_in, ld_out)


def _dev(t):
  return torch.cuda.device(t.device)


_SPLITK_WS = {}


def _splitk_ws(nbytes, like):
  """Scratch for the split-K dense paths: one buffer per (device, stream), grown on demand (warm-up), reused by
  consecutive ops on that stream (stream order makes that safe)."""
  if nbytes == 0:
    return None, 0
  key = (like.device, torch.cuda.current_stream(like.device).cuda_stream)
  t = _SPLITK_WS.get(key)
  if t is None or t.numel() * 4 < nbytes:
    t = torch.empty(max(nbytes // 4 + 4, 1 << 20), dtype=torch.float32, device=like.device)
    _SPLITK_WS[key] = t
  return t, t.numel() * 4


def conv2d_fwd_bits_supported(g):
  """conv2d_fwd(..., relu_bits=) is served for this geometry (the next layer's data gradient then takes the byte mask)."""
  return bool(_lib.lib().seedhip_conv2d_fwd_bits_supported(ctypes.byref(g)))


def conv2d_fwd_outbits_supported(g):
  """conv2d_fwd(..., out_bits=) and conv2d_bwd_data(..., relu_bits=, add=) are served for this geometry."""
  return bool(_lib.lib().seedhip_conv2d_fwd_outbits_supported(ctypes.byref(g)))


def conv2d_fwd(g, x, w, bias, out, in_dtype=IN_F32, in_relu=False, out_relu=False, residual=None, relu_bits=None,
               out_bits=None):
  """relu_bits (uint8 [n_img * oh * ow, cout / 4], where conv2d_fwd_bits_supported; needs out_relu, no residual): also
  receives the ReLU mask of `out` as bytes -- bit r of byte q = out[pixel][4 q + r] > 0.
  out_bits (uint8 [n_img * oh * ow, cout / 4], where conv2d_fwd_outbits_supported; no out_relu, residual allowed): the
  same bytes for an output that the NEXT layer reads through its own ReLU (ImpalaDeep's residual blocks)."""
  if out_bits is not None:
    if relu_bits is not None or out_relu or in_dtype != IN_F32:
      raise ValueError('conv2d_fwd: out_bits needs fp32 input, out_relu=False and no relu_bits')
    flops, nbytes = _conv_cost(g, 4)
    nbytes += (4 * g.n_img * g.oh * g.ow * g.cout if residual is not None else 0) + g.n_img * g.oh * g.ow * g.cout // 8
    with _region(_conv_name('conv_fwd', g), flops, nbytes, pipe=lambda: _conv_pipe(g, 0)):
      with _dev(out):
        _lib.check(_lib.lib().seedhip_conv2d_fwd_outbits(
            ctypes.byref(g), _lib.ptr(x), int(in_relu), _lib.ptr(w), _lib.ptr(bias), _lib.ptr(out), _lib.ptr(residual),
            _lib.ptr(out_bits), _lib.stream()), 'seedhip_conv2d_fwd_outbits')
      return out
  if relu_bits is not None:
    if residual is not None or not out_relu:
      raise ValueError('conv2d_fwd: relu_bits needs out_relu=True and no residual')
    flops, nbytes = _conv_cost(g, 1 if in_dtype else 4)
    with _region(_conv_name('conv_fwd', g), flops, nbytes + g.n_img * g.oh * g.ow * g.cout // 8, pipe=lambda: _conv_pipe(g, 0)):
      with _dev(out):
        _lib.check(_lib.lib().seedhip_conv2d_fwd_bits(
            ctypes.byref(g), _lib.ptr(x), in_dtype, int(in_relu), _lib.ptr(w), _lib.ptr(bias), _lib.ptr(out),
            _lib.ptr(relu_bits), _lib.stream()), 'seedhip_conv2d_fwd_bits')
      return out
  with _region(_conv_name('conv_fwd', g), *_conv_cost(g, 1 if in_dtype else 4), pipe=lambda: _conv_pipe(g, 0)):
    with _dev(out):
      ws, wsb = _splitk_ws(int(_lib.lib().seedhip_conv2d_fwd_workspace_bytes(ctypes.byref(g))), out)
      _lib.check(_lib.lib().seedhip_conv2d_fwd_ws(
          ctypes.byref(g), _lib.ptr(x), in_dtype, int(in_relu), _lib.ptr(w), _lib.ptr(bias), _lib.ptr(out),
          int(out_relu), _lib.ptr(residual), _lib.ptr(ws), wsb, _lib.stream()), 'seedhip_conv2d_fwd')
    return out


def conv2d_bwd_data_bits_supported(g):
  """The data gradient of this geometry can take its ReLU mask as bytes (conv2d_stack_fwd(..., relu_bits=))."""
  return bool(_lib.lib().seedhip_conv2d_bwd_data_bits_supported(ctypes.byref(g)))


def conv2d_bwd_data(g, dy, w, dx, relu_mask=None, add=None, relu_bits=None):
  """relu_bits (uint8, one byte per four input channels) replaces relu_mask where conv2d_bwd_data_bits_supported."""
  flops, nbytes = _conv_cost(g)
  # ALGORITHMIC bytes: a ReLU mask is one BIT per element whatever tensor the kernel reads for it today (an fp32
  # activation read for its sign is avoidable traffic and must not raise the floor); the skip-path add is 4 bytes
  elems = g.n_img * g.ih * g.iw * g.cin
  nbytes += (elems // 8 if relu_mask is not None else 0) + (4 * elems if add is not None else 0)
  if relu_bits is not None:
    if relu_mask is not None:
      raise ValueError('conv2d_bwd_data: relu_bits excludes relu_mask')
    with _region(_conv_name('conv_dgrad', g), flops, nbytes + g.n_img * g.ih * g.iw * g.cin // 8, pipe=lambda: _conv_pipe(g, 1)):
      with _dev(dx):
        if add is not None:                               # (served where conv2d_fwd_outbits_supported)
          _lib.check(_lib.lib().seedhip_conv2d_bwd_data_bits_add(
              ctypes.byref(g), _lib.ptr(dy), _lib.ptr(w), _lib.ptr(dx), _lib.ptr(relu_bits), _lib.ptr(add), _lib.stream()),
              'seedhip_conv2d_bwd_data_bits_add')
        else:
          _lib.check(_lib.lib().seedhip_conv2d_bwd_data_bits(
              ctypes.byref(g), _lib.ptr(dy), _lib.ptr(w), _lib.ptr(dx), _lib.ptr(relu_bits), _lib.stream()),
              'seedhip_conv2d_bwd_data_bits')
      return dx
  with _region(_conv_name('conv_dgrad', g), flops, nbytes, pipe=lambda: _conv_pipe(g, 1)):
    with _dev(dx):
      ws, wsb = _splitk_ws(int(_lib.lib().seedhip_conv2d_bwd_data_workspace_bytes(ctypes.byref(g))), dx)
      _lib.check(_lib.lib().seedhip_conv2d_bwd_data_ws(
          ctypes.byref(g), _lib.ptr(dy), _lib.ptr(w), _lib.ptr(dx), _lib.ptr(relu_mask), _lib.ptr(add),
          _lib.ptr(ws), wsb, _lib.stream()), 'seedhip_conv2d_bwd_data')
    return dx


def conv2d_bwd_data_pool_supported(g):
  """conv2d_bwd_data_pool is served for this (conv) geometry."""
  return bool(_lib.lib().seedhip_conv2d_bwd_data_pool_supported(ctypes.byref(g)))


def conv2d_bwd_data_pool(g, dpooled, argmax, w, dx, d_prepool):
  """Max-pool backward + data gradient of the convolution in front of the pool, in one kernel: dx, and the pre-pool
  gradient d_prepool [n, oh, ow, cout] for the weight gradient (both bit-identical to maxpool_bwd + conv2d_bwd_data)."""
  flops, nbytes = _conv_cost(g)
  elems = g.n_img * g.oh * g.ow * g.cout
  # algorithmic bytes: pooled gradient + argmax in (elems / 4 x 5 B), dx and the pre-pool gradient out, the weights
  nbytes = (elems // 4) * 5 + 4 * elems + 4 * g.n_img * g.ih * g.iw * g.cin + 4 * g.kh * g.kw * g.cin * g.cout
  with _region(_conv_name('conv_dgrad_pool', g), flops, nbytes, pipe=lambda: _conv_pipe(g, 1)):
    with _dev(dx):
      _lib.check(_lib.lib().seedhip_conv2d_bwd_data_pool(
          ctypes.byref(g), _lib.ptr(dpooled), _lib.ptr(argmax), _lib.ptr(w), _lib.ptr(dx), _lib.ptr(d_prepool),
          _lib.stream()), 'seedhip_conv2d_bwd_data_pool')
  return dx


def conv2d_bwd_weight_workspace_bytes(g):
  return int(_lib.lib().seedhip_conv2d_bwd_weight_workspace_bytes(ctypes.byref(g)))


def conv2d_bwd_weight(g, x, dy, dw, dbias, workspace, in_dtype=IN_F32, in_relu=False):
  with _region(_conv_name('conv_wgrad', g), *_conv_cost(g, 1 if in_dtype else 4), pipe=lambda: _conv_pipe(g, 2)):
    with _dev(dw):
      _lib.check(_lib.lib().seedhip_conv2d_bwd_weight(
          ctypes.byref(g), _lib.ptr(x), in_dtype, int(in_relu), _lib.ptr(dy), _lib.ptr(dw), _lib.ptr(dbias),
          _lib.ptr(workspace), workspace.numel() * workspace.element_size(), _lib.stream()),
          'seedhip_conv2d_bwd_weight')


def stack_prepare(state, done_u8, T, B, HW, frames_ext, nvalid):
  with _region('stack_prepare', 0, B * HW * 7):
    with _dev(frames_ext):
      _lib.check(_lib.lib().seedhip_stack_prepare(
          _lib.ptr(state), _lib.ptr(done_u8), T, B, HW, _lib.ptr(frames_ext), _lib.ptr(nvalid), _lib.stream()),
          'seedhip_stack_prepare')


def stack_prepare_indexed(table, rows, zero_mask, done_u8, T, B, HW, frames_ext, nvalid):
  """stack_prepare reading column b's state from table[rows[b]] in place (zero where zero_mask[b])."""
  with _region('stack_prepare', 0, B * HW * 7):
    with _dev(frames_ext):
      _lib.check(_lib.lib().seedhip_stack_prepare_indexed(
          _lib.ptr(table), _lib.ptr(rows), _lib.ptr(zero_mask), _lib.ptr(done_u8), T, B, HW, _lib.ptr(frames_ext),
          _lib.ptr(nvalid), _lib.stream()), 'seedhip_stack_prepare_indexed')


def stack_pack_state_indexed(frames_ext, nvalid, T, B, HW, table, rows, valid_mask):
  """stack_pack_state writing column b's new state to table[rows[b]] in place (rows with valid_mask == 0 skipped)."""
  with _region('stack_pack_state', 0, B * HW * 7):
    with _dev(table):
      _lib.check(_lib.lib().seedhip_stack_pack_state_indexed(
          _lib.ptr(frames_ext), _lib.ptr(nvalid), T, B, HW, _lib.ptr(table), _lib.ptr(rows), _lib.ptr(valid_mask),
          _lib.stream()), 'seedhip_stack_pack_state_indexed')


def unpackbits_u16(packed, out):
  """football/observation.py:48-63: uint16 / int16 words [...] -> uint8 [... * 16] of 0 / 255."""
  with _dev(out):
    _lib.check(_lib.lib().seedhip_unpackbits_u16(_lib.ptr(packed), packed.numel(), _lib.ptr(out), _lib.stream()),
               'seedhip_unpackbits_u16')


def stack_frames_f32(frames_ext, nvalid, T, B, HW, out):
  with _dev(out):
    _lib.check(_lib.lib().seedhip_stack_frames_f32(
        _lib.ptr(frames_ext), _lib.ptr(nvalid), T, B, HW, _lib.ptr(out), _lib.stream()),
        'seedhip_stack_frames_f32')


def stack_pack_state(frames_ext, nvalid, T, B, HW, new_state):
  with _region('stack_pack_state', 0, B * HW * 7):
    with _dev(new_state):
      _lib.check(_lib.lib().seedhip_stack_pack_state(
          _lib.ptr(frames_ext), _lib.ptr(nvalid), T, B, HW, _lib.ptr(new_state), _lib.stream()),
          'seedhip_stack_pack_state')


def conv2d_stack_fwd_bits_supported(g):
  return bool(_lib.lib().seedhip_conv2d_stack_fwd_bits_supported(ctypes.byref(g)))


def conv2d_stack_fwd(g, frames_ext, nvalid, w, bias, out, out_relu=True, relu_bits=None):
  """relu_bits (uint8 [T * B * oh * ow, ld_out / 4], where conv2d_stack_fwd_bits_supported): also receives the ReLU mask
  of `out` as bytes, for conv2d_bwd_data(..., relu_bits=) of the next layer."""
  with _region('stack_conv_fwd', 2.0 * g.T * g.B * g.oh * g.ow * g.cout * g.kh * g.kw * 4, g.T * g.B * g.ih * g.iw + g.T * g.B * g.oh * g.ow * g.cout * 4, pipe=_stack_pipe):
    with _dev(out):
      if relu_bits is not None:
        if not out_relu:
          raise ValueError('conv2d_stack_fwd: relu_bits needs out_relu')
        _lib.check(_lib.lib().seedhip_conv2d_stack_fwd_bits(
            ctypes.byref(g), _lib.ptr(frames_ext), _lib.ptr(nvalid), _lib.ptr(w), _lib.ptr(bias), _lib.ptr(out),
            _lib.ptr(relu_bits), _lib.stream()), 'seedhip_conv2d_stack_fwd_bits')
        return
      _lib.check(_lib.lib().seedhip_conv2d_stack_fwd(
          ctypes.byref(g), _lib.ptr(frames_ext), _lib.ptr(nvalid), _lib.ptr(w), _lib.ptr(bias), _lib.ptr(out),
          int(out_relu), _lib.stream()), 'seedhip_conv2d_stack_fwd')


def conv2d_stack_bwd_weight_workspace_bytes(g):
  return int(_lib.lib().seedhip_conv2d_stack_bwd_weight_workspace_bytes(ctypes.byref(g)))


def conv2d_stack_bwd_weight(g, frames_ext, nvalid, dy, dw, dbias, workspace):
  with _region('stack_conv_wgrad', 2.0 * g.T * g.B * g.oh * g.ow * g.cout * g.kh * g.kw * 4, g.T * g.B * g.ih * g.iw + g.T * g.B * g.oh * g.ow * g.cout * 4, pipe=_stack_pipe):
    with _dev(dw):
      _lib.check(_lib.lib().seedhip_conv2d_stack_bwd_weight(
          ctypes.byref(g), _lib.ptr(frames_ext), _lib.ptr(nvalid), _lib.ptr(dy), _lib.ptr(dw), _lib.ptr(dbias),
          _lib.ptr(workspace), workspace.numel() * workspace.element_size(), _lib.stream()),
          'seedhip_conv2d_stack_bwd_weight')


def impala_loss_workspace_bytes(T, B):
  return int(_lib.lib().seedhip_impala_loss_workspace_bytes(T, B))


def impala_loss_fwd_bwd(logits, logits_ld, baseline, baseline_ld, beh_logits, actions, rewards, done_u8,
                        T, B, A, d_logits, d_baseline, scalars, workspace, vs=None, pg=None,
                        entropy_cost=0.00025, baseline_cost=0.5, kl_cost=0.0, discounting=0.99,
                        lambda_=1.0, max_abs_reward=0.0, clip_rho=1.0, clip_pg_rho=1.0,
                        mean_denominator=None, entropy_cost_param=None, d_entropy_cost_param=None,
                        entropy_cost_adjustment_speed=10.0, target_entropy=None):
  """entropy_cost_param (device scalar) given: the learner's learnable entropy cost exp(speed * param) and, with
  target_entropy, the adjustment loss of learner.py:127-135 (seedhip_impala_loss_fwd_bwd_adaptive)."""
  n = float(T * B if mean_denominator is None else mean_denominator)
  with _region('impala_loss', 0, T * B * (12 * A + 29)):
    with _dev(scalars):
      if entropy_cost_param is not None:
        _lib.check(_lib.lib().seedhip_impala_loss_fwd_bwd_adaptive(
            _lib.ptr(logits), logits_ld, _lib.ptr(baseline), baseline_ld, _lib.ptr(beh_logits), _lib.ptr(actions),
            actions.element_size(), _lib.ptr(rewards), _lib.ptr(done_u8), T, B, A,
            _lib.ptr(entropy_cost_param), float(entropy_cost_adjustment_speed), int(bool(target_entropy)),
            float(target_entropy or 0.0), _lib.ptr(d_entropy_cost_param),
            baseline_cost, kl_cost, discounting, lambda_, max_abs_reward, clip_rho, clip_pg_rho,
            n, _lib.ptr(d_logits), _lib.ptr(d_baseline), _lib.ptr(vs), _lib.ptr(pg), _lib.ptr(scalars),
            _lib.ptr(workspace), workspace.numel() * workspace.element_size(), _lib.stream()),
            'seedhip_impala_loss_fwd_bwd_adaptive')
        return
      _lib.check(_lib.lib().seedhip_impala_loss_fwd_bwd(
          _lib.ptr(logits), logits_ld, _lib.ptr(baseline), baseline_ld, _lib.ptr(beh_logits), _lib.ptr(actions),
          actions.element_size(), _lib.ptr(rewards), _lib.ptr(done_u8), T, B, A,
          entropy_cost, baseline_cost, kl_cost, discounting, lambda_, max_abs_reward, clip_rho, clip_pg_rho,
          n, _lib.ptr(d_logits), _lib.ptr(d_baseline), _lib.ptr(vs), _lib.ptr(pg), _lib.ptr(scalars),
          _lib.ptr(workspace), workspace.numel() * workspace.element_size(), _lib.stream()),
          'seedhip_impala_loss_fwd_bwd')


def adam_flat(params, grads, m, v, lr_t, beta_1, beta_2, epsilon, grad_scale=1.0, clamp=None, guard=None):
  """clamp = (index, lo, hi): params[index] is clipped to [lo, hi] after its update (Keras variable constraint).
  guard: int32[1] device tensor; non-zero on the device = the update is skipped (seedhip_adam_flat_guarded)."""
  ci, lo, hi = clamp if clamp is not None else (-1, 0.0, 0.0)
  with _region('adam_flat', 0, params.numel() * 28):
    with _dev(params):
      _lib.check(_lib.lib().seedhip_adam_flat_guarded(
          _lib.ptr(params), _lib.ptr(grads), _lib.ptr(m), _lib.ptr(v), params.numel(), lr_t, None, beta_1, beta_2,
          epsilon, grad_scale, int(ci), float(lo), float(hi), _lib.ptr(guard), _lib.stream()), 'seedhip_adam_flat')


def adam_flat_dev_lr(params, grads, m, v, lr_t_dev, beta_1, beta_2, epsilon, grad_scale=1.0, clamp=None, guard=None):
  ci, lo, hi = clamp if clamp is not None else (-1, 0.0, 0.0)
  with _region('adam_flat', 0, params.numel() * 28):
    with _dev(params):
      _lib.check(_lib.lib().seedhip_adam_flat_guarded(
          _lib.ptr(params), _lib.ptr(grads), _lib.ptr(m), _lib.ptr(v), params.numel(), 0.0, _lib.ptr(lr_t_dev), beta_1,
          beta_2, epsilon, grad_scale, int(ci), float(lo), float(hi), _lib.ptr(guard), _lib.stream()),
          'seedhip_adam_flat_dev_lr')


def global_norm_workspace_bytes():
  return int(_lib.lib().seedhip_global_norm_workspace_bytes())


def clip_by_global_norm(grads, clip_norm, sumsq_out, workspace):
  with _dev(grads):
    _lib.check(_lib.lib().seedhip_clip_by_global_norm(
        _lib.ptr(grads), grads.numel(), float(clip_norm), _lib.ptr(sumsq_out), _lib.ptr(workspace),
        workspace.numel() * workspace.element_size(), _lib.stream()), 'seedhip_clip_by_global_norm')


def conv3x3_u8_pool_fwd(x_u8, w, bias, pooled, argmax, pooled_bits=None):
  """Fused Conv2D(16, 3, 'same')(x/255) + MaxPool2D(3, 2, 'same') of ImpalaDeep's first stage
  (dmlab/networks.py:31-37, :98-100).  x_u8 [n, ih, iw, 3]; pooled / argmax [n, ceil(ih/2), ceil(iw/2), 16]."""
  n, ih, iw, cin = x_u8.shape
  cout = pooled.shape[-1]
  with _region('convpool_fwd[%dx%dx%d->%d]' % (ih, iw, cin, cout), 2.0 * n * ih * iw * cout * 9 * cin,
               x_u8.numel() + pooled.numel() * 5):
    with _dev(pooled):
      _lib.check(_lib.lib().seedhip_conv3x3_u8_pool_fwd_bits(
          _lib.ptr(x_u8), n, ih, iw, cin, _lib.ptr(w), _lib.ptr(bias), cout, _lib.ptr(pooled), _lib.ptr(argmax),
          _lib.ptr(pooled_bits), _lib.stream()), 'seedhip_conv3x3_u8_pool_fwd_bits')


def conv3x3_u8_pool_bwd_workspace_bytes(n, ih, iw):
  return int(_lib.lib().seedhip_conv3x3_u8_pool_bwd_workspace_bytes(n, ih, iw))


def conv3x3_u8_pool_bwd(x_u8, dpooled, argmax, dw, dbias, workspace):
  n, ih, iw, cin = x_u8.shape
  cout = dpooled.shape[-1]
  with _region('convpool_bwd[%dx%dx%d->%d]' % (ih, iw, cin, cout), 2.0 * n * ih * iw * cout * 9 * cin,
               x_u8.numel() + dpooled.numel() * 5):
    with _dev(dw):
      _lib.check(_lib.lib().seedhip_conv3x3_u8_pool_bwd(
          _lib.ptr(x_u8), n, ih, iw, cin, _lib.ptr(dpooled), _lib.ptr(argmax), cout, _lib.ptr(dw), _lib.ptr(dbias),
          _lib.ptr(workspace), workspace.numel() * workspace.element_size(), _lib.stream()),
                 'seedhip_conv3x3_u8_pool_bwd')


def maxpool_fwd(x, y, argmax, y_bits=None):
  """MaxPool2D(3, 2, 'same') on NHWC fp32 (dmlab/networks.py:36-37); y_bits (uint8 [pixels, c / 4]): the ReLU mask of y
  as bytes (conv2d_fwd(out_bits=))."""
  n, ih, iw, c = x.shape
  with _region('maxpool_fwd[%dx%dx%d]' % (ih, iw, c), 0, x.numel() * 4 + y.numel() * 5):
    with _dev(y):
      _lib.check(_lib.lib().seedhip_maxpool3x3s2_same_fwd_bits(n, ih, iw, c, _lib.ptr(x), _lib.ptr(y), _lib.ptr(argmax),
                                                              _lib.ptr(y_bits), _lib.stream()),
                 'seedhip_maxpool3x3s2_same_fwd_bits')


def maxpool_bwd(dy, argmax, dx):
  n, ih, iw, c = dx.shape
  with _region('maxpool_bwd[%dx%dx%d]' % (ih, iw, c), 0, dx.numel() * 4 + dy.numel() * 5):
    with _dev(dx):
      _lib.check(_lib.lib().seedhip_maxpool3x3s2_same_bwd(n, ih, iw, c, _lib.ptr(dy), _lib.ptr(argmax), _lib.ptr(dx),
                                                         _lib.stream()), 'seedhip_maxpool3x3s2_same_bwd')


def lstm_assemble_inputs(x, ldx, feat, num_actions, reward, prev_actions, clip_reward, rows):
  with _dev(x):
    _lib.check(_lib.lib().seedhip_lstm_assemble_inputs(
        _lib.ptr(x), ldx, feat, num_actions, _lib.ptr(reward), _lib.ptr(prev_actions), prev_actions.element_size(),
        int(clip_reward), rows, _lib.stream()), 'seedhip_lstm_assemble_inputs')


def as_u8(t):
  """bool -> uint8 without a copy (torch stores bool as one byte holding 0 / 1); anything else is converted."""
  t = t.contiguous()
  return t.view(torch.uint8) if t.dtype == torch.bool else t.to(torch.uint8)


def lstm_mask_state(h0, c0, done0_u8, B, H, hin, cin):
  with _dev(hin):
    _lib.check(_lib.lib().seedhip_lstm_mask_state(_lib.ptr(h0), _lib.ptr(c0), _lib.ptr(done0_u8), B, H, _lib.ptr(hin),
                                                  _lib.ptr(cin), _lib.stream()), 'seedhip_lstm_mask_state')


def lstm_step_supported(B, H):
  return bool(_lib.lib().seedhip_lstm_step_supported(B, H))


def lstm_permute_u(u, H, up):
  with _dev(up):
    _lib.check(_lib.lib().seedhip_lstm_permute_u(_lib.ptr(u), H, _lib.ptr(up), _lib.stream()), 'seedhip_lstm_permute_u')


def lstm_step_fwd(hin, up, zx, cin, done_next_u8, B, H, z, h_out, ld_h, hin_next, cin_next):
  """One LSTM step in one launch: z = zx + hin U, gates, done-reset of the next state (dmlab/networks.py:152-171)."""
  with _region('lstm_step_fwd', 2.0 * B * H * 4 * H, (H * 4 * H + B * H * 12) * 4):
    with _dev(z):
      _lib.check(_lib.lib().seedhip_lstm_step_fwd(
          _lib.ptr(hin), _lib.ptr(up), _lib.ptr(zx), _lib.ptr(cin), _lib.ptr(done_next_u8), B, H, _lib.ptr(z),
          _lib.ptr(h_out), ld_h, _lib.ptr(hin_next), _lib.ptr(cin_next), _lib.stream()), 'seedhip_lstm_step_fwd')


def lstm_seq_supported(T1, B, H):
  return bool(_lib.lib().seedhip_lstm_seq_supported(T1, B, H))


def lstm_seq_fwd(up, zx, done_u8, T1, B, H, z, h_out, ld_h, hin, cin, sync_ws, sticky=None):
  """All T1 LSTM steps in one launch (resident workgroups + grid barrier); bit-identical to T1 lstm_step_fwd calls.
  sync_ws: int32[2] device tensor; sync_ws[1] != 0 afterwards = barrier timed out, outputs invalid.
  sticky: int32[1] device tensor set to 1 on a timeout and never cleared by the library (adam_flat*(guard=))."""
  with _region('lstm_seq_fwd', 2.0 * T1 * B * H * 4 * H, (H * 4 * H + T1 * B * H * 12) * 4):
    with _dev(z):
      _lib.check(_lib.lib().seedhip_lstm_seq_fwd_sticky(
          _lib.ptr(up), _lib.ptr(zx), _lib.ptr(done_u8), T1, B, H, _lib.ptr(z), _lib.ptr(h_out), ld_h, _lib.ptr(hin),
          _lib.ptr(cin), _lib.ptr(sync_ws), _lib.ptr(sticky), _lib.stream()), 'seedhip_lstm_seq_fwd')


def lstm_seq_bwd_workspace_bytes(B, H):
  return int(_lib.lib().seedhip_lstm_seq_bwd_workspace_bytes(B, H))


def lstm_seq_bwd(up, z, cin, dh_out, ld_dh, done_u8, T1, B, H, dz, ring_ws, sync_ws, sticky=None):
  """The whole backward recurrence (cell backward + dh_rec = dz U^T per step) in one launch."""
  with _region('lstm_seq_bwd', 2.0 * (T1 - 1) * B * H * 4 * H, (H * 4 * H + T1 * B * H * 11) * 4):
    with _dev(dz):
      _lib.check(_lib.lib().seedhip_lstm_seq_bwd_sticky(
          _lib.ptr(up), _lib.ptr(z), _lib.ptr(cin), _lib.ptr(dh_out), ld_dh, _lib.ptr(done_u8), T1, B, H, _lib.ptr(dz),
          _lib.ptr(ring_ws), _lib.ptr(sync_ws), _lib.ptr(sticky), _lib.stream()), 'seedhip_lstm_seq_bwd')


def lstm_gates_fwd(z, cin, done_next_u8, B, H, h_out, ld_h, hin_next, cin_next):
  with _region('lstm_gates_fwd', 0, B * H * 4 * 8):
    with _dev(h_out):
      _lib.check(_lib.lib().seedhip_lstm_gates_fwd(
          _lib.ptr(z), _lib.ptr(cin), _lib.ptr(done_next_u8), B, H, _lib.ptr(h_out), ld_h, _lib.ptr(hin_next),
          _lib.ptr(cin_next), _lib.stream()), 'seedhip_lstm_gates_fwd')


def lstm_gates_bwd(z, cin, dh_out, ld_dh, dh_rec, dc_rec, done_next_u8, B, H, dz, dc_prev):
  with _region('lstm_gates_bwd', 0, B * H * 4 * 13):
    with _dev(dz):
      _lib.check(_lib.lib().seedhip_lstm_gates_bwd(
          _lib.ptr(z), _lib.ptr(cin), _lib.ptr(dh_out), ld_dh, _lib.ptr(dh_rec), _lib.ptr(dc_rec),
          _lib.ptr(done_next_u8), B, H, _lib.ptr(dz), _lib.ptr(dc_prev), _lib.stream()), 'seedhip_lstm_gates_bwd')


def dueling_fwd(va, ld, rows, A, q, action):
  with _dev(q):
    _lib.check(_lib.lib().seedhip_dueling_fwd(_lib.ptr(va), ld, rows, A, _lib.ptr(q), _lib.ptr(action), _lib.stream()),
               'seedhip_dueling_fwd')


def dueling_bwd(dq, rows, A, d_va, ld):
  with _dev(d_va):
    _lib.check(_lib.lib().seedhip_dueling_bwd(_lib.ptr(dq), rows, A, _lib.ptr(d_va), ld, _lib.stream()),
               'seedhip_dueling_bwd')


def r2d2_loss_workspace_bytes(T, B, n_steps):
  return int(_lib.lib().seedhip_r2d2_loss_workspace_bytes(T, B, n_steps))


def r2d2_loss_fwd_bwd(training_q, target_q, actions_i32, rewards, done_u8, importance_weights, T, B, A, gamma,
                      n_steps, eta, epsilon, mean_denominator, loss_b, prio_b, d_training_q, total, workspace):
  with _region('r2d2_loss', 0, T * B * (12 * A + 20)):
    with _dev(total):
      _lib.check(_lib.lib().seedhip_r2d2_loss_fwd_bwd(
          _lib.ptr(training_q), _lib.ptr(target_q), _lib.ptr(actions_i32), _lib.ptr(rewards), _lib.ptr(done_u8),
          _lib.ptr(importance_weights), T, B, A, gamma, n_steps, eta, epsilon, float(mean_denominator),
          _lib.ptr(loss_b), _lib.ptr(prio_b), _lib.ptr(d_training_q), _lib.ptr(total), _lib.ptr(workspace),
          workspace.numel() * workspace.element_size(), _lib.stream()), 'seedhip_r2d2_loss_fwd_bwd')


def rows_move(dst, dst_rows, src, src_rows, n, row_bytes):
  """dst[dst_rows[i]] = src[src_rows[i]] over rows of row_bytes bytes (None rows = identity; src None = zeros)."""
  if n == 0:
    return
  with _dev(dst):
    _lib.check(_lib.lib().seedhip_rows_move(_lib.ptr(dst), _lib.ptr(dst_rows), _lib.ptr(src), _lib.ptr(src_rows), n,
                                            row_bytes, _lib.stream()), 'seedhip_rows_move')


def rows_move_masked(dst, dst_rows, src, src_rows, n, row_bytes, mask_u8, zero_where_masked=False):
  """rows_move with a per-row mask (see seedhip_rows_move_masked)."""
  if n == 0:
    return
  with _dev(dst):
    _lib.check(_lib.lib().seedhip_rows_move_masked(
        _lib.ptr(dst), _lib.ptr(dst_rows), _lib.ptr(src), _lib.ptr(src_rows), n, row_bytes, _lib.ptr(mask_u8),
        int(zero_where_masked), _lib.stream()), 'seedhip_rows_move_masked')


def inference_pre(env_ids, run_ids, reward, raw_reward, done_u8, n, num_envs, num_action_repeats, run_ids_tab,
                  info_frames, info_return, info_raw, actions_tab, store_index, reset_mask, prev_actions,
                  episode_stats, stats_count, error_flag, ids_safe, valid, stamp_tab, call_counter, will_complete=None,
                  full_length=0):
  with _dev(reset_mask):
    _lib.check(_lib.lib().seedhip_inference_pre(
        _lib.ptr(env_ids), _lib.ptr(run_ids), _lib.ptr(reward), _lib.ptr(raw_reward), _lib.ptr(done_u8), n, num_envs,
        num_action_repeats, _lib.ptr(run_ids_tab), _lib.ptr(info_frames), _lib.ptr(info_return), _lib.ptr(info_raw),
        _lib.ptr(actions_tab), _lib.ptr(store_index), _lib.ptr(reset_mask), _lib.ptr(prev_actions),
        _lib.ptr(episode_stats), episode_stats.shape[0], _lib.ptr(stats_count), _lib.ptr(error_flag),
        _lib.ptr(ids_safe), _lib.ptr(valid), _lib.ptr(stamp_tab), _lib.ptr(call_counter), _lib.ptr(will_complete),
        int(full_length), _lib.stream()), 'seedhip_inference_pre')


def inference_post(env_ids, valid, actions, logits, logits_ld, num_actions, rng_state, n, num_envs, full_length,
                   batch_capacity, store_index, actions_tab, batch_count, append_rows, complete, carry, batch_cols,
                   emit_env, emit_col, emit_count, last_rows, error_flag, batch_start=None):
  """logits given (a view whose first element is row 0's first logit): the actions are sampled in the kernel and
  written to `actions`; logits None: `actions` is an input."""
  with _dev(complete):
    _lib.check(_lib.lib().seedhip_inference_post(
        _lib.ptr(env_ids), _lib.ptr(valid), _lib.ptr(actions), _lib.ptr(logits), logits_ld, num_actions,
        _lib.ptr(rng_state), n, num_envs, full_length, batch_capacity, _lib.ptr(store_index),
        _lib.ptr(actions_tab), _lib.ptr(batch_count), _lib.ptr(append_rows), _lib.ptr(complete), _lib.ptr(carry),
        _lib.ptr(batch_cols), _lib.ptr(emit_env), _lib.ptr(emit_col), _lib.ptr(emit_count), _lib.ptr(last_rows),
        _lib.ptr(error_flag), _lib.ptr(batch_start), _lib.stream()), 'seedhip_inference_post')


def emit_unrolls(dsts, srcs, row_bytes, emit_env, emit_col, emit_count, max_unrolls, full_length, num_envs,
                 batch_capacity):
  """Completed unrolls (compact list of inference_post) -> training batch, every field in one launch (groups of 16)."""
  for lo in range(0, len(dsts), 16):
    d, s_, rb = dsts[lo:lo + 16], srcs[lo:lo + 16], row_bytes[lo:lo + 16]
    k = len(d)
    PA = ctypes.c_void_p * k
    da, sa = PA(*[t.data_ptr() for t in d]), PA(*[t.data_ptr() for t in s_])
    ra = (ctypes.c_longlong * k)(*rb)
    with _dev(d[0]):
      _lib.check(_lib.lib().seedhip_emit_unrolls(
          k, ctypes.cast(da, ctypes.c_void_p), ctypes.cast(sa, ctypes.c_void_p), ctypes.cast(ra, ctypes.c_void_p),
          _lib.ptr(emit_env), _lib.ptr(emit_col), _lib.ptr(emit_count), int(max_unrolls), int(full_length),
          int(num_envs), int(batch_capacity), _lib.stream()), 'seedhip_emit_unrolls')


def categorical_sample(logits, ld, rows, num_actions, rng_state, actions):
  """actions[r] ~ Categorical(logits row r) (Gumbel-max over Philox randoms; advances rng_state[1])."""
  with _dev(actions):
    _lib.check(_lib.lib().seedhip_categorical_sample(_lib.ptr(logits), ld, rows, num_actions, _lib.ptr(rng_state),
                                                     _lib.ptr(actions), _lib.stream()), 'seedhip_categorical_sample')


def row_op(dst, src, row_bytes, n, dst_rows=None, src_rows=None, mask=None, zero_where_masked=False, dst_pitch=0,
           src_pitch=0):
  """One operation of rows_move_ops (tensors are only pointer carriers; the caller keeps them alive)."""
  return (dst, src, int(row_bytes), int(dst_pitch), int(src_pitch), dst_rows, src_rows, int(n), mask,
          int(zero_where_masked))


def rows_move_ops(op_list):
  """Independent row moves in one launch (groups of 32)."""
  op_list = [o for o in op_list if o[7] > 0]
  for lo in range(0, len(op_list), 32):
    chunk = op_list[lo:lo + 32]
    arr = (_lib.RowOp * len(chunk))()
    for k, (dst, src, rb, dp, sp, dr, sr, n, mask, z) in enumerate(chunk):
      a = arr[k]
      a.dst, a.src, a.row_bytes, a.dst_pitch, a.src_pitch = dst.data_ptr(), (src.data_ptr() if src is not None else None), rb, dp, sp
      a.dst_rows = dr.data_ptr() if dr is not None else None
      a.src_rows = sr.data_ptr() if sr is not None else None
      a.n, a.row_mask, a.zero_where_masked = n, (mask.data_ptr() if mask is not None else None), z
    with _dev(chunk[0][0]):
      _lib.check(_lib.lib().seedhip_rows_move_ops(len(chunk), ctypes.cast(arr, ctypes.c_void_p), _lib.stream()),
                 'seedhip_rows_move_ops')


def rows_move_multi(dsts, srcs, row_bytes, dst_rows, src_rows, n, mask_u8=None, zero_where_masked=False):
  """One launch moving rows of several fields (lists of tensors; src entries may be None = zeros)."""
  nf = len(dsts)
  if n == 0 or nf == 0:
    return
  for lo in range(0, nf, 16):
    d, s_, rb = dsts[lo:lo + 16], srcs[lo:lo + 16], row_bytes[lo:lo + 16]
    k = len(d)
    PA = ctypes.c_void_p * k
    da = PA(*[t.data_ptr() for t in d])
    sa = PA(*[(t.data_ptr() if t is not None else None) for t in s_])
    ra = (ctypes.c_longlong * k)(*rb)
    with _dev(d[0]):
      _lib.check(_lib.lib().seedhip_rows_move_multi(
          k, ctypes.cast(da, ctypes.c_void_p), ctypes.cast(sa, ctypes.c_void_p), ctypes.cast(ra, ctypes.c_void_p),
          _lib.ptr(dst_rows), _lib.ptr(src_rows), n, _lib.ptr(mask_u8), int(zero_where_masked), _lib.stream()),
          'seedhip_rows_move_multi')


def replay_sample(priorities, limit, priority_exp, is_exp, uniforms, indices, weights, workspace):
  with _dev(indices):
    _lib.check(_lib.lib().seedhip_replay_sample(
        _lib.ptr(priorities), limit, float(priority_exp), float(is_exp), _lib.ptr(uniforms), uniforms.numel(),
        _lib.ptr(indices), _lib.ptr(weights), _lib.ptr(workspace), workspace.numel() * workspace.element_size(),
        _lib.stream()), 'seedhip_replay_sample')


def replay_sample_workspace_bytes(limit):
  return int(_lib.lib().seedhip_replay_sample_workspace_bytes(limit))


def epsilon_greedy(actions, env_ids, epsilons, num_actions, rng_state, replaced=None):
  """apply_epsilon_greedy (agents/r2d2/learner.py:147-177) in place on int64 actions; advances rng_state[1]."""
  with _dev(actions):
    _lib.check(_lib.lib().seedhip_epsilon_greedy(
        _lib.ptr(actions), _lib.ptr(env_ids), _lib.ptr(epsilons), actions.numel(), epsilons.numel(), int(num_actions),
        _lib.ptr(rng_state), _lib.ptr(replaced), _lib.stream()), 'seedhip_epsilon_greedy')


def replay_time_rows(slots, steps, replay_rows, batch_rows):
  """Row indices moving `slots.numel()` unrolls of `steps` steps between replay rows and a time-major batch."""
  with _dev(slots):
    _lib.check(_lib.lib().seedhip_replay_time_rows(_lib.ptr(slots), slots.numel(), int(steps), _lib.ptr(replay_rows),
                                                   _lib.ptr(batch_rows), _lib.stream()), 'seedhip_replay_time_rows')


def heads_supported(feat, ldh):
  return bool(_lib.lib().seedhip_heads_supported(int(feat), int(ldh)))


def heads_fwd(x, ldx, w, bias, rows, feat, ldh, y):
  """y [rows, ldh] = x [rows, feat] w [feat, ldh] + bias (the packed policy / baseline heads, dmlab/networks.py:116-124)."""
  with _region('heads_fwd', 2.0 * rows * feat * ldh, rows * (feat + ldh) * 4):
    with _dev(y):
      _lib.check(_lib.lib().seedhip_heads_fwd(_lib.ptr(x), int(ldx), _lib.ptr(w), _lib.ptr(bias), int(rows), int(feat),
                                              int(ldh), _lib.ptr(y), _lib.stream()), 'seedhip_heads_fwd')


# ---- central inference in six launches for the frame-stacked Atari agents (csrc/servestep.hip) ---------------------- #
def serve_conv0_split_bytes(cout):
  return int(_lib.lib().seedhip_serve_conv0_split_bytes(int(cout)))


def serve_split_conv0(w0, cout, split):
  """The first conv's W / 255 as three bf16 planes in the register image of its kernel (what serve_begin also does)."""
  with _dev(split):
    _lib.check(_lib.lib().seedhip_serve_split_conv0(_lib.ptr(w0), int(cout), _lib.ptr(split), _lib.stream()),
               'seedhip_serve_split_conv0')


def serve_heads_image_bytes(feat):
  return int(_lib.lib().seedhip_serve_heads_image_bytes(int(feat)))


def serve_begin(step, w0, cout, split, heads_w, feat, ldh, heads_image, like):
  """step: _lib.ServeStep.  All bookkeeping of learner.py:353-381 + store index / completions + the first conv's weight
  planes and the heads' register image."""
  with _region('serve_begin', 0, 0):
    with _dev(like):
      _lib.check(_lib.lib().seedhip_serve_begin(ctypes.byref(step), _lib.ptr(w0), int(cout), _lib.ptr(split),
                                                _lib.ptr(heads_w), int(feat), int(ldh), _lib.ptr(heads_image),
                                                _lib.stream()), 'seedhip_serve_begin')


def conv2d_stack_fwd_rows_supported(g):
  return bool(_lib.lib().seedhip_conv2d_stack_fwd_rows_supported(ctypes.byref(g)))


def conv2d_stack_fwd_rows(g, obs, store_obs, hist_rows, nvalid, w_split, bias, out, out_relu=True):
  """First conv of one inference step: stacks = request frames + the store's history rows."""
  n = g.B
  flops = 2.0 * n * g.oh * g.ow * g.cout * g.kh * g.kw * 4
  nbytes = n * g.ih * g.iw * 4 + n * g.oh * g.ow * g.cout * 4           # 4 frames read, activation written
  with _region('stack_conv_fwd_rows', flops, nbytes, pipe=_stack_pipe):
    with _dev(out):
      _lib.check(_lib.lib().seedhip_conv2d_stack_fwd_rows(
          ctypes.byref(g), _lib.ptr(obs), _lib.ptr(store_obs), _lib.ptr(hist_rows),
          _lib.ptr(nvalid), _lib.ptr(w_split), _lib.ptr(bias), _lib.ptr(out), int(out_relu), _lib.stream()),
          'seedhip_conv2d_stack_fwd_rows')
  return out


def dense_fwd_partial_workspace_bytes(g):
  return int(_lib.lib().seedhip_dense_fwd_partial_workspace_bytes(ctypes.byref(g)))


def dense_fwd_partial(g, x, w, workspace, in_relu=False):
  """Split-K partial sums of a Dense layer [slices][rows][cout] in `workspace`; returns the number of slices."""
  slices = ctypes.c_int(0)
  with _region(_conv_name('dense_fwd_partial', g), *_conv_cost(g, 4), pipe=lambda: _conv_pipe(g, 0)):
    with _dev(workspace):
      _lib.check(_lib.lib().seedhip_dense_fwd_partial(
          ctypes.byref(g), _lib.ptr(x), int(in_relu), _lib.ptr(w), _lib.ptr(workspace),
          workspace.numel() * workspace.element_size(), ctypes.byref(slices), _lib.stream()), 'seedhip_dense_fwd_partial')
  return slices.value


def serve_finish(step, fields, fc_partial, slices, fc_bias, feat, heads_image, heads_b, ldh, num_actions, actions, obs,
                 store_obs, hw):
  n = step.n
  with _region('serve_finish', 2.0 * n * feat * ldh, n * (slices * feat + ldh) * 4 + 2 * n * hw):
    with _dev(actions):
      _lib.check(_lib.lib().seedhip_serve_finish(
          ctypes.byref(step), ctypes.byref(fields), _lib.ptr(fc_partial), int(slices), _lib.ptr(fc_bias), int(feat),
          _lib.ptr(heads_image), _lib.ptr(heads_b), int(ldh), int(num_actions), _lib.ptr(actions), _lib.ptr(obs),
          _lib.ptr(store_obs), int(hw), _lib.stream()), 'seedhip_serve_finish')


def serve_emit(step, batch, store, row_bytes, first_table, batch_first, store_obs, hw):
  """Completed unrolls -> training batch ring, carry, first-state hand-over (<= 16 fields per call)."""
  k = len(batch)
  PA = ctypes.c_void_p * k
  da, sa = PA(*[t.data_ptr() for t in batch]), PA(*[t.data_ptr() for t in store])
  ra = (ctypes.c_longlong * k)(*row_bytes)
  with _region('serve_emit', 0, 0):
    with _dev(first_table):
      _lib.check(_lib.lib().seedhip_serve_emit(
          ctypes.byref(step), k, ctypes.cast(da, ctypes.c_void_p), ctypes.cast(sa, ctypes.c_void_p),
          ctypes.cast(ra, ctypes.c_void_p), _lib.ptr(first_table), _lib.ptr(batch_first), _lib.ptr(store_obs), int(hw),
          _lib.stream()), 'seedhip_serve_emit')



# ---- streams on a subset of the compute units ------------------------------------------------------------------------ #
def cu_mask_stream(device, keep, num_cus=256):
  """A torch stream whose kernels run only on the compute units i with keep(i) true (hipExtStreamCreateWithCUMask through
  the C ABI).  Returns (torch.cuda.ExternalStream, number of CUs kept).  The stream lives as long as the process."""
  words = (num_cus + 31) // 32
  mask = (ctypes.c_uint * words)()
  kept = 0
  for i in range(num_cus):
    if keep(i):
      mask[i // 32] |= 1 << (i % 32)
      kept += 1
  if kept == 0:
    raise ValueError('cu_mask_stream: empty mask')
  out = ctypes.c_void_p()
  with torch.cuda.device(device):
    _lib.check(_lib.lib().seedhip_stream_create_cu_mask(ctypes.cast(mask, ctypes.c_void_p), words, ctypes.byref(out)),
               'seedhip_stream_create_cu_mask')
  return torch.cuda.ExternalStream(out.value, device=device), kept
