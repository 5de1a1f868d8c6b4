"""GPU parity tests: every HIP kernel, called through the C ABI, vs the CPU oracle
on identical seeded inputs.  Tolerances are stated per test."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import categorical_np, frames_np, loss_np, nets_torch, vtrace_np
from tests import synth

pytestmark = pytest.mark.gpu


def dev(a, device):
  return torch.as_tensor(np.ascontiguousarray(a)).to(device)


# ------------------------------- V-trace ------------------------------------- #
def _run_vtrace(device, inp, **kw):
  from seed_rl_amd import vtrace
  t = {k: dev(v, device) for k, v in inp.items()}
  out = vtrace.from_importance_weights(**t, **kw)
  return out.vs.cpu().numpy(), out.pg_advantages.cpu().numpy()


@pytest.mark.parametrize('seed', [0, 1, 2])
@pytest.mark.parametrize('kw', [dict(), dict(lambda_=0.95),
                                dict(clip_rho_threshold=3.7, clip_pg_rho_threshold=2.2),
                                dict(clip_rho_threshold=None, clip_pg_rho_threshold=None)])
def test_vtrace_cfg1_parity(device, seed, kw):
  """BASELINE.json configs[0]: [T=20,B=32,A=6]; bar = 1e-5 max-abs vs the fp32 oracle."""
  for stress in (False, True):
    inp = synth.vtrace_inputs(seed, 20, 32, 6, stress=stress)
    vs, pg = _run_vtrace(device, inp, **kw)
    ref = vtrace_np.from_importance_weights(**inp, **kw)
    # 1e-5 absolute is the north_star bar for the learner's setting (rho clipped).  With
    # clipping disabled and |log rho| up to 2.5 the outputs reach ~1e3, where one fp32 ulp
    # already exceeds 1e-5: that case is held to 4 ulp of the output scale instead.
    unclipped = kw.get('clip_rho_threshold', 1.0) is None
    tol_vs = 4 * np.spacing(np.float32(np.abs(ref.vs).max())) if unclipped else 1e-5
    tol_pg = 4 * np.spacing(np.float32(np.abs(ref.pg_advantages).max())) if unclipped else 1e-5
    assert np.max(np.abs(vs - ref.vs)) <= max(tol_vs, 1e-5)
    assert np.max(np.abs(pg - ref.pg_advantages)) <= max(tol_pg, 1e-5)
    ref64 = vtrace_np.from_importance_weights(**inp, **kw, dtype=np.float64)
    assert np.max(np.abs(vs - ref64.vs)) <= 1e-4 * max(1.0, np.abs(ref64.vs).max() / 50)  # fp32 distance to fp64


def test_vtrace_reference_golden(device):
  """tests/vtrace_test.py:120-145 inputs through the HIP kernel."""
  from tests.test_oracle_golden import _ref_vtrace_inputs
  v = _ref_vtrace_inputs()
  kw = dict(clip_rho_threshold=v.pop('clip_rho_threshold'), clip_pg_rho_threshold=v.pop('clip_pg_rho_threshold'))
  vs, pg = _run_vtrace(device, v, **kw)
  gt = vtrace_np.ground_truth_calculation(**v, **kw)
  np.testing.assert_allclose(vs, gt.vs, rtol=1e-6, atol=1e-6)         # reference's own tolerance
  np.testing.assert_allclose(pg, gt.pg_advantages, rtol=1e-6, atol=1e-6)


# the float4 path picks its cache policy by working set (csrc/vtrace.hip): (20, 262144) = 147 MB plain loads / stores,
# (100, 262144) = 0.73 GB non-temporal with one step in flight, (240, 262144) = 1.76 GB non-temporal with two
@pytest.mark.parametrize('T,B', [(1, 1), (3, 7), (20, 512), (20, 4096), (5, 65536), (20, 262144), (100, 33),
                                 (100, 262144), (240, 262144)])
def test_vtrace_shapes(device, T, B):
  inp = synth.vtrace_inputs(3, T, B, 4) if B <= 4096 else None
  if inp is None:
    rng = np.random.default_rng(T + B)
    inp = dict(target_action_log_probs=rng.uniform(-2, 0, (T, B)).astype(np.float32),
               behaviour_action_log_probs=rng.uniform(-2, 0, (T, B)).astype(np.float32),
               discounts=(0.99 * (rng.uniform(size=(T, B)) > 0.05)).astype(np.float32),
               rewards=rng.uniform(0, 3, (T, B)).astype(np.float32),
               values=rng.uniform(0, 3, (T, B)).astype(np.float32),
               bootstrap_value=rng.uniform(0, 3, (B,)).astype(np.float32))
  vs, pg = _run_vtrace(device, inp, lambda_=0.95)
  ref = vtrace_np.from_importance_weights(**inp, lambda_=0.95)
  assert np.max(np.abs(vs - ref.vs)) <= 1e-5 and np.max(np.abs(pg - ref.pg_advantages)) <= 1e-5


def test_vtrace_extra_dims_and_errors(device):
  """vtrace.py:49-51 trailing dims; :99-107 rank errors; empty T."""
  from seed_rl_amd import vtrace
  rng = np.random.default_rng(0)
  T, B, C = 6, 5, 3
  mk = lambda *s: rng.uniform(-1, 1, s).astype(np.float32)
  inp = dict(target_action_log_probs=mk(T, B, C), behaviour_action_log_probs=mk(T, B, C),
             discounts=np.full((T, B, C), 0.9, np.float32), rewards=mk(T, B, C), values=mk(T, B, C),
             bootstrap_value=mk(B, C))
  vs, pg = _run_vtrace(device, inp)
  ref = vtrace_np.from_importance_weights(**inp)
  assert vs.shape == (T, B, C) and np.max(np.abs(vs - ref.vs)) <= 1e-5
  bad = {k: dev(v, device) for k, v in inp.items()}
  bad['bootstrap_value'] = bad['values']
  with pytest.raises(ValueError):
    vtrace.from_importance_weights(**bad)
  with pytest.raises(Exception):
    vtrace.from_importance_weights(**{k: torch.as_tensor(v) for k, v in inp.items()})   # CPU tensors: no fallback
  e = {k: dev(v[:0] if k != 'bootstrap_value' else v, device) for k, v in inp.items()}
  out = vtrace.from_importance_weights(**e)
  assert out.vs.shape == (0, B, C)


def test_vtrace_linearity_full_size(device):
  """Size-independent property at cfg4 global size (T=20,B=4096): with rho clipped to
  constants (log-rho = 0) V-trace is linear in (rewards, values, bootstrap)."""
  rng = np.random.default_rng(5)
  T, B = 20, 4096
  z = np.zeros((T, B), np.float32)
  disc = (0.99 * (rng.uniform(size=(T, B)) > 0.05)).astype(np.float32)
  def run(r, v, b):
    return _run_vtrace(device, dict(target_action_log_probs=z, behaviour_action_log_probs=z, discounts=disc,
                                    rewards=r, values=v, bootstrap_value=b))
  r1, v1, b1 = rng.normal(size=(T, B)).astype(np.float32), rng.normal(size=(T, B)).astype(np.float32), rng.normal(size=B).astype(np.float32)
  r2, v2, b2 = rng.normal(size=(T, B)).astype(np.float32), rng.normal(size=(T, B)).astype(np.float32), rng.normal(size=B).astype(np.float32)
  a1, _ = run(r1, v1, b1); a2, _ = run(r2, v2, b2); a3, _ = run(r1 + r2, v1 + v2, b1 + b2)
  assert np.max(np.abs(a3 - (a1 + a2))) < 2e-4


# ------------------------------ categorical ---------------------------------- #
@pytest.mark.parametrize('A', [1, 3, 6, 9, 18, 37, 100])
def test_categorical(device, A):
  from seed_rl_amd import parametric_distribution as pd
  rng = np.random.default_rng(A)
  logits = (rng.normal(size=(7, 5, A)) * 3).astype(np.float32)
  for dt in (np.int32, np.int64):
    act = rng.integers(0, A, (7, 5)).astype(dt)
    d = pd.categorical_distribution(A)
    lp = d.log_prob(dev(logits, device), dev(act, device)).cpu().numpy()
    ent = d.entropy(dev(logits, device)).cpu().numpy()
    np.testing.assert_allclose(lp, categorical_np.log_prob(logits, act), rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(ent, categorical_np.entropy(logits), rtol=1e-5, atol=2e-6)


def test_categorical_reference_golden(device):
  """tests/vtrace_test.py:88-115."""
  from seed_rl_amd import parametric_distribution as pd
  logits = (np.arange(7 * 2 * 3, dtype=np.float32).reshape(7, 2, 3) + 10)
  act = np.random.default_rng(0).integers(0, 2, (7, 2)).astype(np.int32)
  lp = pd.categorical_distribution(3, torch.int32).log_prob(dev(logits, device), dev(act, device)).cpu().numpy()
  sm = np.exp(logits) / np.exp(logits).sum(-1, keepdims=True)
  gt = np.take_along_axis(np.log(sm), act[..., None].astype(np.int64), -1)[..., 0]
  np.testing.assert_allclose(lp, gt, rtol=1e-6, atol=1e-6)


# ------------------------------ loss head ------------------------------------ #
def _run_loss(device, tgt, base, beh, act, rew, done, ld=None, **kw):
  from seed_rl_amd import ops
  T1, B, A = tgt.shape
  T = T1 - 1
  ld = ld or A
  if ld == A:
    logits_d = dev(tgt, device); base_d = dev(base, device); bld = 1
    d_logits = torch.full((T1, B, A), 7.0, device=device); d_base = torch.full((T1, B), 7.0, device=device)
  else:   # head-GEMM layout: [rows, ld] with baseline in column A
    head = np.zeros((T1, B, ld), np.float32); head[..., :A] = tgt; head[..., A] = base
    head_d = dev(head, device)
    logits_d = head_d; base_d = head_d.reshape(-1)[A:]; bld = ld
    d_head = torch.zeros((T1, B, ld), device=device)
    d_logits = d_head; d_base = d_head.reshape(-1)[A:]
  vs = torch.empty((T, B), device=device); pg = torch.empty((T, B), device=device)
  scalars = torch.zeros(16, device=device)
  ws = torch.empty(ops.impala_loss_workspace_bytes(T, B) // 4 + 1, device=device)
  ops.impala_loss_fwd_bwd(logits_d, ld, base_d, bld, dev(beh, device), dev(act, device), dev(rew, device),
                          dev(done.astype(np.uint8), device), T, B, A, d_logits, d_base, scalars, ws, vs, pg, **kw)
  torch.cuda.synchronize()
  if ld == A:
    dl, db = d_logits.cpu().numpy(), d_base.cpu().numpy()
  else:
    dh = d_head.cpu().numpy(); dl, db = dh[..., :A], dh[..., A]
    assert np.all(dh[..., A + 1:] == 0)
  return scalars.cpu().numpy(), dl, db, vs.cpu().numpy(), pg.cpu().numpy()


@pytest.mark.parametrize('seed', [0, 1, 2])
@pytest.mark.parametrize('cfg', [
    dict(T=20, B=32, A=6, kw=dict()),
    dict(T=20, B=32, A=6, kw=dict(lambda_=0.95, kl_cost=0.1, max_abs_reward=1.0, entropy_cost=0.01)),
    dict(T=20, B=37, A=18, kw=dict(), adt=np.int32),
    dict(T=5, B=3, A=9, kw=dict(lambda_=0.95)),
    dict(T=100, B=16, A=18, kw=dict()),
])
def test_loss_head_parity(device, seed, cfg):
  """V-trace outputs <= 1e-5 (north_star bar); loss scalars / gradients fp32-rounding close."""
  tgt, base, beh, act, rew, done = synth.loss_inputs(seed, cfg['T'], cfg['B'], cfg['A'], cfg.get('adt', np.int64))
  ref = loss_np.compute_loss_from_outputs(tgt, base, beh, act, rew, done, **cfg['kw'])
  for ld in (None, ((cfg['A'] + 1 + 3) // 4) * 4):
    sc, dl, db, vs, pg = _run_loss(device, tgt, base, beh, act, rew, done, ld=ld, **cfg['kw'])
    assert np.max(np.abs(vs - ref.vs)) <= 1e-5
    assert np.max(np.abs(pg - ref.pg_advantages)) <= 1e-5
    for i, name in enumerate(['total_loss', 'policy_loss', 'v_loss', 'entropy_loss', 'kl_loss', 'entropy',
                              'kl_mean', 'value_mean', 'v_l2_error']):
      r = float(getattr(ref, name))
      assert abs(sc[i] - r) <= 2e-5 * max(1.0, abs(r)), (name, sc[i], r)
    assert sc[9] == ref.max_action_abs
    np.testing.assert_allclose(dl, ref.d_policy_logits, rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(db, ref.d_baseline, rtol=1e-4, atol=1e-7)


def test_loss_head_mean_denominator_shards(device):
  """Data-parallel mean semantics: two column shards with the GLOBAL N sum to the
  single-batch result."""
  tgt, base, beh, act, rew, done = synth.loss_inputs(7, 20, 64, 6)
  full = _run_loss(device, tgt, base, beh, act, rew, done)
  n = 20 * 64
  parts = [_run_loss(device, tgt[:, s], base[:, s], beh[:, s], act[:, s], rew[:, s], done[:, s],
                     mean_denominator=n) for s in (slice(0, 32), slice(32, 64))]
  assert abs(parts[0][0][0] + parts[1][0][0] - full[0][0]) < 1e-5
  np.testing.assert_allclose(np.concatenate([parts[0][1], parts[1][1]], 1), full[1], rtol=1e-5, atol=1e-8)


# ------------------------------ Adam ------------------------------------------ #
def test_adam_flat_matches_keras_restatement(device):
  from seed_rl_amd import ops
  rng = np.random.default_rng(0)
  n = 100003
  p0 = rng.normal(size=n).astype(np.float32)
  for b1, eps in ((0.9, 1e-7), (0.0, 3.125e-7)):
    p = torch.tensor(p0.copy())
    opt = nets_torch.KerasAdam([p], nets_torch.polynomial_decay(4.8e-4, 1000), beta_1=b1, epsilon=eps)
    pd_ = dev(p0, device).clone(); m = torch.zeros(n, device=device); v = torch.zeros(n, device=device)
    for it in range(3):
      g = rng.normal(size=n).astype(np.float32)
      lr = nets_torch.polynomial_decay(4.8e-4, 1000)(it)
      t = it + 1
      lr_t = lr * np.sqrt(1 - 0.999 ** t) / (1 - b1 ** t)
      ops.adam_flat(pd_, dev(g, device), m, v, float(lr_t), b1, 0.999, eps, 1.0)
      opt.apply_gradients([torch.tensor(g)])
    np.testing.assert_allclose(pd_.cpu().numpy(), p.numpy(), rtol=2e-6, atol=2e-7)


def test_clip_by_global_norm(device):
  from seed_rl_amd import ops
  rng = np.random.default_rng(1)
  g0 = rng.normal(size=300001).astype(np.float32)
  for clip in (40.0, 1000.0):
    g = dev(g0, device).clone()
    ss = torch.zeros(1, device=device)
    ws = torch.empty(ops.global_norm_workspace_bytes() // 4, device=device)
    ops.clip_by_global_norm(g, clip, ss, ws)
    norm = np.sqrt((g0.astype(np.float64) ** 2).sum())
    assert abs(np.sqrt(float(ss[0])) - norm) < 1e-3 * norm
    np.testing.assert_allclose(g.cpu().numpy(), g0 * (clip / max(norm, clip)), rtol=1e-5)


# ------------------------------ frame stacking -------------------------------- #
@pytest.mark.parametrize('T,B,H,W', [(6, 3, 5, 4), (21, 4, 84, 84), (1, 2, 7, 3)])
def test_stack_frames_parity(device, T, B, H, W):
  """Bit-exact vs oracle stack_frames (atari/networks.py:57-173), incl. the new state."""
  from seed_rl_amd import ops
  rng = np.random.default_rng(T * B)
  frames = rng.integers(0, 256, (T, B, H, W, 1)).astype(np.uint8)
  done = rng.uniform(size=(T, B)) < 0.3
  state = rng.integers(0, 2 ** 24, (B, H * W)).astype(np.int32)
  ref, ref_state = frames_np.stack_frames(frames, state, done, 4)
  HW = H * W
  ext = torch.zeros((T + 3, B, HW), dtype=torch.uint8, device=device)
  ext[3:] = dev(frames.reshape(T, B, HW), device)
  nv = torch.zeros((T, B), dtype=torch.uint8, device=device)
  ops.stack_prepare(dev(state, device), dev(done.astype(np.uint8), device), T, B, HW, ext, nv)
  out = torch.empty((T, B, HW, 4), device=device)
  ops.stack_frames_f32(ext, nv, T, B, HW, out)
  new_state = torch.empty((B, HW), dtype=torch.int32, device=device)
  ops.stack_pack_state(ext, nv, T, B, HW, new_state)
  np.testing.assert_array_equal(out.cpu().numpy().reshape(ref.shape), ref)
  np.testing.assert_array_equal(new_state.cpu().numpy(), ref_state)


def test_stack_frames_reference_golden(device):
  """atari/networks_test.py:186-247 sequences (incl. done in the middle)."""
  from seed_rl_amd import ops
  def run(frames, done, state):
    T = len(frames)
    ext = torch.zeros((T + 3, 1, 1), dtype=torch.uint8, device=device)
    ext[3:, 0, 0] = torch.tensor(frames, dtype=torch.uint8)
    nv = torch.zeros((T, 1), dtype=torch.uint8, device=device)
    ops.stack_prepare(state, torch.tensor(done, dtype=torch.uint8, device=device).reshape(T, 1), T, 1, 1, ext, nv)
    out = torch.empty((T, 1, 1, 4), device=device)
    ops.stack_frames_f32(ext, nv, T, 1, 1, out)
    ns = torch.empty((1, 1), dtype=torch.int32, device=device)
    ops.stack_pack_state(ext, nv, T, 1, 1, ns)
    return out.cpu().numpy().reshape(T, 4), ns
  st = torch.zeros((1, 1), dtype=torch.int32, device=device)
  o, st = run([1], [0], st); assert o.tolist() == [[1, 0, 0, 0]]
  o, st2 = run([2], [0], st); assert o.tolist() == [[2, 1, 0, 0]]
  o, _ = run([3, 4, 5, 6, 7, 8], [0] * 6, st2)
  assert o[0].tolist() == [3, 2, 1, 0] and o[5].tolist() == [8, 7, 6, 5]
  o, st3 = run([2], [1], st); assert o.tolist() == [[2, 0, 0, 0]]
  o, _ = run([3, 4, 5, 6, 7, 8], [0, 0, 0, 0, 1, 0], st3)
  assert o[0].tolist() == [3, 2, 0, 0] and o[5].tolist() == [8, 7, 0, 0]


# ------------------------------ conv / dense ---------------------------------- #
CONV_CASES = [
    # n, ih, iw, cin, kh, kw, stride, padding, cout      (the layer shapes of the three agents)
    (3, 72, 96, 3, 3, 3, 1, 'same', 16),      # ImpalaDeep stack0 conv (u8 input)
    (3, 36, 48, 16, 3, 3, 1, 'same', 16),     # stack0 res conv
    (3, 36, 48, 16, 3, 3, 1, 'same', 32),     # stack1 conv
    (5, 18, 24, 32, 3, 3, 1, 'same', 32),     # stack1/2 res conv
    (7, 20, 20, 16, 4, 4, 2, 'valid', 32),    # shallow conv2
    (4, 20, 20, 32, 4, 4, 2, 'valid', 64),    # DQN conv2
    (4, 9, 9, 64, 3, 3, 1, 'valid', 64),      # DQN conv3
    (300, 1, 1, 2592, 1, 1, 1, 'valid', 256), # shallow FC
    (300, 1, 1, 256, 1, 1, 1, 'valid', 20),   # heads (A=18 -> ld 20)
    (70, 1, 1, 268, 1, 1, 1, 'valid', 1024),  # LSTM input projection
    (3, 11, 9, 8, 5, 3, 2, 'valid', 12),      # odd shape
    (2, 10, 14, 64, 3, 3, 1, 'valid', 64),    # halo wgrad, 4-way row split (ow % 4 == 0)
    (3, 13, 20, 32, 3, 3, 1, 'same', 16),     # halo wgrad, 2-way row split, partial last band
    (70, 9, 12, 32, 3, 3, 1, 'same', 32),     # ImpalaDeep stack2 res conv, many images per workgroup
    (64, 1, 1, 2592, 1, 1, 1, 'valid', 256),  # inference-batch FC: split-K forward / data gradient
    (256, 1, 1, 512, 1, 1, 1, 'valid', 2048), # LSTM(512) recurrent step GEMM: split-K
    (3, 11, 13, 8, 3, 3, 2, 'valid', 16),     # stride-2 data gradient: parity classes with different tap counts
    (2, 12, 10, 16, 4, 3, 2, 'valid', 32),    # 4x3 kernel, stride 2
    (2100, 1, 1, 268, 1, 1, 1, 'valid', 1024),# gemm.h: K % 32 != 0, ragged last row tile, 128-wide tiles
    (2500, 1, 1, 256, 1, 1, 1, 'valid', 20),  # gemm.h: N = 20 < tile, split-K
    (2060, 1, 1, 2592, 1, 1, 1, 'valid', 256),# gemm.h at the shallow FC shape
    (300, 1, 1, 256, 1, 1, 1, 'valid', 18),   # cout % 4 != 0: Dense accessors of the implicit-GEMM core
    (37, 1, 1, 36, 1, 1, 1, 'valid', 44),     # everything ragged
    (1500, 20, 20, 16, 4, 4, 2, 'valid', 32), # wsgemm.h: persistent workgroups walk several m-tiles
    (37, 19, 21, 16, 4, 4, 2, 'valid', 32),   # wsgemm.h specialised kernel, odd map: super-pixels hang over dX and dY, ragged last tile
    (21, 20, 22, 8, 4, 4, 2, 'valid', 64),    # wsgemm.h specialised kernel: 64 output channels forward, 32-column data gradient (NKT = 4 / 8)
    (3, 10, 12, 16, 3, 3, 1, 'same', 64),     # gather-GEMM forward with 'same' padding (taps predicated at the border)
    (2, 8, 9, 64, 3, 3, 1, 'same', 64),       # gather-GEMM forward + data gradient with padding
    (2, 11, 9, 64, 5, 5, 1, 'same', 128),     # 5x5: 25 taps, K = 1600
    (3, 13, 11, 16, 4, 4, 2, 'valid', 64),    # stride 2, odd map: super-pixels past the dY border
    (2, 9, 7, 4, 3, 3, 1, 'same', 64),        # gather-GEMM, 4 channels per tap: 8 taps per k-tile, K = 36 (ragged)
    (2, 12, 9, 8, 5, 3, 1, 'valid', 72),      # 8 channels per tap, non-square kernel, cout % 16 != 0
    (3, 14, 14, 16, 6, 6, 3, 'valid', 64),    # stride 3: nine parity classes in the super-pixel data gradient
    (2, 7, 5, 128, 1, 1, 1, 'valid', 64),     # 1x1 conv on a map (not Dense: ih, iw > 1)
    (1, 6, 6, 32, 2, 2, 2, 'valid', 256),     # 2x2/2 (non-overlapping), wide output
    (2, 8, 8, 64, 3, 3, 1, 'same', 192),      # cout = 3 column tiles of 64
    (2, 17, 50, 32, 3, 3, 1, 'same', 32),     # halo kernels, second tiling budget (80 KB / 10 prefetch vectors), odd map
    (3, 10, 48, 32, 3, 3, 1, 'same', 16),     # 32 -> 16 on 48-pixel rows: forward of the shape ImpalaDeep has as a data gradient
    (5, 9, 12, 16, 3, 3, 1, 'same', 32),      # weight gradient: one band per 9x12 image (6 dY prefetch vectors)
    (130, 9, 9, 64, 3, 3, 1, 'valid', 64),    # data gradient in image-block x position order: two blocks + a padded third, taps skipped at the border
    (70, 20, 20, 32, 4, 4, 2, 'valid', 64),   # the same for the stride-2 super-pixel GEMM (DQN conv2)
    (2100, 1, 1, 68, 1, 1, 1, 'valid', 132),  # xgemm.h (bf16x6; served from 2048 rows): ragged in M, N and K at once
    (2049, 1, 1, 100, 1, 1, 1, 'valid', 128), # xgemm.h: one row past a tile, K = 3 k-tiles + 4
    (2112, 1, 1, 4100, 1, 1, 1, 'valid', 384),# xgemm.h: split-K forward / data gradient with a ragged last slice
    (2050, 20, 20, 16, 4, 4, 2, 'valid', 32), # wfw.h (image-resident forward, from 2048 images): 5 images per workgroup, ragged last tile
    (2100, 19, 21, 16, 4, 4, 2, 'valid', 32), # wfw.h: odd map (72 pixels per image), slot rounded up to a KB multiple
]


@pytest.mark.parametrize('case', CONV_CASES)
def test_conv_fwd_bwd_parity(device, case):
  """fp32 MFMA conv/dense fwd, dgrad, wgrad vs torch-CPU fp32 (oracle conv2d).
  Tolerance: 2e-4 relative to the output scale (fp32 accumulation-order noise, K<=2592)."""
  from seed_rl_amd import ops
  n, ih, iw, cin, kh, kw, stride, padding, cout = case
  rng = np.random.default_rng(abs(hash(case)) % 1000)
  u8 = cin == 3
  if u8:
    x_raw = rng.integers(0, 256, (n, ih, iw, cin)).astype(np.uint8)
    x = torch.tensor(x_raw).float() / 255
  else:
    x_raw = rng.normal(size=(n, ih, iw, cin)).astype(np.float32)
    x = torch.tensor(x_raw)
  w = (rng.normal(size=(kh, kw, cin, cout)) / np.sqrt(kh * kw * cin)).astype(np.float32)
  b = rng.normal(size=(cout,)).astype(np.float32)
  in_relu = not u8
  x.requires_grad_(True)
  wt = torch.tensor(w, requires_grad=True); bt = torch.tensor(b, requires_grad=True)
  y = F.relu(nets_torch.conv2d(F.relu(x) if in_relu else x, wt, bt, stride, padding))
  dy = rng.normal(size=y.shape).astype(np.float32)
  y.backward(torch.tensor(dy))
  yr = y.detach().numpy()

  g = ops.conv_geom(n, ih, iw, cin, kh, kw, stride, padding, cout)
  xd = dev(x_raw, device); wd = dev(w, device); bd = dev(b, device)
  out = torch.full((n, g.oh, g.ow, cout), 7.0, device=device)
  ops.conv2d_fwd(g, xd, wd, bd, out, in_dtype=ops.IN_U8_DIV255 if u8 else ops.IN_F32, in_relu=in_relu, out_relu=True)
  tol = 2e-4 * max(1.0, np.abs(yr).max())
  assert np.max(np.abs(out.cpu().numpy() - yr)) <= tol

  dz = dev(np.ascontiguousarray(dy * (yr > 0), np.float32), device)
  dw = torch.full(w.shape, 7.0, device=device); db = torch.full(b.shape, 7.0, device=device)
  ws = torch.empty(ops.conv2d_bwd_weight_workspace_bytes(g) // 4 + 1, device=device)
  ops.conv2d_bwd_weight(g, xd, dz, dw, db, ws, in_dtype=ops.IN_U8_DIV255 if u8 else ops.IN_F32, in_relu=in_relu)
  gw = wt.grad.numpy(); gb = bt.grad.numpy()
  assert np.max(np.abs(dw.cpu().numpy() - gw)) <= 3e-4 * max(1.0, np.abs(gw).max())
  assert np.max(np.abs(db.cpu().numpy() - gb)) <= 3e-4 * max(1.0, np.abs(gb).max())
  if not u8:
    dx = torch.full(x_raw.shape, 7.0, device=device)
    ops.conv2d_bwd_data(g, dz, wd, dx, relu_mask=xd)
    gx = x.grad.numpy()
    assert np.max(np.abs(dx.cpu().numpy() - gx)) <= 2e-4 * max(1.0, np.abs(gx).max())


@pytest.mark.parametrize('n,cin,cout', [(300, 531, 2048), (2100, 265, 1024), (64, 19, 36)])
def test_dense_padded_rows(device, n, cin, cout):
  """Dense layers whose input rows carry pad columns (the LSTM input [features | reward | one-hot action] has
  512 + 1 + 18 = 531 columns in rows of 532): forward / weight gradient / data gradient through gemm.h with
  K % 4 != 0, against torch fp32.  The pad column holds finite junk."""
  from seed_rl_amd import ops
  rng = np.random.default_rng(n + cin)
  ld = (cin + 3) // 4 * 4
  xp = rng.normal(size=(n, ld)).astype(np.float32)            # pad columns: finite junk
  x = xp[:, :cin]
  w = (rng.normal(size=(cin, cout)) / np.sqrt(cin)).astype(np.float32)
  b = rng.normal(size=(cout,)).astype(np.float32)
  xt = torch.tensor(x.copy(), requires_grad=True); wt = torch.tensor(w, requires_grad=True)
  bt = torch.tensor(b, requires_grad=True)
  y = xt @ wt + bt
  dy = rng.normal(size=y.shape).astype(np.float32)
  y.backward(torch.tensor(dy))
  g = ops.dense_geom(n, cin, cout, ld_in=ld)
  xd, wd, bd, dyd = dev(xp, device), dev(w, device), dev(b, device), dev(dy, device)
  out = torch.full((n, cout), 7.0, device=device)
  ops.conv2d_fwd(g, xd, wd, bd, out)
  yr = y.detach().numpy()
  assert np.max(np.abs(out.cpu().numpy() - yr)) <= 2e-4 * max(1.0, np.abs(yr).max())
  dw = torch.full(w.shape, 7.0, device=device); db = torch.full(b.shape, 7.0, device=device)
  ws = torch.empty(ops.conv2d_bwd_weight_workspace_bytes(g) // 4 + 1, device=device)
  ops.conv2d_bwd_weight(g, xd, dyd, dw, db, ws)
  gw, gb = wt.grad.numpy(), bt.grad.numpy()
  assert np.max(np.abs(dw.cpu().numpy() - gw)) <= 3e-4 * max(1.0, np.abs(gw).max())
  assert np.max(np.abs(db.cpu().numpy() - gb)) <= 3e-4 * max(1.0, np.abs(gb).max())
  dx = torch.full((n, ld), 7.0, device=device)
  ops.conv2d_bwd_data(g, dyd, wd, dx)
  gx = xt.grad.numpy()
  assert np.max(np.abs(dx.cpu().numpy()[:, :cin] - gx)) <= 2e-4 * max(1.0, np.abs(gx).max())


def test_conv_residual_and_accumulate(device):
  """Residual epilogue (dmlab/networks.py:58) and dgrad accumulate-into (skip path)."""
  from seed_rl_amd import ops
  rng = np.random.default_rng(0)
  n, h, w_, c = 2, 9, 12, 32
  x = rng.normal(size=(n, h, w_, c)).astype(np.float32)
  k = (rng.normal(size=(3, 3, c, c)) * 0.05).astype(np.float32)
  b = rng.normal(size=c).astype(np.float32)
  res = rng.normal(size=(n, h, w_, c)).astype(np.float32)
  ref = nets_torch.conv2d(torch.tensor(x), torch.tensor(k), torch.tensor(b), 1, 'same') + torch.tensor(res)
  g = ops.conv_geom(n, h, w_, c, 3, 3, 1, 'same', c)
  out = torch.empty((n, h, w_, c), device=device)
  ops.conv2d_fwd(g, dev(x, device), dev(k, device), dev(b, device), out, residual=dev(res, device))
  assert np.max(np.abs(out.cpu().numpy() - ref.numpy())) < 2e-4
  dy = rng.normal(size=(n, h, w_, c)).astype(np.float32)
  add = rng.normal(size=(n, h, w_, c)).astype(np.float32)
  xt = torch.tensor(x, requires_grad=True)
  nets_torch.conv2d(xt, torch.tensor(k), None, 1, 'same').backward(torch.tensor(dy))
  dx = torch.empty((n, h, w_, c), device=device)
  ops.conv2d_bwd_data(g, dev(dy, device), dev(k, device), dx, add=dev(add, device))
  assert np.max(np.abs(dx.cpu().numpy() - (xt.grad.numpy() + add))) < 2e-4


@pytest.mark.parametrize('T1,B,cout', [(5, 3, 16), (21, 8, 16), (21, 37, 32), (4, 2, 8), (1, 64, 16)])
def test_stack_conv_parity(device, T1, B, cout):
  """Fused stack_frames + /255 + conv1 (u8 frames in, never materialising the fp32
  stacked tensor) vs oracle stack_frames -> conv2d, forward and weight gradient."""
  from seed_rl_amd import ops
  u = synth.atari_unroll(11, T1, B, done_p=0.15, zero_state=False)
  stacked, _ = frames_np.stack_frames(u['frames'], u['frame_state'], u['done'], 4)
  rng = np.random.default_rng(0)
  # cout % 16 == 0 -> dedicated LDS-ring kernel (csrc/stackconv.hip); cout = 8 -> generic implicit-GEMM path
  w = (rng.normal(size=(8, 8, 4, cout)) / 16).astype(np.float32)
  b = rng.normal(size=cout).astype(np.float32)
  wt = torch.tensor(w, requires_grad=True); bt = torch.tensor(b, requires_grad=True)
  y = F.relu(nets_torch.conv2d(torch.tensor(stacked / np.float32(255)).reshape(T1 * B, 84, 84, 4), wt, bt, 4, 'valid'))
  dy = rng.normal(size=y.shape).astype(np.float32)
  y.backward(torch.tensor(dy))
  HW = 84 * 84
  ext = torch.zeros((T1 + 3, B, HW), dtype=torch.uint8, device=device)
  ext[3:] = dev(u['frames'].reshape(T1, B, HW), device)
  nv = torch.zeros((T1, B), dtype=torch.uint8, device=device)
  ops.stack_prepare(dev(u['frame_state'], device), dev(u['done'].astype(np.uint8), device), T1, B, HW, ext, nv)
  g = ops.StackConvGeom(T1, B, 84, 84, 20, 20, 8, 8, 4, cout, cout)
  out = torch.empty((T1 * B, 20, 20, cout), device=device)
  ops.conv2d_stack_fwd(g, ext, nv, dev(w, device), dev(b, device), out, out_relu=True)
  yr = y.detach().numpy()
  assert np.max(np.abs(out.cpu().numpy() - yr)) <= 2e-4 * max(1.0, np.abs(yr).max())
  dz = dev(np.ascontiguousarray(dy * (yr > 0), np.float32), device)
  dw = torch.empty(w.shape, device=device); db = torch.empty(b.shape, device=device)
  ws = torch.empty(ops.conv2d_stack_bwd_weight_workspace_bytes(g) // 4 + 1, device=device)
  ops.conv2d_stack_bwd_weight(g, ext, nv, dz, dw, db, ws)
  gw = wt.grad.numpy()
  assert np.max(np.abs(dw.cpu().numpy() - gw)) <= 3e-4 * max(1.0, np.abs(gw).max())
  assert np.max(np.abs(db.cpu().numpy() - bt.grad.numpy())) <= 3e-4 * max(1.0, np.abs(bt.grad.numpy()).max())


def test_stack_conv_fwd_fp32_accuracy(device):
  """The bf16x3 forward (exact operand split, fp32 accumulate on the bf16 matrix pipe) must be as close to the
  fp64 ground truth as an fp32 evaluation is: its max error may not exceed 2x that of torch's fp32 conv on the
  same inputs (both are fp32-summation-order noise, ~1e-6 of the output scale)."""
  from seed_rl_amd import ops
  T1, B, cout = 6, 5, 16
  u = synth.atari_unroll(3, T1, B, done_p=0.1, zero_state=False)
  stacked, _ = frames_np.stack_frames(u['frames'], u['frame_state'], u['done'], 4)
  rng = np.random.default_rng(1)
  w = (rng.normal(size=(8, 8, 4, cout)) / 16).astype(np.float32)
  b = rng.normal(size=cout).astype(np.float32)
  x32 = torch.tensor(stacked / np.float32(255)).reshape(T1 * B, 84, 84, 4)
  y32 = nets_torch.conv2d(x32, torch.tensor(w), torch.tensor(b), 4, 'valid').numpy()
  y64 = nets_torch.conv2d(torch.tensor(stacked.astype(np.float64) / 255.0).reshape(T1 * B, 84, 84, 4),
                          torch.tensor(w.astype(np.float64)), torch.tensor(b.astype(np.float64)), 4, 'valid').numpy()
  HW = 84 * 84
  ext = torch.zeros((T1 + 3, B, HW), dtype=torch.uint8, device=device)
  ext[3:] = dev(u['frames'].reshape(T1, B, HW), device)
  nv = torch.zeros((T1, B), dtype=torch.uint8, device=device)
  ops.stack_prepare(dev(u['frame_state'], device), dev(u['done'].astype(np.uint8), device), T1, B, HW, ext, nv)
  g = ops.StackConvGeom(T1, B, 84, 84, 20, 20, 8, 8, 4, cout, cout)
  out = torch.empty((T1 * B, 20, 20, cout), device=device)
  ops.conv2d_stack_fwd(g, ext, nv, dev(w, device), dev(b, device), out, out_relu=False)
  err_hip = np.max(np.abs(out.cpu().numpy().astype(np.float64) - y64))
  err_f32 = np.max(np.abs(y32.astype(np.float64) - y64))
  scale = np.abs(y64).max()
  assert err_hip <= max(2.0 * err_f32, 2e-6 * scale), (err_hip, err_f32, scale)

  # weight gradient: dW = X^T dY with dY split exactly into three bf16 parts
  dy = rng.normal(size=y32.shape).astype(np.float32)
  x64 = torch.tensor(stacked.astype(np.float64) / 255.0).reshape(T1 * B, 84, 84, 4)
  w64 = torch.tensor(w.astype(np.float64), requires_grad=True)
  nets_torch.conv2d(x64, w64, None, 4, 'valid').backward(torch.tensor(dy.astype(np.float64)))
  w32 = torch.tensor(w, requires_grad=True)
  nets_torch.conv2d(x32, w32, None, 4, 'valid').backward(torch.tensor(dy))
  dw = torch.empty(w.shape, device=device); db = torch.empty(b.shape, device=device)
  ws = torch.empty(ops.conv2d_stack_bwd_weight_workspace_bytes(g) // 4 + 1, device=device)
  ops.conv2d_stack_bwd_weight(g, ext, nv, dev(dy, device), dw, db, ws)
  g64 = w64.grad.numpy()
  e_hip = np.max(np.abs(dw.cpu().numpy().astype(np.float64) - g64))
  e_f32 = np.max(np.abs(w32.grad.numpy().astype(np.float64) - g64))
  assert e_hip <= max(2.0 * e_f32, 2e-6 * np.abs(g64).max()), (e_hip, e_f32, np.abs(g64).max())


@pytest.mark.parametrize('n,cin,cout', [(2048, 2592, 256), (2304, 3136, 512)])   # (>= 2048 rows: below, gemm.h serves)
def test_x6_gemm_fp32_accuracy(device, n, cin, cout):
  """The Dense kernels of xgemm.h evaluate fp32 x fp32 products on the bf16 matrix pipe through the exact three-way
  operand split, keeping the six partial products above 2^-25 of a product.  They must be as close to an fp64
  evaluation as an fp32 evaluation is: max error <= 2x that of torch's fp32 matmul on the same inputs (forward, data
  gradient, weight gradient; K = 2592 / 3136 as in the Atari agents, reduction over n rows for the weight gradient)."""
  from seed_rl_amd import ops
  rng = np.random.default_rng(cin)
  x = rng.normal(size=(n, cin)).astype(np.float32)
  w = (rng.normal(size=(cin, cout)) / np.sqrt(cin)).astype(np.float32)
  b = rng.normal(size=cout).astype(np.float32)
  dy = rng.normal(size=(n, cout)).astype(np.float32)
  x64, w64, dy64 = x.astype(np.float64), w.astype(np.float64), dy.astype(np.float64)
  tx, tw, tdy = torch.tensor(x), torch.tensor(w), torch.tensor(dy)
  g = ops.dense_geom(n, cin, cout)
  xd, wd, bd, dyd = dev(x, device), dev(w, device), dev(b, device), dev(dy, device)

  def check(name, hip, f32, f64):
    e_hip = np.max(np.abs(hip.astype(np.float64) - f64))
    e_f32 = np.max(np.abs(f32.astype(np.float64) - f64))
    scale = np.abs(f64).max()
    print('x6 %s n=%d cin=%d cout=%d: err hip %.3e  torch fp32 %.3e  scale %.3e' % (name, n, cin, cout, e_hip, e_f32, scale))
    assert e_hip <= max(2.0 * e_f32, 2e-6 * scale), (name, e_hip, e_f32, scale)

  out = torch.empty((n, cout), device=device)
  ops.conv2d_fwd(g, xd, wd, bd, out)
  check('fwd', out.cpu().numpy(), (tx @ tw + torch.tensor(b)).numpy(), x64 @ w64 + b.astype(np.float64))
  dx = torch.empty((n, cin), device=device)
  ops.conv2d_bwd_data(g, dyd, wd, dx)
  check('dgrad', dx.cpu().numpy(), (tdy @ tw.T).numpy(), dy64 @ w64.T)
  dw = torch.empty((cin, cout), device=device); db = torch.empty(cout, device=device)
  ws = torch.empty(ops.conv2d_bwd_weight_workspace_bytes(g) // 4 + 4, device=device)
  ops.conv2d_bwd_weight(g, xd, dyd, dw, db, ws)
  check('wgrad', dw.cpu().numpy(), (tx.T @ tdy).numpy(), x64.T @ dy64)
  check('bias grad', db.cpu().numpy(), tdy.sum(0).numpy(), dy64.sum(0))


@pytest.mark.parametrize('n,in_relu,out_relu,bias', [(256, False, True, True), (300, True, True, False), (1029, False, True, True),
                                                     (2048, False, True, True), (2049, True, True, True), (2309, False, False, False),
                                                     (8448, False, True, True)])
def test_wfx_conv2_forward_fp32_accuracy(device, n, in_relu, out_relu, bias):
  """The second Atari conv's forward on the bf16 matrix pipe (wfx.h: exact three-way split of activations and weights,
  six plane products, rows of the run staged once into an LDS ring).  As close to an fp64 evaluation as torch's fp32
  convolution is (max error <= 2x), for runs of 1 / 2 / 5 / 8 / 9 / 10 / 33 images per workgroup with ragged last
  workgroups and dead lanes in the last round; and bit-identical from call to call."""
  from seed_rl_amd import ops
  rng = np.random.default_rng(n)
  x = rng.normal(size=(n, 20, 20, 16)).astype(np.float32)
  w = (rng.normal(size=(4, 4, 16, 32)) / 16).astype(np.float32)
  b = rng.normal(size=32).astype(np.float32) if bias else None
  tx, tw = torch.tensor(x).permute(0, 3, 1, 2), torch.tensor(w).permute(3, 2, 0, 1)
  if in_relu: tx = F.relu(tx)
  y32 = F.conv2d(tx, tw, torch.tensor(b) if bias else None, stride=2).permute(0, 2, 3, 1)
  y64 = F.conv2d(tx.double(), tw.double(), torch.tensor(b).double() if bias else None, stride=2).permute(0, 2, 3, 1)
  if out_relu: y32, y64 = F.relu(y32), F.relu(y64)
  g = ops.conv_geom(n, 20, 20, 16, 4, 4, 2, 'valid', 32)
  xd, wd = dev(x, device), dev(w, device)
  bd = dev(b, device) if bias else None
  out = torch.full((n, 9, 9, 32), 7.0, device=device)
  ops.conv2d_fwd(g, xd, wd, bd, out, in_relu=in_relu, out_relu=out_relu)
  got = out.cpu().numpy().astype(np.float64)
  e_hip = np.max(np.abs(got - y64.numpy())); e_f32 = np.max(np.abs(y32.numpy().astype(np.float64) - y64.numpy()))
  print('wfx n=%d: err hip %.3e  torch fp32 %.3e' % (n, e_hip, e_f32))
  assert e_hip <= max(2.0 * e_f32, 2e-6 * np.abs(y64.numpy()).max()), (e_hip, e_f32)
  out2 = torch.full((n, 9, 9, 32), -3.0, device=device)
  ops.conv2d_fwd(g, xd, wd, bd, out2, in_relu=in_relu, out_relu=out_relu)
  assert torch.equal(out, out2)


@pytest.mark.parametrize('n,mask', [(256, True), (300, False), (1029, True), (2048, True), (2049, True), (2309, False), (8448, True)])
def test_wdx_conv2_data_gradient_fp32_accuracy(device, n, mask):
  """The second Atari conv's data gradient on the bf16 matrix pipe (wdx.h: super-pixel GEMM, dY staged once as padded
  rows, ReLU mask by LDS-DMA).  As close to an fp64 evaluation as torch's fp32 gradient is (max error <= 2x), masked
  positions exactly zero, bit-identical from call to call; runs of 1 / 2 / 5 / 8 / 9 / 10 / 33 images per workgroup."""
  from seed_rl_amd import ops
  rng = np.random.default_rng(n + 1)
  x = rng.normal(size=(n, 20, 20, 16)).astype(np.float32)
  w = (rng.normal(size=(4, 4, 16, 32)) / 16).astype(np.float32)
  dy = rng.normal(size=(n, 9, 9, 32)).astype(np.float32)
  tw = torch.tensor(w).permute(3, 2, 0, 1)
  tdy = torch.tensor(dy).permute(0, 3, 1, 2)
  g32 = F.conv_transpose2d(tdy, tw, stride=2).permute(0, 2, 3, 1)
  g64 = F.conv_transpose2d(tdy.double(), tw.double(), stride=2).permute(0, 2, 3, 1)
  if mask:
    keep = torch.tensor(x > 0)
    g32, g64 = g32 * keep, g64 * keep
  g = ops.conv_geom(n, 20, 20, 16, 4, 4, 2, 'valid', 32)
  xd, wd, dyd = dev(x, device), dev(w, device), dev(dy, device)
  dx = torch.full((n, 20, 20, 16), 7.0, device=device)
  ops.conv2d_bwd_data(g, dyd, wd, dx, relu_mask=xd if mask else None)
  got = dx.cpu().numpy().astype(np.float64)
  e_hip = np.max(np.abs(got - g64.numpy())); e_f32 = np.max(np.abs(g32.numpy().astype(np.float64) - g64.numpy()))
  print('wdx n=%d: err hip %.3e  torch fp32 %.3e' % (n, e_hip, e_f32))
  assert e_hip <= max(2.0 * e_f32, 2e-6 * np.abs(g64.numpy()).max()), (e_hip, e_f32)
  if mask:
    assert np.all(got[x <= 0] == 0.0)
  dx2 = torch.full((n, 20, 20, 16), -3.0, device=device)
  ops.conv2d_bwd_data(g, dyd, wd, dx2, relu_mask=xd if mask else None)
  assert torch.equal(dx, dx2)


@pytest.mark.parametrize('n,h,w,mode', [(512, 18, 24, 'fwd_res'), (600, 18, 24, 'fwd_plain'), (700, 9, 12, 'fwd_res'),
                                        (515, 18, 24, 'dg_mask_add'), (640, 9, 12, 'dg_mask'), (1300, 9, 12, 'dg_plain'),
                                        (256, 36, 48, 'fwd_res'), (300, 36, 48, 'fwd_plain'), (259, 36, 48, 'dg_mask_add'),
                                        (513, 36, 48, 'dg_mask'), (270, 36, 48, 'dg_plain')])
def test_wsx_conv3x3_fp32_accuracy(device, n, h, w, mode):
  """ImpalaDeep's 32 -> 32 (18 x 24, 9 x 12 maps) and 16 -> 16 (36 x 48) 3x3 'same' layers on the bf16 matrix pipe
  (wsx.h: padded rows staged once, K split between the two waves of a tile, epilogue operands by LDS-DMA; wsy.h: all
  weights per wave, 16 x 16 x 32 MFMAs, outputs from the registers): forward with ReLU on the input, bias, residual and without;
  data gradient (weights flipped / transposed in the kernel) with ReLU mask and skip-path add and without.  As close to
  an fp64 evaluation as torch's fp32 convolution is (<= 2x), bit-identical from call to call; ragged last workgroups."""
  from seed_rl_amd import ops
  rng = np.random.default_rng(n + h)
  C = 16 if h == 36 else 32
  x = rng.normal(size=(n, h, w, C)).astype(np.float32)
  wt = (rng.normal(size=(3, 3, C, C)) / np.sqrt(9 * C)).astype(np.float32)
  b = rng.normal(size=C).astype(np.float32)
  extra = rng.normal(size=(n, h, w, C)).astype(np.float32)
  extra2 = rng.normal(size=(n, h, w, C)).astype(np.float32)
  g = ops.conv_geom(n, h, w, C, 3, 3, 1, 'same', C)
  xd, wd, bd, ed, e2d = dev(x, device), dev(wt, device), dev(b, device), dev(extra, device), dev(extra2, device)
  tx = torch.tensor(x).permute(0, 3, 1, 2)
  tw = torch.tensor(wt).permute(3, 2, 0, 1)               # [co, ci, kh, kw]
  te, te2 = torch.tensor(extra), torch.tensor(extra2)

  def run():
    out = torch.full((n, h, w, C), 7.0, device=device)
    if mode == 'fwd_res':
      ops.conv2d_fwd(g, xd, wd, bd, out, in_relu=True, out_relu=False, residual=ed)
    elif mode == 'fwd_plain':
      ops.conv2d_fwd(g, xd, wd, None, out, in_relu=False, out_relu=True)
    elif mode == 'dg_mask_add':
      ops.conv2d_bwd_data(g, xd, wd, out, relu_mask=ed, add=e2d)
    elif mode == 'dg_mask':
      ops.conv2d_bwd_data(g, xd, wd, out, relu_mask=ed)
    else:
      ops.conv2d_bwd_data(g, xd, wd, out)
    return out

  def ref(dt):
    a, k = tx.to(dt), tw.to(dt)
    if mode == 'fwd_res':
      return F.conv2d(F.relu(a), k, torch.tensor(b).to(dt), padding=1).permute(0, 2, 3, 1) + te.to(dt)
    if mode == 'fwd_plain':
      return F.relu(F.conv2d(a, k, None, padding=1)).permute(0, 2, 3, 1)
    y = F.conv_transpose2d(a, k, padding=1).permute(0, 2, 3, 1)     # gradient of conv2d(., k) wrt its input, at dY = a
    if mode == 'dg_mask_add':
      return y * (te > 0).to(dt) + te2.to(dt)
    if mode == 'dg_mask':
      return y * (te > 0).to(dt)
    return y

  got = run()
  r32, r64 = ref(torch.float32).numpy().astype(np.float64), ref(torch.float64).numpy()
  e_hip = np.max(np.abs(got.cpu().numpy().astype(np.float64) - r64)); e_f32 = np.max(np.abs(r32 - r64))
  print('wsx %s n=%d %dx%d: err hip %.3e  torch fp32 %.3e' % (mode, n, h, w, e_hip, e_f32))
  assert e_hip <= max(2.0 * e_f32, 2e-6 * np.abs(r64).max()), (mode, e_hip, e_f32)
  assert torch.equal(got, run())


@pytest.mark.parametrize('n,mode', [(64, 'fwd'), (257, 'fwd'), (601, 'fwd_nobias'), (65, 'dg'), (515, 'dg')])
def test_fgx_stack_entry_conv_fp32_accuracy(device, n, mode):
  """ImpalaDeep's 16 -> 32 stack-entry layer on the 36 x 48 map (fgx.h: bands of the batch staged once as bf16 planes,
  weights in registers, one tap per MFMA step): forward with and without bias (an odd number of bands: the second band
  of the last unit does not exist) and the data gradient.  As close to an fp64 evaluation as torch's fp32 convolution
  (<= 2x), bit-identical from call to call, the bf16 pipe reported by conv2d_pipe."""
  from seed_rl_amd import ops
  rng = np.random.default_rng(n)
  fwd = mode != 'dg'
  x = rng.normal(size=(n, 36, 48, 16 if fwd else 32)).astype(np.float32)
  wt = (rng.normal(size=(3, 3, 16, 32)) / 12).astype(np.float32)
  b = rng.normal(size=32).astype(np.float32)
  g = ops.conv_geom(n, 36, 48, 16, 3, 3, 1, 'same', 32)
  assert ops.conv2d_pipe(g, 0 if fwd else 1) == 6
  xd, wd, bd = dev(x, device), dev(wt, device), dev(b, device)
  tx = torch.tensor(x).permute(0, 3, 1, 2)
  tw = torch.tensor(wt).permute(3, 2, 0, 1)

  def run():
    out = torch.full((n, 36, 48, 32 if fwd else 16), 7.0, device=device)
    if fwd: ops.conv2d_fwd(g, xd, wd, bd if mode == 'fwd' else None, out)
    else: ops.conv2d_bwd_data(g, xd, wd, out)
    return out

  def ref(dt):
    if fwd:
      return F.conv2d(tx.to(dt), tw.to(dt), torch.tensor(b).to(dt) if mode == 'fwd' else None, padding=1).permute(0, 2, 3, 1)
    return F.conv_transpose2d(tx.to(dt), tw.to(dt), padding=1).permute(0, 2, 3, 1)

  got = run()
  r32, r64 = ref(torch.float32).numpy().astype(np.float64), ref(torch.float64).numpy()
  e_hip = np.max(np.abs(got.cpu().numpy().astype(np.float64) - r64)); e_f32 = np.max(np.abs(r32 - r64))
  print('fgx %s n=%d: err hip %.3e  torch fp32 %.3e' % (mode, n, e_hip, e_f32))
  assert e_hip <= max(2.0 * e_f32, 2e-6 * np.abs(r64).max()), (mode, e_hip, e_f32)
  assert torch.equal(got, run())


def _nibbles(x):
  """bit q of byte [..., quad] = x[..., 4 quad + q] > 0"""
  v = (x.reshape(x.shape[:-1] + (x.shape[-1] // 4, 4)) > 0).astype(np.uint8)
  return (v * np.array([1, 2, 4, 8], np.uint8)).sum(-1).astype(np.uint8)


@pytest.mark.parametrize('n,h,w', [(300, 36, 48), (515, 18, 24), (700, 9, 12)])
def test_residual_block_relu_byte_masks(device, n, h, w):
  """ImpalaDeep's residual-block layers with their ReLU masks as bytes (wsx.h / wsy.h): conv2d_fwd(out_bits=) writes the
  sign of its output from the epilogue -- one byte per four channels, ragged last workgroup -- next to the output of
  the plain call, bit for bit (with and without ReLU on the input and residual); conv2d_bwd_data(relu_bits=, add=)
  equals the fp32-mask call bit for bit (with and without the skip-path add; zeros, negative zeros and subnormals in
  the masked tensor: > 0 is the test, as in ReluGrad)."""
  from seed_rl_amd import ops
  rng = np.random.default_rng(n)
  C = 16 if h == 36 else 32
  g = ops.conv_geom(n, h, w, C, 3, 3, 1, 'same', C)
  assert ops.conv2d_fwd_outbits_supported(g) and ops.conv2d_bwd_data_bits_supported(g)
  x = rng.normal(size=(n, h, w, C)).astype(np.float32)
  x[0, 0, 0, :4] = [0.0, -0.0, 1e-40, -1e-40]
  wt = (rng.normal(size=(3, 3, C, C)) / np.sqrt(9 * C)).astype(np.float32)
  b = rng.normal(size=C).astype(np.float32)
  res = rng.normal(size=(n, h, w, C)).astype(np.float32)
  dy = rng.normal(size=(n, h, w, C)).astype(np.float32)
  xd, wd, bd, rd, dyd = dev(x, device), dev(wt, device), dev(b, device), dev(res, device), dev(dy, device)
  for in_relu, residual in ((True, None), (True, rd), (False, None)):
    plain = torch.full((n, h, w, C), 7.0, device=device)
    ops.conv2d_fwd(g, xd, wd, bd, plain, in_relu=in_relu, residual=residual)
    out = torch.full((n, h, w, C), -7.0, device=device)
    bits = torch.full((n, h, w, C // 4), 0xA5, dtype=torch.uint8, device=device)
    ops.conv2d_fwd(g, xd, wd, bd, out, in_relu=in_relu, residual=residual, out_bits=bits)
    assert torch.equal(out, plain)
    assert np.array_equal(bits.cpu().numpy(), _nibbles(out.cpu().numpy()))
  xbits = torch.tensor(_nibbles(x), device=device)
  for add in (None, rd):
    ref = torch.full((n, h, w, C), 7.0, device=device)
    ops.conv2d_bwd_data(g, dyd, wd, ref, relu_mask=xd, add=add)
    got = torch.full((n, h, w, C), -7.0, device=device)
    ops.conv2d_bwd_data(g, dyd, wd, got, relu_bits=xbits, add=add)
    assert torch.equal(got, ref)


def test_pool_outputs_relu_byte_masks(device):
  """The pooled tensors that enter a stack's first residual block: maxpool_fwd(y_bits=) and
  conv3x3_u8_pool_fwd(pooled_bits=) write the sign of their output as bytes and leave output and argmax unchanged."""
  from seed_rl_amd import ops
  rng = np.random.default_rng(3)
  x = dev(rng.normal(size=(37, 36, 48, 32)).astype(np.float32), device)
  y0 = torch.empty((37, 18, 24, 32), device=device); a0 = torch.empty((37, 18, 24, 32), dtype=torch.uint8, device=device)
  y1, a1 = torch.empty_like(y0), torch.empty_like(a0)
  bits = torch.full((37, 18, 24, 8), 0xA5, dtype=torch.uint8, device=device)
  ops.maxpool_fwd(x, y0, a0)
  ops.maxpool_fwd(x, y1, a1, y_bits=bits)
  assert torch.equal(y0, y1) and torch.equal(a0, a1)
  assert np.array_equal(bits.cpu().numpy(), _nibbles(y1.cpu().numpy()))
  fr = torch.tensor(rng.integers(0, 256, size=(21, 72, 96, 3)).astype(np.uint8), device=device)
  w = dev((rng.normal(size=(3, 3, 3, 16)) / 5).astype(np.float32), device); b = dev(rng.normal(size=16).astype(np.float32), device)
  p0 = torch.empty((21, 36, 48, 16), device=device); g0 = torch.empty((21, 36, 48, 16), dtype=torch.uint8, device=device)
  p1, g1 = torch.empty_like(p0), torch.empty_like(g0)
  pb = torch.full((21, 36, 48, 4), 0xA5, dtype=torch.uint8, device=device)
  ops.conv3x3_u8_pool_fwd(fr, w, b, p0, g0)
  ops.conv3x3_u8_pool_fwd(fr, w, b, p1, g1, pooled_bits=pb)
  assert torch.equal(p0, p1) and torch.equal(g0, g1)
  assert np.array_equal(pb.cpu().numpy(), _nibbles(p1.cpu().numpy()))


@pytest.mark.parametrize('n', [64, 129, 600])
def test_pool_backward_inside_the_data_gradient(device, n):
  """ImpalaDeep's stack-entry layer behind its max-pool (fgx.h, POOL loader): conv2d_bwd_data_pool rebuilds the pre-pool
  gradient from the pooled map's gradient and the argmax bytes while it stages it -- dx AND the pre-pool gradient it
  writes for the weight gradient are bit-identical to maxpool_bwd followed by conv2d_bwd_data (ties in the pool, image
  borders, a ragged last workgroup)."""
  from seed_rl_amd import ops
  rng = np.random.default_rng(n)
  a = rng.normal(size=(n, 36, 48, 32)).astype(np.float32)
  a[:, ::3, ::5] = np.round(a[:, ::3, ::5])                 # ties inside pool windows: the first maximum wins
  wt = (rng.normal(size=(3, 3, 16, 32)) / 12).astype(np.float32)
  dp = rng.normal(size=(n, 18, 24, 32)).astype(np.float32)
  g = ops.conv_geom(n, 36, 48, 16, 3, 3, 1, 'same', 32)
  assert ops.conv2d_bwd_data_pool_supported(g)
  ad, wd, dpd = dev(a, device), dev(wt, device), dev(dp, device)
  y = torch.empty((n, 18, 24, 32), device=device); arg = torch.empty((n, 18, 24, 32), dtype=torch.uint8, device=device)
  ops.maxpool_fwd(ad, y, arg)
  d_a0 = torch.full((n, 36, 48, 32), 7.0, device=device); dx0 = torch.full((n, 36, 48, 16), 7.0, device=device)
  ops.maxpool_bwd(dpd, arg, d_a0)
  ops.conv2d_bwd_data(g, d_a0, wd, dx0)
  d_a1 = torch.full((n, 36, 48, 32), -7.0, device=device); dx1 = torch.full((n, 36, 48, 16), -7.0, device=device)
  ops.conv2d_bwd_data_pool(g, dpd, arg, wd, dx1, d_a1)
  assert torch.equal(d_a1, d_a0)
  assert torch.equal(dx1, dx0)
  dx2 = torch.full((n, 36, 48, 16), 3.0, device=device)
  ops.conv2d_bwd_data_pool(g, dpd, arg, wd, dx2, d_a1)
  assert torch.equal(dx2, dx1)


@pytest.mark.parametrize('n,mode', [(1024, 'fwd'), (1030, 'fwd_plain'), (1027, 'dg_mask'), (1100, 'dg')])
def test_cgx_dqn_conv3_fp32_accuracy(device, n, mode):
  """The DQN torso's third convolution, 3 x 3 'valid' 64 -> 64 on 9 x 9 maps (cgx.h: units of whole images, the reduction
  split over four waves whose partial sums meet in LDS): forward with bias + ReLU and plain, data gradient with and
  without the ReLU mask; image counts that are not a multiple of the unit.  As close to an fp64 evaluation as torch's
  fp32 convolution (<= 2x), bit-identical from call to call, the bf16 pipe reported by conv2d_pipe."""
  from seed_rl_amd import ops
  rng = np.random.default_rng(n)
  fwd = mode.startswith('fwd')
  x = rng.normal(size=(n, 9, 9, 64) if fwd else (n, 7, 7, 64)).astype(np.float32)
  wt = (rng.normal(size=(3, 3, 64, 64)) / 24).astype(np.float32)
  b = rng.normal(size=64).astype(np.float32)
  mask = rng.normal(size=(n, 9, 9, 64)).astype(np.float32)
  g = ops.conv_geom(n, 9, 9, 64, 3, 3, 1, 'valid', 64)
  assert ops.conv2d_pipe(g, 0 if fwd else 1) == 6
  xd, wd, bd, md = dev(x, device), dev(wt, device), dev(b, device), dev(mask, device)
  tx = torch.tensor(x).permute(0, 3, 1, 2)
  tw = torch.tensor(wt).permute(3, 2, 0, 1)

  def run():
    out = torch.full((n, 7, 7, 64) if fwd else (n, 9, 9, 64), 7.0, device=device)
    if mode == 'fwd': ops.conv2d_fwd(g, xd, wd, bd, out, out_relu=True)
    elif mode == 'fwd_plain': ops.conv2d_fwd(g, xd, wd, None, out)
    elif mode == 'dg_mask': ops.conv2d_bwd_data(g, xd, wd, out, relu_mask=md)
    else: ops.conv2d_bwd_data(g, xd, wd, out)
    return out

  def ref(dt):
    if mode == 'fwd':
      return F.relu(F.conv2d(tx.to(dt), tw.to(dt), torch.tensor(b).to(dt))).permute(0, 2, 3, 1)
    if mode == 'fwd_plain':
      return F.conv2d(tx.to(dt), tw.to(dt)).permute(0, 2, 3, 1)
    y = F.conv_transpose2d(tx.to(dt), tw.to(dt)).permute(0, 2, 3, 1)
    return y * (torch.tensor(mask) > 0).to(dt) if mode == 'dg_mask' else y

  got = run()
  r32, r64 = ref(torch.float32).numpy().astype(np.float64), ref(torch.float64).numpy()
  e_hip = np.max(np.abs(got.cpu().numpy().astype(np.float64) - r64)); e_f32 = np.max(np.abs(r32 - r64))
  print('cgx %s n=%d: err hip %.3e  torch fp32 %.3e' % (mode, n, e_hip, e_f32))
  assert e_hip <= max(2.0 * e_f32, 2e-6 * np.abs(r64).max()), (mode, e_hip, e_f32)
  assert torch.equal(got, run())


@pytest.mark.parametrize('n,relu', [(512, True), (777, False)])
def test_cgx_dqn_conv2_forward_fp32_accuracy(device, n, relu):
  """The DQN torso's second convolution, 4 x 4 stride 2 'valid' 32 -> 64 on 20 x 20 maps, forward (cgx.h: one image per
  unit, even and odd columns of a row stored apart, two taps x 8 channels per reduction step).  As close to an fp64
  evaluation as torch's fp32 convolution (<= 2x), bit-identical from call to call."""
  from seed_rl_amd import ops
  rng = np.random.default_rng(n)
  x = rng.normal(size=(n, 20, 20, 32)).astype(np.float32)
  wt = (rng.normal(size=(4, 4, 32, 64)) / 22).astype(np.float32)
  b = rng.normal(size=64).astype(np.float32)
  g = ops.conv_geom(n, 20, 20, 32, 4, 4, 2, 'valid', 64)
  assert ops.conv2d_pipe(g, 0) == 6
  xd, wd, bd = dev(x, device), dev(wt, device), dev(b, device)
  tx = torch.tensor(x).permute(0, 3, 1, 2); tw = torch.tensor(wt).permute(3, 2, 0, 1)

  def run():
    out = torch.full((n, 9, 9, 64), 7.0, device=device)
    ops.conv2d_fwd(g, xd, wd, bd if relu else None, out, out_relu=relu)
    return out

  def ref(dt):
    y = F.conv2d(tx.to(dt), tw.to(dt), torch.tensor(b).to(dt) if relu else None, stride=2)
    return (F.relu(y) if relu else y).permute(0, 2, 3, 1)

  got = run()
  r32, r64 = ref(torch.float32).numpy().astype(np.float64), ref(torch.float64).numpy()
  e_hip = np.max(np.abs(got.cpu().numpy().astype(np.float64) - r64)); e_f32 = np.max(np.abs(r32 - r64))
  print('cgx conv2 n=%d: err hip %.3e  torch fp32 %.3e' % (n, e_hip, e_f32))
  assert e_hip <= max(2.0 * e_f32, 2e-6 * np.abs(r64).max()), (e_hip, e_f32)
  assert torch.equal(got, run())


@pytest.mark.parametrize('n,mask', [(512, True), (777, False)])
def test_cgx_dqn_conv2_data_gradient_fp32_accuracy(device, n, mask):
  """The data gradient of the DQN torso's second convolution (4 x 4 stride 2, 32 <- 64; cgx2.h: four stride-parity classes
  of 2 x 2 taps, the eight waves = class x half of the dY channels, an odd image count) with and without the ReLU mask:
  as close to fp64 as torch's fp32 transposed convolution (<= 2x), exact zeros under the mask, bit-identical call to call."""
  from seed_rl_amd import ops
  rng = np.random.default_rng(n)
  dy = rng.normal(size=(n, 9, 9, 64)).astype(np.float32)
  wt = (rng.normal(size=(4, 4, 32, 64)) / 22).astype(np.float32)
  x = rng.normal(size=(n, 20, 20, 32)).astype(np.float32)
  g = ops.conv_geom(n, 20, 20, 32, 4, 4, 2, 'valid', 64)
  assert ops.conv2d_pipe(g, 1) == 6
  dyd, wd, xd = dev(dy, device), dev(wt, device), dev(x, device)
  tdy = torch.tensor(dy).permute(0, 3, 1, 2); tw = torch.tensor(wt).permute(3, 2, 0, 1)

  def run():
    out = torch.full((n, 20, 20, 32), 7.0, device=device)
    ops.conv2d_bwd_data(g, dyd, wd, out, relu_mask=xd if mask else None)
    return out

  def ref(dt):
    y = F.conv_transpose2d(tdy.to(dt), tw.to(dt), stride=2).permute(0, 2, 3, 1)
    return y * (torch.tensor(x) > 0).to(dt) if mask else y

  got = run()
  r32, r64 = ref(torch.float32).numpy().astype(np.float64), ref(torch.float64).numpy()
  e_hip = np.max(np.abs(got.cpu().numpy().astype(np.float64) - r64)); e_f32 = np.max(np.abs(r32 - r64))
  print('cgx2 dgrad n=%d: err hip %.3e  torch fp32 %.3e' % (n, e_hip, e_f32))
  assert e_hip <= max(2.0 * e_f32, 2e-6 * np.abs(r64).max()), (e_hip, e_f32)
  if mask:
    assert np.all(got.cpu().numpy()[x <= 0] == 0.0)
  assert torch.equal(got, run())


WGX_SHAPES = {                                             # name -> (ih, iw, cin, k, stride, padding, cout)
    'atari2': (20, 20, 16, 4, 2, 'valid', 32), 'deep16': (36, 48, 16, 3, 1, 'same', 16), 'deep16x32': (36, 48, 16, 3, 1, 'same', 32),
    'deep32a': (18, 24, 32, 3, 1, 'same', 32), 'deep32b': (9, 12, 32, 3, 1, 'same', 32)}


def _wgrad_refs(x, dy, k, stride, padding, cout, in_relu):
  """(dW, db) of conv2d(relu?(x), W) wrt W / bias at upstream gradient dy: torch fp32 and fp64, [kh, kw, cin, cout]."""
  out = []
  for dt in (torch.float32, torch.float64):
    tx = torch.tensor(x).to(dt).permute(0, 3, 1, 2)
    if in_relu: tx = F.relu(tx)
    w = torch.zeros((cout, x.shape[3], k, k), dtype=dt, requires_grad=True)
    y = F.conv2d(tx, w, None, stride=stride, padding=1 if padding == 'same' else 0)
    tdy = torch.tensor(dy).to(dt).permute(0, 3, 1, 2)
    y.backward(tdy)
    out.append((w.grad.permute(2, 3, 1, 0).numpy().astype(np.float64), tdy.sum((0, 2, 3)).numpy().astype(np.float64)))
  return out


@pytest.mark.parametrize('name,n,in_relu', [('atari2', 33, False), ('atari2', 300, True), ('atari2', 1029, False), ('atari2', 10752, False),
                                            ('deep16', 40, True), ('deep16', 259, False), ('deep16x32', 70, True), ('deep16x32', 300, False),
                                            ('deep32a', 64, False), ('deep32a', 515, True), ('deep32b', 70, True), ('deep32b', 1300, False)])
def test_wgx_weight_gradient_fp32_accuracy(device, name, n, in_relu):
  """Weight gradients on the bf16 matrix pipe (wgx.h: exact three-way split of X and dY, six plane products, MFMA
  operands by transposing LDS reads; second Atari conv and ImpalaDeep's 3x3 layers): as close to an fp64 evaluation as
  torch's fp32 gradient is (max error <= 2x), dW and the bias gradient; one unit per workgroup up to long runs with a
  ragged last workgroup, bands at the top / bottom of 'same'-padded images; bit-identical from call to call."""
  from seed_rl_amd import ops
  ih, iw, cin, k, stride, padding, cout = WGX_SHAPES[name]
  rng = np.random.default_rng(n + ih)
  g = ops.conv_geom(n, ih, iw, cin, k, k, stride, padding, cout)
  x = rng.normal(size=(n, ih, iw, cin)).astype(np.float32)
  dy = rng.normal(size=(n, g.oh, g.ow, cout)).astype(np.float32)
  (w32, b32), (w64, b64) = _wgrad_refs(x, dy, k, stride, padding, cout, in_relu)
  xd, dyd = dev(x, device), dev(dy, device)
  ws = torch.empty(ops.conv2d_bwd_weight_workspace_bytes(g) // 4 + 4, device=device)

  def run(fill):
    dw = torch.full((k, k, cin, cout), fill, device=device); db = torch.full((cout,), fill, device=device)
    ops.conv2d_bwd_weight(g, xd, dyd, dw, db, ws, in_relu=in_relu)
    return dw, db
  dw, db = run(7.0)
  assert ops.conv2d_pipe(g, 2) == 6
  for what, got, r32, r64 in (('dW', dw, w32, w64), ('db', db, b32, b64)):
    e_hip = np.max(np.abs(got.cpu().numpy().astype(np.float64) - r64)); e_f32 = np.max(np.abs(r32 - r64))
    print('wgx %s %s n=%d: err hip %.3e  torch fp32 %.3e  scale %.3e' % (name, what, n, e_hip, e_f32, np.abs(r64).max()))
    assert e_hip <= max(2.0 * e_f32, 2e-6 * np.abs(r64).max()), (name, what, e_hip, e_f32)
  dw2, db2 = run(-3.0)
  assert torch.equal(dw, dw2) and torch.equal(db, db2)


@pytest.mark.parametrize('name,n', [('atari2', 300), ('deep16', 64), ('deep32a', 96)])
def test_wgx_weight_gradient_ill_conditioned(device, name, n):
  """The same kernels on inputs a split into bf16 parts could get wrong: channels scaled by 2^+-40 (wide exponent
  spread across the rows of dW), post-ReLU activations with 90 % zeros, a block of values below 2^-110 (the low parts
  are bf16 subnormals / flush to zero: bounded by 2^-126 per product, invisible next to the fp32 reference's own
  rounding), and non-finite inputs: an inf / NaN in X or dY must make exactly the entries of dW non-finite that
  torch's fp32 gradient has non-finite."""
  from seed_rl_amd import ops
  ih, iw, cin, k, stride, padding, cout = WGX_SHAPES[name]
  rng = np.random.default_rng(n)
  g = ops.conv_geom(n, ih, iw, cin, k, k, stride, padding, cout)
  x = rng.normal(size=(n, ih, iw, cin)).astype(np.float32)
  x *= (rng.random(size=x.shape) < 0.1)                     # 90 % zeros
  x *= np.exp2(rng.integers(-40, 41, size=cin)).astype(np.float32)
  x[: n // 4, :, :, 0] = (rng.normal(size=(n // 4, ih, iw)) * 2.0 ** -115).astype(np.float32)
  dy = rng.normal(size=(n, g.oh, g.ow, cout)).astype(np.float32)
  dy *= np.exp2(rng.integers(-40, 41, size=cout)).astype(np.float32)
  ws = torch.empty(ops.conv2d_bwd_weight_workspace_bytes(g) // 4 + 4, device=device)

  def run(xa, dya):
    dw = torch.full((k, k, cin, cout), 7.0, device=device); db = torch.full((cout,), 7.0, device=device)
    ops.conv2d_bwd_weight(g, dev(xa, device), dev(dya, device), dw, db, ws)
    return dw.cpu().numpy().astype(np.float64), db.cpu().numpy().astype(np.float64)
  (w32, b32), (w64, b64) = _wgrad_refs(x, dy, k, stride, padding, cout, False)
  dw, db = run(x, dy)
  # per (input channel, output channel) block: the scales differ by up to 2^160 between blocks
  for ci in range(cin):
    for co in range(0, cout, 8):
      blk = np.s_[:, :, ci, co:co + 8]
      for j in range(8):
        sl = (slice(None), slice(None), ci, co + j)
        e_hip = np.max(np.abs(dw[sl] - w64[sl])); e_f32 = np.max(np.abs(w32[sl] - w64[sl]))
        assert e_hip <= max(2.0 * e_f32, 2e-6 * np.abs(w64[sl]).max(), 1e-37), (name, ci, co + j, e_hip, e_f32)
  # bias gradient: a plain sum of dY -- any fp32 summation order is within a few ulps of sum |dy| of the column
  assert np.all(np.abs(db - b64) <= np.maximum(2.0 * np.abs(b32 - b64), 2e-7 * np.abs(dy.astype(np.float64)).sum((0, 1, 2))))
  # non-finite inputs
  xi, dyi = x.copy(), dy.copy()
  xi[1, ih // 2, iw // 2, 3] = np.inf
  xi[2, 0, 0, 5] = np.nan
  dyi[3, g.oh - 1, g.ow - 1, 7] = -np.inf
  # reference: the im2col evaluation with the zero padding as explicit zeros (0 * inf = NaN, which is what the kernel's
  # zero-filled halo computes; torch's padded convolution skips out-of-image taps instead -- the two differ only in
  # entries (tap, ci, co) whose tap leaves the image at a pixel where dY[., co] is non-finite: pinned here)
  xp = np.pad(xi, ((0, 0), (1, 1), (1, 1), (0, 0))) if padding == 'same' else xi
  (w32, b32), _ = _wgrad_refs(xp, dyi, k, stride, 'valid', cout, False)
  dw, db = run(xi, dyi)
  assert np.array_equal(np.isfinite(dw), np.isfinite(w32)), (np.sum(~np.isfinite(dw)), np.sum(~np.isfinite(w32)))
  assert np.array_equal(np.isfinite(db), np.isfinite(b32))
  if padding == 'same':                                   # ... and it is a superset of torch's own non-finite set
    (t32, _), _ = _wgrad_refs(xi, dyi, k, stride, padding, cout, False)
    assert not np.any(np.isfinite(dw) & ~np.isfinite(t32))


def _ill_scale(rng, n):
  return np.exp2(rng.integers(-40, 41, size=n)).astype(np.float32)


def _col_check(name, got, r32, r64, axis_last=True):
  """Per output column (last axis): |got - fp64| <= max(2 |torch fp32 - fp64|, 2e-6 max|fp64|) taken over the column --
  the columns of an ill-scaled problem differ by up to 2^160, a whole-tensor maximum would only test the largest."""
  g, a, b = (np.asarray(t, np.float64).reshape(-1, np.shape(t)[-1]) for t in (got, r32, r64))
  e_hip, e_f32, sc = np.abs(g - b).max(0), np.abs(a - b).max(0), np.abs(b).max(0)
  bad = e_hip > np.maximum(np.maximum(2.0 * e_f32, 2e-6 * sc), 1e-37)
  assert not bad.any(), (name, np.nonzero(bad)[0][:8], e_hip[bad][:4], e_f32[bad][:4], sc[bad][:4])


@pytest.mark.parametrize('kind', ['x6', 'x8', 'wfx', 'wdx', 'wsx_fwd', 'wsx_dg', 'wsy_fwd', 'wsy_dg', 'fgx_fwd', 'fgx_dg', 'cgx_fwd', 'cgx_dg', 'cgx2_fwd', 'cgx2_dg'])
def test_bf16x6_kernels_ill_conditioned(device, kind):
  """VERDICT r4 task 7b.  Every kernel that evaluates fp32 x fp32 on the bf16 pipe through the three-way split, on inputs
  the split could get wrong: input channels / output channels scaled by 2^+-40 (wide exponent spread across the
  reduction AND across the outputs), post-ReLU activations with 90 % zeros, a block of values below 2^-110 (low parts
  become bf16 subnormals), and non-finite inputs: the output must be non-finite exactly where torch's fp32 result is
  (an inf splits into h = inf, m = inf - inf = NaN: non-finite either way; the zero halo of a 'same' layer is compared
  with the explicit-zero-padding evaluation, in which 0 * inf = NaN like in the kernels' zero-filled LDS rows)."""
  from seed_rl_amd import ops
  rng = np.random.default_rng(hash(kind) % 1000)
  if kind in ('x6', 'x8'):
    n, cin, cout = (2304, 1000, 256) if kind == 'x6' else (4224, 520, 264)
    x = rng.normal(size=(n, cin)).astype(np.float32) * (rng.random(size=(n, cin)) < 0.1) * _ill_scale(rng, cin)
    x[: n // 4, 0] = (rng.normal(size=n // 4) * 2.0 ** -115).astype(np.float32)
    w = (rng.normal(size=(cin, cout)) / np.sqrt(cin)).astype(np.float32) * _ill_scale(rng, cout)
    dy = rng.normal(size=(n, cout)).astype(np.float32) * _ill_scale(rng, cout)
    g = ops.dense_geom(n, cin, cout)
    ws = torch.empty(ops.conv2d_bwd_weight_workspace_bytes(g) // 4 + 4, device=device)

    def run(xa, wa, dya):
      xd, wd, dyd = dev(xa, device), dev(wa, device), dev(dya, device)
      out = torch.empty((n, cout), device=device); dx = torch.empty((n, cin), device=device)
      dw = torch.empty((cin, cout), device=device); db = torch.empty(cout, device=device)
      ops.conv2d_fwd(g, xd, wd, None, out)
      ops.conv2d_bwd_data(g, dyd, wd, dx)
      ops.conv2d_bwd_weight(g, xd, dyd, dw, db, ws)
      return [t.cpu().numpy() for t in (out, dx, dw)]

    def ref(xa, wa, dya, dt):
      tx, tw, tdy = (torch.tensor(a).to(dt) for a in (xa, wa, dya))
      return [(tx @ tw).numpy(), (tdy @ tw.T).numpy(), (tx.T @ tdy).numpy()]
    names = ('fwd', 'dgrad', 'wgrad')
    inj = lambda xa, wa, dya: (xa.__setitem__((5, 7), np.inf), xa.__setitem__((9, 3), np.nan), dya.__setitem__((11, 2), -np.inf))
    args = (x, w, dy)
  else:
    if kind in ('wfx', 'wdx'):
      n, ih, iw, cin, k, stride, padding, cout = 300, 20, 20, 16, 4, 2, 'valid', 32
    elif kind.startswith('wsx'):
      n, ih, iw, cin, k, stride, padding, cout = 520, 18, 24, 32, 3, 1, 'same', 32
    elif kind.startswith('fgx'):
      n, ih, iw, cin, k, stride, padding, cout = 131, 36, 48, 16, 3, 1, 'same', 32
    elif kind in ('cgx2_fwd', 'cgx2_dg'):
      n, ih, iw, cin, k, stride, padding, cout = 515, 20, 20, 32, 4, 2, 'valid', 64
    elif kind.startswith('cgx'):
      n, ih, iw, cin, k, stride, padding, cout = 1025, 9, 9, 64, 3, 1, 'valid', 64
    else:
      n, ih, iw, cin, k, stride, padding, cout = 260, 36, 48, 16, 3, 1, 'same', 16
    g = ops.conv_geom(n, ih, iw, cin, k, k, stride, padding, cout)
    fwd = kind in ('wfx', 'wsx_fwd', 'wsy_fwd', 'fgx_fwd', 'cgx_fwd', 'cgx2_fwd')
    src_c, dst_c = (cin, cout) if fwd else (cout, cin)
    shape = (n, ih, iw, cin) if fwd else (n, g.oh, g.ow, cout)
    x = rng.normal(size=shape).astype(np.float32) * (rng.random(size=shape) < 0.1) * _ill_scale(rng, src_c)
    x[: n // 4, :, :, 0] = (rng.normal(size=(n // 4,) + shape[1:3]) * 2.0 ** -115).astype(np.float32)
    w = (rng.normal(size=(k, k, cin, cout)) / np.sqrt(k * k * cin)).astype(np.float32)
    w = w * (_ill_scale(rng, cout) if fwd else _ill_scale(rng, cin)[:, None])
    pad = 1 if padding == 'same' else 0

    def run(xa, wa, _):
      xd, wd = dev(xa, device), dev(wa, device)
      out = torch.full((n, g.oh, g.ow, cout) if fwd else (n, ih, iw, cin), 7.0, device=device)
      if fwd: ops.conv2d_fwd(g, xd, wd, None, out)
      else: ops.conv2d_bwd_data(g, xd, wd, out)
      return [out.cpu().numpy()]

    def ref(xa, wa, _, dt, explicit_pad=False):
      tx = torch.tensor(xa).to(dt).permute(0, 3, 1, 2); tw = torch.tensor(wa).to(dt).permute(3, 2, 0, 1)
      if fwd:
        if explicit_pad and pad:
          return [F.conv2d(F.pad(tx, (1, 1, 1, 1)), tw, None, stride=stride).permute(0, 2, 3, 1).numpy()]
        return [F.conv2d(tx, tw, None, stride=stride, padding=pad).permute(0, 2, 3, 1).numpy()]
      return [F.conv_transpose2d(tx, tw, stride=stride, padding=pad).permute(0, 2, 3, 1).numpy()]
    names = (kind,)
    inj = lambda xa, wa, dya: (xa.__setitem__((1, shape[1] // 2, shape[2] // 2, 3), np.inf), xa.__setitem__((2, 0, 0, 5), np.nan),
                               xa.__setitem__((3, shape[1] - 1, shape[2] - 1, 7), -np.inf))
    args = (x, w, None)
    assert ops.conv2d_pipe(g, 0 if fwd else 1) == 6
  got, r32, r64 = run(*args), ref(*args, torch.float32), ref(*args, torch.float64)
  for nm, a, b, c in zip(names, got, r32, r64):
    _col_check(kind + ' ' + nm, a, b, c)
  bad = tuple(None if a is None else a.copy() for a in args)
  inj(*bad)
  got, r32 = run(*bad), ref(*bad, torch.float32)
  for nm, a, b in zip(names, got, r32):
    assert np.array_equal(np.isfinite(a), np.isfinite(b)), (kind, nm, int((~np.isfinite(a)).sum()), int((~np.isfinite(b)).sum()))


@pytest.mark.parametrize('n,cin,cout', [(4100, 520, 264), (4096, 2592, 256), (4224, 256, 1024)])   # (>= 4096 rows)
def test_x8_gemm_epilogues_and_tails(device, n, cin, cout):
  """The 8-wave bf16x6 Dense kernels (xgemm8.h: 128 x 256 tiles, the small operand pre-split into k-tile slabs) on ragged
  shapes -- rows not a multiple of 128, K not a multiple of 32, a last column tile of 8 columns -- with every fused
  epilogue the layers use: bias + residual + ReLU and input ReLU (forward), ReLU mask + accumulate (data gradient),
  input ReLU + bias gradient (weight gradient).  Held to an fp64 evaluation at the error of torch's fp32 matmul."""
  from seed_rl_amd import _lib, ops
  import ctypes
  rng = np.random.default_rng(n + cin)
  x = rng.normal(size=(n, cin)).astype(np.float32)
  w = (rng.normal(size=(cin, cout)) / np.sqrt(cin)).astype(np.float32)
  b = rng.normal(size=cout).astype(np.float32)
  res = rng.normal(size=(n, cout)).astype(np.float32)
  dy = rng.normal(size=(n, cout)).astype(np.float32)
  mask = rng.normal(size=(n, cin)).astype(np.float32)
  add = rng.normal(size=(n, cin)).astype(np.float32)
  g = ops.dense_geom(n, cin, cout)
  for which in range(3):
    assert _lib.lib().seedhip_conv2d_pipe(ctypes.byref(g), which) == 6
  f64 = lambda a: a.astype(np.float64)
  xr = np.maximum(x, 0.0)

  def check(name, hip, f32, ref):
    e_hip, e_f32, scale = np.max(np.abs(f64(hip) - ref)), np.max(np.abs(f64(f32) - ref)), np.abs(ref).max()
    print('x8 %s n=%d cin=%d cout=%d: err hip %.3e  torch fp32 %.3e  scale %.3e' % (name, n, cin, cout, e_hip, e_f32, scale))
    assert e_hip <= max(2.0 * e_f32, 2e-6 * scale), (name, e_hip, e_f32, scale)

  t = torch.tensor
  out = torch.empty((n, cout), device=device)
  ops.conv2d_fwd(g, dev(x, device), dev(w, device), dev(b, device), out, in_relu=True, out_relu=True, residual=dev(res, device))
  check('fwd', out.cpu().numpy(), torch.relu(t(xr) @ t(w) + t(b) + t(res)).numpy(), np.maximum(f64(xr) @ f64(w) + f64(b) + f64(res), 0.0))
  out2 = torch.empty((n, cout), device=device)
  ops.conv2d_fwd(g, dev(x, device), dev(w, device), None, out2)
  check('fwd plain', out2.cpu().numpy(), (t(x) @ t(w)).numpy(), f64(x) @ f64(w))
  dx = torch.empty((n, cin), device=device)
  ops.conv2d_bwd_data(g, dev(dy, device), dev(w, device), dx, relu_mask=dev(mask, device), add=dev(add, device))
  check('dgrad', dx.cpu().numpy(), ((t(dy) @ t(w).T) * (t(mask) > 0) + t(add)).numpy(), (f64(dy) @ f64(w).T) * (mask > 0) + f64(add))
  dw = torch.empty((cin, cout), device=device); db = torch.empty(cout, device=device)
  ws = torch.empty(ops.conv2d_bwd_weight_workspace_bytes(g) // 4 + 4, device=device)
  ops.conv2d_bwd_weight(g, dev(x, device), dev(dy, device), dw, db, ws, in_relu=True)
  check('wgrad', dw.cpu().numpy(), (t(xr).T @ t(dy)).numpy(), f64(xr).T @ f64(dy))
  check('bias grad', db.cpu().numpy(), t(dy).sum(0).numpy(), f64(dy).sum(0))
  dw2 = torch.empty((cin, cout), device=device)
  ops.conv2d_bwd_weight(g, dev(x, device), dev(dy, device), dw2, None, ws)
  check('wgrad no bias', dw2.cpu().numpy(), (t(x).T @ t(dy)).numpy(), f64(x).T @ f64(dy))


def test_x8_data_gradient_path_in_child():
  """The library reads SEEDHIP_X8 once: the 8-wave DATA-GRADIENT kernel (both operands pre-split; off by default, the
  xgemm.h kernel is faster at these K) is exercised by the same test in a child process with every x8 path on."""
  import os, subprocess, sys
  if os.environ.get('SEEDHIP_X8') == '7':
    pytest.skip('this IS the child')
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  r = subprocess.run([sys.executable, '-m', 'pytest', os.path.join(root, 'tests', 'test_gpu_kernels.py'), '-q', '-x', '-m', 'gpu',
                      '-k', 'test_x8_gemm_epilogues_and_tails', '-p', 'no:cacheprovider'],
                     env=dict(os.environ, SEEDHIP_X8='7'), capture_output=True, text=True, timeout=600, cwd=root)
  assert r.returncode == 0 and '3 passed' in r.stdout, (r.stdout[-3000:], r.stderr[-2000:])


# Every switch the library / the agents still read from the environment selects a DIFFERENT kernel path than the default
# one; each is exercised once, in a child process (the library reads its switches once), by the tests that cover the
# default path.  ("every knob that stays gets one GPU test in a child process".)
KNOB_CHILDREN = [
    # (environment, test file, -k expression)
    (dict(SEEDHIP_CONV_BF16X6='0'), 'test_gpu_kernels.py', 'conv_fwd_bwd_parity or conv_residual'),   # conv layers back on the fp32 pipe
    (dict(SEEDHIP_X6='0', SEEDHIP_X8='0'), 'test_gpu_kernels.py', 'dense_padded_rows or test_conv_fwd_bwd_parity'),   # Dense layers on gemm.h
    (dict(SEEDHIP_STACK_BF16='0'), 'test_gpu_kernels.py', 'test_stack_conv_parity'),                    # fp32-MFMA first conv
    (dict(SEEDHIP_STACK_W8='0'), 'test_gpu_kernels.py', 'test_stack_conv_parity or test_relu_byte_mask_pair'),  # five-wave first conv (forward: what tensors above 2 GB take)
    (dict(SEEDHIP_STACK_W8='0'), 'test_gpu_store.py', 'fused_inference_matches'),                      # five-wave rows kernel of central inference
    (dict(SEEDHIP_CONVPOOL_MFMA='0'), 'test_gpu_kernels.py', 'test_convpool_fused_parity'),             # round-1 vector-ALU first stage
    (dict(SEEDHIP_RELU_BITS='0'), 'test_gpu_agent.py', ''),                                             # fp32 ReLU masks through the shallow torso
    (dict(SEEDHIP_LSTM_SEQ='0'), 'test_gpu_deep.py', ''),                                               # one launch per LSTM step
]


@pytest.mark.parametrize('k', range(len(KNOB_CHILDREN)))
def test_environment_switches_in_child(k):
  import os, subprocess, sys
  env, fname, expr = KNOB_CHILDREN[k]
  if os.environ.get('SEEDHIP_KNOB_CHILD') == '1':
    pytest.skip('this IS a child')
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  cmd = [sys.executable, '-m', 'pytest', os.path.join(root, 'tests', fname), '-q', '-x', '-m', 'gpu', '-p', 'no:cacheprovider']
  if expr:
    cmd += ['-k', expr]
  r = subprocess.run(cmd, env=dict(os.environ, SEEDHIP_KNOB_CHILD='1', **env), capture_output=True, text=True, timeout=900, cwd=root)
  assert r.returncode == 0 and ' passed' in r.stdout and ' failed' not in r.stdout, (env, r.stdout[-3000:], r.stderr[-2000:])


@pytest.mark.parametrize('T1,B', [(3, 5), (6, 37), (6, 50), (21, 37)])   # (300 / 777 images: the data gradient on wdx.h, byte mask in registers)
def test_relu_byte_mask_pair(device, T1, B):
  """The shallow Atari torso's ReLU mask as bytes (seedhip_conv2d_stack_fwd_bits -> seedhip_conv2d_bwd_data_bits): the
  forward's activation is bit-identical to the plain call, byte [pixel][q] bit r = act[pixel][4 q + r] > 0, and the
  second conv's data gradient through the bytes is bit-identical to the one through the fp32 activation (TF autodiff's
  ReluGrad of atari torso layers 1 -> 2; ragged last tiles: 37 * 6 * 100 super-pixels is not a multiple of 16 * 8)."""
  from seed_rl_amd import ops
  cout, N = 16, T1 * B
  u = synth.atari_unroll(5, T1, B, done_p=0.2, zero_state=False)
  rng = np.random.default_rng(2)
  w = (rng.normal(size=(8, 8, 4, cout)) / 16).astype(np.float32)
  b = (rng.normal(size=cout) * 0.5).astype(np.float32)
  HW = 84 * 84
  ext = torch.zeros((T1 + 3, B, HW), dtype=torch.uint8, device=device)
  ext[3:] = dev(u['frames'].reshape(T1, B, HW), device)
  nv = torch.zeros((T1, B), dtype=torch.uint8, device=device)
  ops.stack_prepare(dev(u['frame_state'], device), dev(u['done'].astype(np.uint8), device), T1, B, HW, ext, nv)
  g0 = ops.StackConvGeom(T1, B, 84, 84, 20, 20, 8, 8, 4, cout, cout)
  g1 = ops.conv_geom(N, 20, 20, 16, 4, 4, 2, 'valid', 32)
  assert ops.conv2d_stack_fwd_bits_supported(g0) and ops.conv2d_bwd_data_bits_supported(g1)
  # geometries the byte-mask kernels do not serve say so (the callers then keep the fp32 mask)
  assert not ops.conv2d_bwd_data_bits_supported(ops.conv_geom(N, 20, 20, 32, 4, 4, 2, 'valid', 64))
  assert not ops.conv2d_bwd_data_bits_supported(ops.conv_geom(N, 9, 9, 64, 3, 3, 1, 'valid', 64))
  assert not ops.conv2d_stack_fwd_bits_supported(ops.StackConvGeom(T1, B, 64, 64, 15, 15, 8, 8, 4, cout, cout))
  wd, bd = dev(w, device), dev(b, device)
  ref = torch.empty((N, 20, 20, cout), device=device)
  ops.conv2d_stack_fwd(g0, ext, nv, wd, bd, ref, out_relu=True)
  out = torch.empty_like(ref)
  bits = torch.full((N, 20, 20, cout // 4), 0xAA, dtype=torch.uint8, device=device)
  ops.conv2d_stack_fwd(g0, ext, nv, wd, bd, out, out_relu=True, relu_bits=bits)
  assert torch.equal(out, ref)
  pos = (ref > 0).reshape(N, 20, 20, cout // 4, 4).to(torch.uint8)
  want = pos[..., 0] | (pos[..., 1] << 1) | (pos[..., 2] << 2) | (pos[..., 3] << 3)
  assert torch.equal(bits, want)
  frac = float(pos.float().mean())
  assert 0.2 < frac < 0.8, frac                       # the mask is not trivially all-ones / all-zeros
  dy = dev(rng.normal(size=(N, 9, 9, 32)).astype(np.float32), device)
  w1 = dev((rng.normal(size=(4, 4, 16, 32)) / 16).astype(np.float32), device)
  dx_ref = torch.full((N, 20, 20, 16), float('nan'), device=device)
  ops.conv2d_bwd_data(g1, dy, w1, dx_ref, relu_mask=ref)
  dx = torch.full((N, 20, 20, 16), float('nan'), device=device)
  ops.conv2d_bwd_data(g1, dy, w1, dx, relu_bits=bits)
  assert torch.equal(dx, dx_ref)
  assert bool((dx[~(ref > 0)] == 0).all())
  with pytest.raises(Exception):                      # no silent fallback for a geometry without the kernel
    ops.conv2d_bwd_data(ops.conv_geom(N, 20, 20, 32, 4, 4, 2, 'valid', 64), dy, w1, dx, relu_bits=bits)


@pytest.mark.parametrize('n', [2100, 2304])
def test_dense_relu_byte_mask_pair(device, n):
  """The same pair one layer up (r5): the second Atari conv's forward also writes the ReLU mask of its output as bytes
  (seedhip_conv2d_fwd_bits, wfx.h), and the Dense layer's data gradient (xgemm.h) reads those bytes -- indexed like
  dX / 4 floats -- instead of the 111 MB fp32 activation.  Bit-identical to the fp32-mask path on both sides."""
  from seed_rl_amd import ops
  rng = np.random.default_rng(n)
  g1 = ops.conv_geom(n, 20, 20, 16, 4, 4, 2, 'valid', 32)
  gfc = ops.dense_geom(n, 2592, 256)
  assert ops.conv2d_fwd_bits_supported(g1) and ops.conv2d_bwd_data_bits_supported(gfc)
  assert not ops.conv2d_fwd_bits_supported(ops.conv_geom(n, 20, 20, 32, 4, 4, 2, 'valid', 64))
  assert not ops.conv2d_bwd_data_bits_supported(ops.dense_geom(64, 2592, 256))          # (below the bf16x6 kernel's row count)
  x = dev(rng.normal(size=(n, 20, 20, 16)).astype(np.float32), device)
  w = dev((rng.normal(size=(4, 4, 16, 32)) / 16).astype(np.float32), device)
  b = dev((rng.normal(size=32) * 0.5).astype(np.float32), device)
  ref = torch.empty((n, 9, 9, 32), device=device)
  ops.conv2d_fwd(g1, x, w, b, ref, out_relu=True)
  out = torch.empty_like(ref)
  bits = torch.full((n, 9, 9, 8), 0xAA, dtype=torch.uint8, device=device)
  ops.conv2d_fwd(g1, x, w, b, out, out_relu=True, relu_bits=bits)
  assert torch.equal(out, ref)
  pos = (ref > 0).reshape(n, 9, 9, 8, 4).to(torch.uint8)
  assert torch.equal(bits, pos[..., 0] | (pos[..., 1] << 1) | (pos[..., 2] << 2) | (pos[..., 3] << 3))
  assert 0.2 < float(pos.float().mean()) < 0.8
  dz = dev(rng.normal(size=(n, 256)).astype(np.float32), device)
  wfc = dev((rng.normal(size=(2592, 256)) / 50).astype(np.float32), device)
  dx_ref = torch.full((n, 2592), float('nan'), device=device)
  ops.conv2d_bwd_data(gfc, dz, wfc, dx_ref, relu_mask=ref.view(n, 2592))
  dx = torch.full((n, 2592), float('nan'), device=device)
  ops.conv2d_bwd_data(gfc, dz, wfc, dx, relu_bits=bits)
  assert torch.equal(dx, dx_ref)
  assert bool((dx[~(ref.view(n, 2592) > 0)] == 0).all())


@pytest.mark.parametrize('n,ih,iw', [(3, 72, 96), (2, 11, 9), (5, 8, 12), (1, 3, 3)])
def test_convpool_fused_parity(device, n, ih, iw):
  """Fused Conv2D(16, 3, 'same')(x/255) + MaxPool2D(3, 2, 'same') of ImpalaDeep's first stage
  (dmlab/networks.py:31-37) vs the oracle conv -> max-pool and torch autograd through both: pooled output,
  and dW / db of the conv for a random gradient on the pooled tensor.  Even, odd and minimum map sizes (TF 'SAME'
  pads differ), partial last band."""
  from seed_rl_amd import ops
  rng = np.random.default_rng(ih * 100 + iw)
  x = rng.integers(0, 256, (n, ih, iw, 3)).astype(np.uint8)
  w = (rng.normal(size=(3, 3, 3, 16)) / np.sqrt(27)).astype(np.float32)
  b = rng.normal(size=(16,)).astype(np.float32)
  wt = torch.tensor(w, requires_grad=True); bt = torch.tensor(b, requires_grad=True)
  a = nets_torch.conv2d(torch.tensor(x).float() / 255, wt, bt, 1, 'same')
  y = nets_torch.max_pool_3x3_s2_same(a)
  dy = rng.normal(size=tuple(y.shape)).astype(np.float32)
  y.backward(torch.tensor(dy))
  ph, pw = (ih + 1) // 2, (iw + 1) // 2
  xd = dev(x, device)
  pooled = torch.full((n, ph, pw, 16), 7.0, device=device)
  arg = torch.full((n, ph, pw, 16), 99, dtype=torch.uint8, device=device)
  ops.conv3x3_u8_pool_fwd(xd, dev(w, device), dev(b, device), pooled, arg)
  yr = y.detach().numpy()
  assert np.max(np.abs(pooled.cpu().numpy() - yr)) <= 2e-5 * max(1.0, np.abs(yr).max())
  assert int(arg.max()) <= 8
  dw = torch.full(w.shape, 7.0, device=device); db = torch.full(b.shape, 7.0, device=device)
  ws = torch.empty(ops.conv3x3_u8_pool_bwd_workspace_bytes(n, ih, iw) // 4 + 1, device=device)
  ops.conv3x3_u8_pool_bwd(xd, dev(dy, device), arg, dw, db, ws)
  gw, gb = wt.grad.numpy(), bt.grad.numpy()
  # a near-tie inside a pooling window may pick a different (equal up to fp32 rounding) maximum than the oracle's
  # summation order does: such a flip moves one dY entry between two pixels, bounded by |dy|max * |x|max per entry
  assert np.max(np.abs(db.cpu().numpy() - gb)) <= 1e-4 * max(1.0, np.abs(gb).max())
  assert np.max(np.abs(dw.cpu().numpy() - gw)) <= 2e-3 * max(1.0, np.abs(gw).max())


@pytest.mark.parametrize('B,H', [(70, 128), (256, 512), (33, 1024), (256, 256)])
def test_lstm_step_fused(device, B, H):
  """One-launch LSTM step (recurrent GEMM + gates + done-reset, csrc/lstm_step.hip) vs the Keras LSTMCell formulas
  in torch fp32 (dmlab/networks.py:152-171): pre-activations, h, and the masked next-step state; ragged last row tile."""
  from seed_rl_amd import ops
  rng = np.random.default_rng(B + H)
  hin = rng.normal(size=(B, H)).astype(np.float32)
  cin = rng.normal(size=(B, H)).astype(np.float32)
  U = (rng.normal(size=(H, 4 * H)) / np.sqrt(H)).astype(np.float32)
  zx = rng.normal(size=(B, 4 * H)).astype(np.float32)
  done_next = (rng.uniform(size=B) < 0.3).astype(np.uint8)
  z_ref = torch.tensor(zx) + torch.tensor(hin) @ torch.tensor(U)
  i, f, g, o = [z_ref[:, k * H:(k + 1) * H] for k in range(4)]
  c_ref = torch.sigmoid(f) * torch.tensor(cin) + torch.sigmoid(i) * torch.tanh(g)
  h_ref = torch.sigmoid(o) * torch.tanh(c_ref)
  keep = torch.tensor(1.0 - done_next.astype(np.float32))[:, None]
  assert ops.lstm_step_supported(B, H)
  up = torch.empty((H, 4 * H), device=device)
  ops.lstm_permute_u(dev(U, device), H, up)
  z = torch.full((B, 4 * H), 7.0, device=device); h = torch.full((B, H), 7.0, device=device)
  hn = torch.full((B, H), 7.0, device=device); cn = torch.full((B, H), 7.0, device=device)
  ops.lstm_step_fwd(dev(hin, device), up, dev(zx, device), dev(cin, device), dev(done_next, device), B, H, z, h, H, hn, cn)
  tol = 2e-5 * max(1.0, float(z_ref.abs().max()))
  assert float((z.cpu() - z_ref).abs().max()) <= tol
  assert float((h.cpu() - h_ref).abs().max()) <= 2e-5
  assert float((hn.cpu() - h_ref * keep).abs().max()) <= 2e-5
  assert float((cn.cpu() - c_ref * keep).abs().max()) <= 5e-5


@pytest.mark.gpu
@pytest.mark.parametrize('T1,B,H', [(7, 256, 512), (5, 70, 256), (3, 32, 128), (121, 256, 512)])
def test_lstm_seq_matches_steps(device, T1, B, H):
  """Whole-unroll LSTM kernel (resident workgroups + grid barrier, csrc/lstm_step.hip lstm_seq_fwd_kernel) is
  BIT-identical to T1 launches of the per-step kernel (same tiling, same summation order), including done-resets,
  a ragged last row tile, and the R2D2 unroll length; the barrier's abort flag stays clear."""
  from seed_rl_amd import ops
  if not ops.lstm_seq_supported(T1, B, H):
    pytest.skip('grid of %d workgroups is not co-resident on this device' % (((B + 31) // 32) * (H // 16)))
  rng = np.random.default_rng(T1 * 1000 + B + H)
  U = dev((rng.normal(size=(H, 4 * H)) / np.sqrt(H)).astype(np.float32), device)
  zx = dev(rng.normal(size=(T1, B, 4 * H)).astype(np.float32), device)
  done = dev((rng.uniform(size=(T1, B)) < 0.1).astype(np.uint8), device)
  h0 = dev(rng.normal(size=(B, H)).astype(np.float32), device)
  c0 = dev(rng.normal(size=(B, H)).astype(np.float32), device)
  up = torch.empty((H, 4 * H), device=device)
  ops.lstm_permute_u(U, H, up)
  outs = []
  for mode in ('step', 'seq'):
    hin = torch.full((T1 + 1, B, H), 7.0, device=device); cin = torch.full((T1 + 1, B, H), 7.0, device=device)
    z = torch.full((T1, B, 4 * H), 7.0, device=device); hout = torch.full((T1, B, H), 7.0, device=device)
    ops.lstm_mask_state(h0, c0, done[0], B, H, hin[0], cin[0])
    if mode == 'step':
      for t in range(T1):
        ops.lstm_step_fwd(hin[t], up, zx[t], cin[t], done[t + 1] if t + 1 < T1 else None, B, H, z[t], hout[t], H,
                          hin[t + 1], cin[t + 1])
    else:
      sync = torch.full((2,), 5, dtype=torch.int32, device=device)
      ops.lstm_seq_fwd(up, zx, done, T1, B, H, z, hout.view(T1 * B, H), H, hin, cin, sync)
      torch.cuda.synchronize()
      assert int(sync[1]) == 0, 'grid barrier timed out'
    outs.append((z, hout, hin, cin))
  for a, b in zip(*outs):
    assert torch.equal(a, b)


@pytest.mark.gpu
def test_lstm_seq_bounded_wait(device, monkeypatch):
  """A producer that never delivers (test hook SEEDHIP_LSTM_SEQ_FAULT) must not hang the GPU: every wait in the
  whole-unroll kernel is bounded, the kernel finishes, and the abort flag reports it."""
  import time
  from seed_rl_amd import ops
  T1, B, H = 4, 64, 128
  if not ops.lstm_seq_supported(T1, B, H):
    pytest.skip('not co-resident')
  rng = np.random.default_rng(3)
  U = dev((rng.normal(size=(H, 4 * H)) / np.sqrt(H)).astype(np.float32), device)
  zx = dev(rng.normal(size=(T1, B, 4 * H)).astype(np.float32), device)
  done = torch.zeros((T1, B), dtype=torch.uint8, device=device)
  up = torch.empty((H, 4 * H), device=device)
  ops.lstm_permute_u(U, H, up)
  hin = torch.zeros((T1 + 1, B, H), device=device); cin = torch.zeros((T1 + 1, B, H), device=device)
  z = torch.empty((T1, B, 4 * H), device=device); hout = torch.empty((T1 * B, H), device=device)
  sync = torch.zeros(2, dtype=torch.int32, device=device)
  monkeypatch.setenv('SEEDHIP_LSTM_SEQ_FAULT', '1')
  torch.cuda.synchronize()
  t0 = time.time()
  ops.lstm_seq_fwd(up, zx, done, T1, B, H, z, hout, H, hin, cin, sync)
  torch.cuda.synchronize()
  assert time.time() - t0 < 20.0
  assert int(sync[1]) == 1
  ring = torch.empty(ops.lstm_seq_bwd_workspace_bytes(B, H) // 4, device=device)
  dz = torch.empty_like(z)
  t0 = time.time()
  ops.lstm_seq_bwd(up, zx, cin, hout, H, done, T1, B, H, dz, ring, sync)
  torch.cuda.synchronize()
  assert time.time() - t0 < 20.0
  assert int(sync[1]) == 1
  monkeypatch.delenv('SEEDHIP_LSTM_SEQ_FAULT')
  ops.lstm_seq_bwd(up, zx, cin, hout, H, done, T1, B, H, dz, ring, sync)
  torch.cuda.synchronize()
  assert int(sync[1]) == 0
  ops.lstm_seq_fwd(up, zx, done, T1, B, H, z, hout, H, hin, cin, sync)
  torch.cuda.synchronize()
  assert int(sync[1]) == 0 and bool(torch.isfinite(hout).all())


@pytest.mark.gpu
@pytest.mark.parametrize('T1,B,H', [(6, 256, 512), (5, 70, 256), (3, 32, 128), (121, 256, 512), (21, 32, 256), (6, 3, 256),
                                    (40, 64, 384)])
def test_lstm_seq_bwd_matches_steps(device, T1, B, H):
  """Whole-recurrence LSTM backward kernel (csrc/lstm_step.hip lstm_seq_bwd_kernel: cell backward + dh_rec = dz U^T
  with the partial sums exchanged through the ring) vs the per-step path (lstm_gates_bwd + dense data gradient):
  same dz up to fp32 summation order, over done-resets, a ragged row tile and the R2D2 unroll length; run twice to
  show it is deterministic and that the ring re-arms correctly."""
  from seed_rl_amd import ops
  if not ops.lstm_seq_supported(T1, B, H):
    pytest.skip('not co-resident on this device')
  rng = np.random.default_rng(T1 * 77 + B + H)
  U = dev((rng.normal(size=(H, 4 * H)) / np.sqrt(H)).astype(np.float32), device)
  Z = dev(rng.normal(size=(T1, B, 4 * H)).astype(np.float32), device)
  Cin = dev(rng.normal(size=(T1 + 1, B, H)).astype(np.float32), device)
  dH = dev(rng.normal(size=(T1 * B, H)).astype(np.float32), device)
  done = dev((rng.uniform(size=(T1, B)) < 0.1).astype(np.uint8), device)
  up = torch.empty((H, 4 * H), device=device)
  ops.lstm_permute_u(U, H, up)
  # per-step reference path
  dZ_ref = torch.empty((T1, B, 4 * H), device=device)
  dcb = [torch.empty((B, H), device=device), torch.empty((B, H), device=device)]
  dhb = torch.empty((B, H), device=device)
  gu = ops.dense_geom(B, H, 4 * H)
  dh_rec = dc_rec = None
  dH3 = dH.view(T1, B, H)
  for t in range(T1 - 1, -1, -1):
    ops.lstm_gates_bwd(Z[t], Cin[t], dH3[t], H, dh_rec, dc_rec, done[t + 1] if t + 1 < T1 else None, B, H, dZ_ref[t],
                       dcb[t & 1])
    dc_rec = dcb[t & 1]
    if t > 0:
      ops.conv2d_bwd_data(gu, dZ_ref[t], U, dhb)
      dh_rec = dhb
  ring = torch.empty(ops.lstm_seq_bwd_workspace_bytes(B, H) // 4, device=device)
  outs = []
  for _ in range(4):
    dZ = torch.full((T1, B, 4 * H), 7.0, device=device)
    sync = torch.full((2,), 5, dtype=torch.int32, device=device)
    ops.lstm_seq_bwd(up, Z, Cin, dH, H, done, T1, B, H, dZ, ring, sync)
    torch.cuda.synchronize()
    assert int(sync[1]) == 0, 'a wait timed out'
    outs.append(dZ)
  for o in outs[1:]:
    assert torch.equal(outs[0], o)
  scale = float(dZ_ref.abs().max())
  assert float((outs[0] - dZ_ref).abs().max()) <= 2e-5 * max(1.0, scale)


@pytest.mark.parametrize('rows,feat,A', [(1000, 256, 18), (37, 512, 6), (16, 64, 31), (5, 256, 9), (10752, 256, 18),
                                         (1, 128, 3)])
def test_heads_fwd(device, rows, feat, A):
  """csrc/heads.hip: the packed policy / baseline heads (dmlab/networks.py:116-124) as one skinny GEMM, against float64
  NumPy: y = x W + b.  Tolerance: fp32 accumulation over `feat` (1e-5 of the output scale)."""
  from seed_rl_amd import ops
  ldh = (A + 1 + 3) // 4 * 4
  assert ops.heads_supported(feat, ldh)
  rng = np.random.default_rng(rows + feat)
  x = np.maximum(rng.normal(size=(rows, feat)), 0).astype(np.float32)           # a ReLU output (about half zeros)
  w = np.zeros((feat, ldh), np.float32); w[:, :A + 1] = rng.normal(size=(feat, A + 1)) / np.sqrt(feat)
  b = np.zeros(ldh, np.float32); b[:A + 1] = rng.normal(size=A + 1)
  t = lambda a: torch.as_tensor(a).to(device)
  y = torch.empty((rows, ldh), device=device)
  ops.heads_fwd(t(x), feat, t(w), t(b), rows, feat, ldh, y)
  want = x.astype(np.float64) @ w.astype(np.float64) + b
  np.testing.assert_allclose(y.cpu().numpy(), want, rtol=0, atol=1e-5 * max(1.0, np.abs(want).max()))
  assert not ops.heads_supported(100, 20) and not ops.heads_supported(256, 36) and not ops.heads_supported(1024, 20)


_W8_CHILD = r'''
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
from tests import synth
from seed_rl_amd import ops
dev = torch.device('cuda')
out = {}
for T1, B, cout in ((21, 37, 16), (6, 50, 32), (1, 64, 16)):
  u = synth.atari_unroll(5, T1, B, done_p=0.2, zero_state=False)
  rng = np.random.default_rng(T1 * 100 + B)
  w = torch.tensor((rng.normal(size=(8, 8, 4, cout)) / 16).astype(np.float32), device=dev)
  b = torch.tensor((rng.normal(size=cout) * 0.5).astype(np.float32), device=dev)
  HW = 84 * 84
  ext = torch.zeros((T1 + 3, B, HW), dtype=torch.uint8, device=dev)
  ext[3:] = torch.tensor(u['frames'].reshape(T1, B, HW), device=dev)
  nv = torch.zeros((T1, B), dtype=torch.uint8, device=dev)
  ops.stack_prepare(torch.tensor(u['frame_state'], device=dev), torch.tensor(u['done'].astype(np.uint8), device=dev), T1, B, HW, ext, nv)
  g = ops.StackConvGeom(T1, B, 84, 84, 20, 20, 8, 8, 4, cout, cout)
  y = torch.empty((T1 * B, 20, 20, cout), device=dev)
  if ops.conv2d_stack_fwd_bits_supported(g):
    bits = torch.zeros((T1 * B, 20, 20, cout // 4), dtype=torch.uint8, device=dev)
    ops.conv2d_stack_fwd(g, ext, nv, w, b, y, out_relu=True, relu_bits=bits)
    out['bits_%d_%d' % (T1, B)] = bits.cpu().numpy()
  else:
    ops.conv2d_stack_fwd(g, ext, nv, w, b, y, out_relu=True)
  out['y_%d_%d' % (T1, B)] = y.cpu().numpy()
np.savez(sys.argv[2], **out)
'''


def test_stack_fwd_eight_waves_bit_identical_to_five_waves(tmp_path):
  """stackconv_fwd_w8_kernel (r6: eight waves, two batch columns per workgroup) against the five-wave kernel it replaced
  (SEEDHIP_STACK_W8=0; also what tensors above 2 GB take): same operands and MFMA order per accumulator, so the
  activations and the ReLU byte masks are BIT-identical -- odd B (a lone column in the last pair), several time chunks,
  two 16-channel slices, T1 = 1.  One child process per build switch (the library reads it once)."""
  import os, subprocess, sys
  if os.environ.get('SEEDHIP_KNOB_CHILD') == '1':
    pytest.skip('this IS a child')
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  res = {}
  for w8 in ('1', '0'):
    path = str(tmp_path / ('w8_%s.npz' % w8))
    r = subprocess.run([sys.executable, '-c', _W8_CHILD, root, path], env=dict(os.environ, SEEDHIP_STACK_W8=w8),
                       capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    res[w8] = np.load(path)
  assert sorted(res['1'].files) == sorted(res['0'].files) and len(res['1'].files) >= 5
  for k in res['1'].files:
    a, b = res['1'][k], res['0'][k]
    assert a.shape == b.shape and a.tobytes() == b.tobytes(), k
    if k.startswith('y_'):
      assert float(np.abs(a).max()) > 0.1 and float((a > 0).mean()) > 0.1, k
