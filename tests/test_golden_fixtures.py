"""Parity against OUTPUTS OF THE REFERENCE'S OWN CODE (tests/golden/reference_outputs.npz, produced by
tests/golden/make_golden.py executing /root/reference/common/vtrace.py, tests/vtrace_test.py's NumPy ground truth
and the R2D2 target/loss functions of agents/r2d2/learner.py on a float32 NumPy stand-in for the few TF ops they use).
CPU tests pin the oracle; GPU tests pin the HIP kernels (through the C ABI) to the same vectors."""
import os

import numpy as np
import pytest
import torch

from oracle import r2d2_np, vtrace_np
from tests import synth

G = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'reference_outputs.npz'))


def _vtrace_cases():
  for n in range(int(G['vtrace_num_cases'])):
    seed, stress, lam, cr, cp = G['vtrace_%02d_meta' % n]
    kw = dict(lambda_=float(lam), clip_rho_threshold=None if cr < 0 else float(cr),
              clip_pg_rho_threshold=None if cp < 0 else float(cp))
    yield n, synth.vtrace_inputs(int(seed), 20, 32, 6, stress=bool(stress)), kw


def test_oracle_vtrace_matches_reference_code():
  worst = 0.0
  for n, inp, kw in _vtrace_cases():
    r = vtrace_np.from_importance_weights(**inp, **kw)
    worst = max(worst, np.abs(r.vs - G['vtrace_%02d_vs' % n]).max(), np.abs(r.pg_advantages - G['vtrace_%02d_pg' % n]).max())
  assert worst <= 1e-6, worst            # same fp32 op order as common/vtrace.py:110-144


def test_reference_fp32_vs_its_own_ground_truth():
  """The reference's test compares common/vtrace.py with an O(T^2) NumPy formula (tests/vtrace_test.py:41-82,
  120-145, assertAllClose): reproduce that comparison and pin our oracle to both."""
  np.testing.assert_allclose(G['reftest_vs_tf32'], G['reftest_vs'], rtol=1e-6, atol=1e-5)
  np.testing.assert_allclose(G['reftest_pg_tf32'], G['reftest_pg'], rtol=1e-6, atol=1e-5)
  assert abs(float(G['reftest_vs'].sum()) - 930.41754) < 1e-3


def test_oracle_r2d2_matches_reference_code():
  np.testing.assert_allclose(r2d2_np.value_function_rescaling(G['r2d2_h_x']), G['r2d2_h'], rtol=1e-6, atol=1e-6)
  np.testing.assert_allclose(r2d2_np.inverse_value_function_rescaling(G['r2d2_h_x']), G['r2d2_hinv'], rtol=2e-5, atol=1e-5)
  for n in range(3):
    g = lambda k: G['r2d2_%d_%s' % (n, k)]
    ns = int(g('nsteps'))
    np.testing.assert_allclose(r2d2_np.n_step_bellman_target(g('rew'), g('done'), g('gq')[..., 0], 0.997, ns), g('nstep'),
                               rtol=1e-6, atol=1e-6)
    loss, prio, _ = r2d2_np.loss_and_priorities(g('tq'), g('gq'), g('rew'), g('done'), g('act'), 0.997, ns)
    np.testing.assert_allclose(loss, g('loss'), rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(prio, g('prio'), rtol=2e-5, atol=1e-6)


@pytest.mark.gpu
def test_hip_vtrace_matches_reference_code(device):
  """BASELINE metric 'V-trace fp32 max-abs-err vs ref': HIP kernel vs the reference code's outputs, bar 1e-5."""
  from seed_rl_amd import vtrace
  worst_learner = 0.0
  for n, inp, kw in _vtrace_cases():
    out = vtrace.from_importance_weights(**{k: torch.as_tensor(v).to(device) for k, v in inp.items()}, **kw)
    for got, ref in ((out.vs, G['vtrace_%02d_vs' % n]), (out.pg_advantages, G['vtrace_%02d_pg' % n])):
      err = np.abs(got.cpu().numpy() - ref).max()
      # 1e-5 absolute; unclipped importance weights (rho up to e^2.5) produce |vs| in the hundreds, where one
      # fp32 ulp already exceeds 1e-5: those cases are held to 1e-6 relative instead
      assert err <= max(1e-5, 1e-6 * np.abs(ref).max()), (n, kw, err, np.abs(ref).max())
      if kw['clip_rho_threshold'] == 1.0:
        worst_learner = max(worst_learner, err)
  assert worst_learner <= 1e-5          # the configuration the learner uses (learner.py:101-108)


@pytest.mark.gpu
def test_hip_r2d2_loss_matches_reference_code(device):
  from seed_rl_amd import ops
  for n in range(3):
    g = lambda k: G['r2d2_%d_%s' % (n, k)]
    tq, ns = g('tq'), int(g('nsteps'))
    T, B, A = tq.shape
    dev = lambda a: torch.as_tensor(np.ascontiguousarray(a)).to(device)
    loss_b = torch.empty(B, device=device); prio = torch.empty(B, device=device); total = torch.empty(1, device=device)
    dq = torch.empty((T, B, A), device=device)
    ws = torch.empty(ops.r2d2_loss_workspace_bytes(T, B, ns) // 4 + 4, device=device)
    ops.r2d2_loss_fwd_bwd(dev(tq), dev(g('gq')), dev(g('act').astype(np.int32)), dev(g('rew')),
                          dev(g('done').astype(np.uint8)), None, T, B, A, 0.997, ns, 0.9, 1e-3, B, loss_b, prio, dq, total, ws)
    np.testing.assert_allclose(loss_b.cpu().numpy(), g('loss'), rtol=3e-5, atol=1e-6)
    np.testing.assert_allclose(prio.cpu().numpy(), g('prio'), rtol=3e-5, atol=1e-6)
