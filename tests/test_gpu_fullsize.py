"""Train-step parity at the BASELINE configs' REAL sizes: the grid-size-dependent dispatch
(gemm.h cost-model tiles / split-K, the persistent ws_fast grid, lstm_seq residency) is exercised at the shapes the
bench runs, and compared with the torch-CPU oracle of the same graph (reference agents/vtrace/learner.py:73-159,
255-280; agents/r2d2/learner.py:333-384, 572-636).

Tolerances (fp32 on both sides, different summation order on MFMA vs oneDNN):
  loss 1e-4 relative; gradients 3e-4 (feed-forward Atari), 1e-3 (LSTM agents) of each tensor's max, 1e-2 for layers
  upstream of a max-pool (argmax routing is discrete: tests/test_gpu_deep.py); parameters after one Adam step: at
  most 1e-4 of the elements further than 5e-5 from the oracle and none further than 2.2 lr (beta_1 = 0 normalises every
  element's update to ~lr, so an element whose gradient sits at the fp32 noise floor can flip sign).
"""
import pytest

from tests import parity

pytestmark = pytest.mark.gpu
LR = 4.8e-4


def _check_params(r):
  assert r['param_frac_gt_5e5'] <= 1e-4, r
  assert r['param_max_abs_err'] <= 2.2 * LR, r


def test_cfg2_atari_T20_B512_A18(device):
  """BASELINE configs[1] exactly as bench.py runs it: T=20, B=512, A=18, flag-default loss."""
  r = parity.atari_step(device, T1=21, B=512, A=18)
  assert r['loss_rel_err'] <= 1e-4, r
  assert r['logits_max_abs_err'] <= 2e-4 and r['baseline_max_abs_err'] <= 2e-4, r
  assert r['grad_max_rel_err'] <= 3e-4, r
  _check_params(r)


def test_cfg2_atari_T20_B512_lambda_kl_clip(device):
  """Same size, every loss term on (lambda 0.95 as gcp/train_dmlab_*.sh, kl_cost, reward clipping), non-zero
  frame-stacking state, more episode ends."""
  r = parity.atari_step(device, T1=21, B=512, A=18, seed=11, done_p=0.05, zero_state=False,
                        loss_kw=dict(lambda_=0.95, kl_cost=0.05, max_abs_reward=1.0))
  assert r['loss_rel_err'] <= 1e-4, r
  assert r['grad_max_rel_err'] <= 3e-4, r
  _check_params(r)


def test_cfg3_dmlab_T20_B16(device):
  """BASELINE configs[2] at T=20 with B=16 columns (the oracle's ~100 GFLOP; B=256 differs only in the grid)."""
  r = parity.deep_step(device, T1=21, B=16, A=9)
  assert r['loss_rel_err'] <= 2e-4, r
  assert r['logits_max_abs_err'] <= 3e-4 and r['baseline_max_abs_err'] <= 3e-4, r
  assert r['grad_max_rel_err_post_pool'] <= 1e-3, r
  assert r['grad_max_rel_err'] <= 1e-2, r
  assert r['param_max_abs_err'] <= 2.2 * LR, r


def test_cfg5_r2d2_T120_burnin40_B4(device):
  """BASELINE configs[4] at its real sequence length: T=120 (121 steps), burn-in 40, n-step 5, both networks."""
  r = parity.r2d2_step(device, T1=121, B=4, A=18, burn_in=40)
  assert r['loss_rel_err'] <= 2e-4, r
  assert r['q_max_abs_err'] <= 5e-4, r
  assert r['priority_max_rel_err'] <= 2e-3, r
  assert r['grad_max_rel_err'] <= 2e-3, r
  assert r['grad_norm_rel_err'] <= 1e-3, r
  assert r['param_max_abs_err'] <= 5e-5, r
