"""Train-step parity at the BASELINE configs' REAL sizes: the grid-size-dependent dispatch
(gemm.h cost-model tiles / split-K, the persistent ws_fast grid, lstm_seq residency) is exercised at the shapes the
bench runs, and compared with the torch-CPU oracle of the same graph (reference agents/vtrace/learner.py:73-159,
255-280; agents/r2d2/learner.py:333-384, 572-636).

Tolerances.  Both sides compute in fp32 with different summation orders (MFMA tiles + split-K slices vs oneDNN's
blocked accumulation), and at these sizes two things separate them that are not defects (r02 measurements, cfg2 T=20
B=512, tools/diag_parity.py): fp32 re-association moves a conv weight gradient (a sum of ~1e6 largely cancelling terms)
by ~4e-4 of its tensor's maximum on EITHER side, and ONE of the 2.75 M Dense pre-activations is +1.3e-7 in fp64 but <= 0
on the GPU -- its ReLU mask flips, which moves column 18 of fc/kernel by 3.3e-3 (the loss still agrees to the last bit,
the logits to 5e-7).  So every test also evaluates the SAME oracle graph in fp64 (oracle/nets_torch.float64_truth) and
holds the HIP gradients, for the headline configuration, to
  99th percentile per tensor of |g - g64| / max|g64|  <= 1e-3 AND <= 3x the fp32 oracle's own + 1e-4
  maximum                                             <= 1e-2 (what a handful of mask / arg-max flips can move),
and the HIP-vs-fp32-oracle maximum to the sum of both maxima against fp64 (+10 %).  (The fp64 evaluation takes ~45 s of
host time per case -- oneDNN has no fp64 convolution -- so the other cases compare with the fp32 oracle only: 99th
percentile <= 1.5e-3, maximum <= 1e-2.)  Loss 1e-4 relative; parameters after one Adam step: at most 1e-3 of the elements
further than 5e-5 from the fp32 oracle and none further than 2.2 lr (beta_1 = 0 normalises every element's update to ~lr,
so an element whose gradient sits at the fp32 noise floor can flip sign).
"""
import pytest

from tests import parity

pytestmark = pytest.mark.gpu
LR = 4.8e-4


def _show(r):
  return {k: v for k, v in r.items() if k != 'grad_rel_err'}


def _record(name, r):
  """Prints the numbers (pytest -s / the captured log) and, when gpurun_out/ exists, keeps them as JSON."""
  import json
  import os
  line = json.dumps({name: _show(r)})
  print(line)
  d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
  if os.path.isdir(d):
    with open(os.path.join(d, 'fullsize_parity.jsonl'), 'a') as f:
      f.write(line + '\n')


def _check_params(r):
  assert r['param_frac_gt_5e5'] <= 1e-3, _show(r)
  assert r['param_max_abs_err'] <= 2.2 * LR, _show(r)


def _check_grads_fp64(r):
  assert r['grad_q99_rel_err_vs_fp64'] <= 1e-3, _show(r)
  # per tensor: HIP's q99 distance to the fp64 evaluation <= 1.25 x the fp32 oracle's own + 5e-5 (r4: 3x + 1e-4 on the
  # worst tensors only, three times looser than what was measured)
  assert r['grad_q99_gate_vs_fp64'] <= 1.0, _show(r)
  # bias vectors: the same formula on the 90th percentile (tests/parity.py: the q99 of 32 elements is their maximum, and
  # the r5 "bias deviation" was one ReLU-tie channel; r6 diagnosis in DESIGN.md section 8)
  assert r['grad_q99_gate_bias_vs_fp64'] <= 1.0, _show(r)
  assert r['grad_max_rel_err_vs_fp64'] <= 1e-2, _show(r)
  assert r['grad_max_rel_err'] <= 1.1 * (r['grad_max_rel_err_vs_fp64'] + r['oracle_grad_max_rel_err_vs_fp64']) + 1e-6, \
      _show(r)


def _check_grads_fp32(r, q99=1.5e-3, mx=1e-2):
  assert r['grad_q99_rel_err'] <= q99, _show(r)
  assert r['grad_max_rel_err'] <= mx, _show(r)


def test_cfg2_atari_T20_B512_A18(device):
  """BASELINE configs[1] exactly as bench.py runs it: T=20, B=512, A=18, flag-default loss."""
  r = parity.atari_step(device, T1=21, B=512, A=18)
  assert r['loss_rel_err'] <= 1e-4, _show(r)
  assert r['logits_max_abs_err'] <= 2e-4 and r['baseline_max_abs_err'] <= 2e-4, _show(r)
  _check_grads_fp64(r)
  _check_params(r)


def test_cfg2_atari_T20_B512_lambda_kl_clip(device):
  """Same size, every loss term on (lambda 0.95 as gcp/train_dmlab_*.sh, kl_cost, reward clipping), non-zero
  frame-stacking state, more episode ends."""
  r = parity.atari_step(device, T1=21, B=512, A=18, seed=11, done_p=0.05, zero_state=False,
                        loss_kw=dict(lambda_=0.95, kl_cost=0.05, max_abs_reward=1.0), truth=False)
  assert r['loss_rel_err'] <= 1e-4, _show(r)
  _check_grads_fp32(r)
  _check_params(r)


def test_cfg3_dmlab_T20_B16(device):
  """BASELINE configs[2] at T=20 with B=16 columns (the oracle's ~100 GFLOP; B=256 differs only in the grid)."""
  r = parity.deep_step(device, T1=21, B=16, A=9, truth=False)
  assert r['loss_rel_err'] <= 2e-4, _show(r)
  assert r['logits_max_abs_err'] <= 3e-4 and r['baseline_max_abs_err'] <= 3e-4, _show(r)
  assert r['grad_max_rel_err_post_pool'] <= 1e-3, _show(r)
  assert r['grad_max_rel_err'] <= 1e-2, _show(r)
  assert r['param_max_abs_err'] <= 2.2 * LR, _show(r)


def test_cfg5_r2d2_T120_burnin40_B4(device):
  """BASELINE configs[4] at its real sequence length: T=120 (121 steps), burn-in 40, n-step 5, both networks."""
  r = parity.r2d2_step(device, T1=121, B=4, A=18, burn_in=40, truth=False)
  assert r['loss_rel_err'] <= 2e-4, _show(r)
  assert r['q_max_abs_err'] <= 5e-4, _show(r)
  assert r['priority_max_rel_err'] <= 2e-3, _show(r)
  _check_grads_fp32(r, q99=1e-3, mx=2e-3)
  assert r['grad_norm_rel_err'] <= 1e-3, _show(r)
  assert r['param_max_abs_err'] <= 5e-5, _show(r)


def test_cfg3_dmlab_T20_B256(device):
  """BASELINE configs[2] exactly as bench.py runs it: T=20, B=256, A=9 (dmlab/networks.py:135-171).  This is the shape
  at which the halo kernels' persistent grids, the gemm.h cost-model plans and the LSTM sequence kernels (8 row tiles x
  32 workgroups) are benched; the fp32 oracle of the same graph takes about a minute of host time."""
  import os
  # the fp64 evaluation of the same graph (oneDNN has no fp64 convolution: ~3 minutes of host time) says how far the
  # fp32 ORACLE itself is from the truth on these 3.7e7-term sums; SEEDHIP_SKIP_FP64=1 skips it (builder's quick runs)
  truth = os.environ.get('SEEDHIP_SKIP_FP64', '0') != '1'
  r = parity.deep_step(device, T1=21, B=256, A=9, truth=truth)
  _record('cfg3_dmlab_T20_B256', r)
  assert r['loss_rel_err'] <= 2e-4, _show(r)
  assert r['logits_max_abs_err'] <= 3e-4 and r['baseline_max_abs_err'] <= 3e-4, _show(r)
  # HIP against the fp32 ORACLE: two fp32-accurate evaluations of 3.7e7-term sums differ by up to the sum of their distances
  # to the truth (triangle inequality).  With the fp64 evaluation present the bound is that sum (+10 %); r4, 3x3 layers
  # on the bf16 pipe: HIP 1.15e-3 and oracle 1.10e-3 from fp64 at q99, 2.13e-3 from each other (1.3-1.45e-3 between the
  # fp32-MFMA kernels and the oracle).  The decisive gate is the fp64 one below
  if truth:
    assert r['grad_q99_rel_err'] <= max(1.5e-3, 1.1 * (r['grad_q99_rel_err_vs_fp64'] + r['oracle_grad_q99_rel_err_vs_fp64'])), _show(r)
  else:
    assert r['grad_q99_rel_err'] <= 2.5e-3, _show(r)
  assert r['grad_max_rel_err_post_pool'] <= 2e-3, _show(r)
  assert r['grad_max_rel_err'] <= 1e-2, _show(r)
  if truth:
    # the gate, PER TENSOR: against fp64 the HIP gradient's q99 distance may exceed the fp32 oracle's own by a quarter
    # (+5e-5) at most (r4: 3x + 1e-4 on the worst tensors; measured 1.05x)
    assert r['grad_q99_rel_err_vs_fp64'] <= 1.5e-3, _show(r)
    assert r['grad_q99_gate_vs_fp64'] <= 1.0, _show(r)
    assert r['grad_q99_gate_bias_vs_fp64'] <= 1.0, _show(r)           # (bias vectors: the same bound on their q90, tests/parity.py)
    assert r['grad_max_rel_err_vs_fp64'] <= 1e-2, _show(r)
  _check_params(r)


def test_cfg5_r2d2_T120_burnin40_B256(device):
  """BASELINE configs[4] exactly as bench.py runs it: T=120 (121 steps), burn-in 40, n-step 5, B=256, both networks
  (agents/r2d2/learner.py:333-384,572-636)."""
  r = parity.r2d2_step(device, T1=121, B=256, A=18, burn_in=40, truth=False)
  _record('cfg5_r2d2_T120_B256', r)
  assert r['loss_rel_err'] <= 2e-4, _show(r)
  assert r['q_max_abs_err'] <= 5e-4, _show(r)
  assert r['priority_max_rel_err'] <= 2e-3, _show(r)
  _check_grads_fp32(r, q99=1e-3, mx=1e-2)
  assert r['grad_norm_rel_err'] <= 1e-3, _show(r)
  assert r['param_max_abs_err'] <= 1e-4, _show(r)
