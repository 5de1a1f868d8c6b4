#!/usr/bin/env python
"""Generates tests/golden/*.npz by EXECUTING THE REFERENCE'S OWN CODE from /root/reference:

  * common/vtrace.py `from_importance_weights`                      (run on tests/golden/tf_shim.py)
  * tests/vtrace_test.py `_ground_truth_calculation`, `_shaped_arange`  (pure NumPy, extracted by ast)
  * agents/r2d2/learner.py `value_function_rescaling`, `inverse_value_function_rescaling`,
    `n_step_bellman_target`, `compute_loss_and_priorities_from_agent_outputs`   (extracted by ast, tf_shim)

TensorFlow / absl are not installable in the build container, so the modules cannot be imported as they are; the
functions above are pure math on a few elementwise TF ops, which tf_shim maps to float32 NumPy.  No reference
source is copied into this repository: this script reads it where it lies and only its OUTPUTS are committed.

  python tests/golden/make_golden.py            # needs /root/reference (build container only)
"""
import ast
import collections
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import tf_shim  # noqa: E402
from tests import synth  # noqa: E402

REF = '/root/reference'


def extract(path, names, namespace):
  """exec()s the named top-level functions of a reference file inside `namespace`."""
  tree = ast.parse(open(path).read())
  body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
  assert len(body) == len(names), (path, names)
  exec(compile(ast.Module(body=body, type_ignores=[]), path, 'exec'), namespace)   # pylint: disable=exec-used
  return namespace


def main():
  out = {}
  # ---- the reference V-trace itself ----
  sys.modules['tensorflow'] = tf_shim
  src = open(os.path.join(REF, 'common/vtrace.py')).read()
  mod = types.ModuleType('ref_vtrace')
  exec(compile(src, 'common/vtrace.py', 'exec'), mod.__dict__)                     # pylint: disable=exec-used
  cases = []
  for seed in (0, 1, 2):
    for stress in (False, True):
      for kw in (dict(), dict(lambda_=0.95), dict(clip_rho_threshold=3.7, clip_pg_rho_threshold=2.2),
                 dict(clip_rho_threshold=None, clip_pg_rho_threshold=None)):
        cases.append((seed, stress, kw))
  for n, (seed, stress, kw) in enumerate(cases):
    inp = synth.vtrace_inputs(seed, 20, 32, 6, stress=stress)
    r = mod.from_importance_weights(**inp, **kw)
    out['vtrace_%02d_vs' % n] = r.vs.numpy().astype(np.float32)
    out['vtrace_%02d_pg' % n] = r.pg_advantages.numpy().astype(np.float32)
    out['vtrace_%02d_meta' % n] = np.array([seed, int(stress), kw.get('lambda_', 1.0),
                                            -1 if kw.get('clip_rho_threshold', 1.0) is None else kw.get('clip_rho_threshold', 1.0),
                                            -1 if kw.get('clip_pg_rho_threshold', 1.0) is None else kw.get('clip_pg_rho_threshold', 1.0)],
                                           np.float64)
  out['vtrace_num_cases'] = np.array(len(cases))
  # ---- the reference test's O(T^2) ground truth on the reference test's inputs (tests/vtrace_test.py:120-145) ----
  ns = extract(os.path.join(REF, 'tests/vtrace_test.py'), ['_shaped_arange', '_ground_truth_calculation'],
               {'np': np, 'collections': collections, 'vtrace': mod})
  seq_len, batch = 5, 5
  log_rhos = ns['_shaped_arange'](seq_len, batch) / (batch * seq_len)
  log_rhos = 5 * (log_rhos - 0.5)
  values = {
      'behaviour_action_log_probs': np.zeros_like(log_rhos), 'target_action_log_probs': log_rhos,
      'discounts': np.array([[0.9 / (b + 1) for b in range(batch)] for _ in range(seq_len)]),
      'rewards': ns['_shaped_arange'](seq_len, batch), 'values': ns['_shaped_arange'](seq_len, batch) / batch,
      'bootstrap_value': ns['_shaped_arange'](batch) + 1.0, 'clip_rho_threshold': 3.7, 'clip_pg_rho_threshold': 2.2}
  gt = ns['_ground_truth_calculation'](**values)
  out['reftest_vs'] = np.asarray(gt.vs, np.float64)
  out['reftest_pg'] = np.asarray(gt.pg_advantages, np.float64)
  r = mod.from_importance_weights(**values)
  out['reftest_vs_tf32'] = r.vs.numpy()
  out['reftest_pg_tf32'] = r.pg_advantages.numpy()
  # ---- R2D2 loss math ----
  flags = types.SimpleNamespace(value_function_rescaling_epsilon=1e-3, n_steps=5)
  ns = extract(os.path.join(REF, 'agents/r2d2/learner.py'),
               ['value_function_rescaling', 'inverse_value_function_rescaling', 'n_step_bellman_target',
                'compute_loss_and_priorities_from_agent_outputs'], {'tf': tf_shim, 'FLAGS': flags})
  x = np.concatenate([np.linspace(-300, 300, 41), [0.0, 1e-3, -1e-3]]).astype(np.float32)
  out['r2d2_h_x'] = x
  out['r2d2_h'] = ns['value_function_rescaling'](tf_shim.Tensor(x)).numpy()
  out['r2d2_hinv'] = ns['inverse_value_function_rescaling'](tf_shim.Tensor(x)).numpy()
  AO = collections.namedtuple('AO', 'action q_values')
  EO = collections.namedtuple('EO', 'reward done')
  for n, (T, B, A, nsteps) in enumerate([(10, 8, 6, 5), (81, 5, 18, 5), (6, 4, 3, 2)]):
    rng = np.random.default_rng(100 + n)
    tq = rng.uniform(0, 1, (T, B, A)).astype(np.float32)
    gq = (rng.uniform(0, 1, (T, B, A)) * 3).astype(np.float32)
    act = rng.integers(0, A, (T, B)).astype(np.int32)
    rew = rng.normal(size=(T, B)).astype(np.float32)
    done = rng.uniform(size=(T, B)) < 0.1
    flags.n_steps = nsteps
    loss, prio = ns['compute_loss_and_priorities_from_agent_outputs'](
        AO(tf_shim.Tensor(tq.argmax(-1).astype(np.int32)), tf_shim.Tensor(tq)), AO(None, tf_shim.Tensor(gq)),
        EO(tf_shim.Tensor(rew), tf_shim.Tensor(done)), AO(tf_shim.Tensor(act), None), 0.997)
    for k, v in (('tq', tq), ('gq', gq), ('act', act), ('rew', rew), ('done', done), ('loss', loss.numpy()),
                 ('prio', prio.numpy()), ('nsteps', np.array(nsteps))):
      out['r2d2_%d_%s' % (n, k)] = v
    bt = ns['n_step_bellman_target'](tf_shim.Tensor(rew), tf_shim.Tensor(done), tf_shim.Tensor(gq[..., 0]), 0.997, nsteps)
    out['r2d2_%d_nstep' % n] = bt.numpy()
  np.savez_compressed(os.path.join(HERE, 'reference_outputs.npz'), **out)
  print('wrote %d arrays to tests/golden/reference_outputs.npz' % len(out))


if __name__ == '__main__':
  main()
