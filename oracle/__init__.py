"""CPU oracle for the SEED-RL learner hot path.  TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (NumPy / plain C / PyTorch-CPU fp32) of the
reference algorithm for the path named in BASELINE.json:north_star.  It exists
to *check* the HIP product path.  Only `tests/`, `__graft_entry__.smoke()` and
the `cpu_baseline` leg of `bench.py` may import it.  Nothing under
`seed_rl_amd/` imports it, and the product path raises if the HIP library is
missing instead of falling back to anything in here.

Parity pinning (see DESIGN.md "Oracle"):
  * vtrace / categorical log-prob: pinned by the reference's own known-answer
    tests (tests/vtrace_test.py:41-82,88-115,120-145), restated in
    tests/test_oracle_golden.py.
  * stack_frames / _unroll_cell: pinned by atari/networks_test.py:119-247.
  * n-step Bellman target / value rescaling: pinned by
    agents/r2d2/learner_test.py:114-198.
  * UnrollStore / Aggregator / make_time_major / batch_apply (oracle/utils_np.py): pinned by
    tests/utils_test.py:70-301,585-606 (tests/test_oracle_utils.py).
  * additionally, tests/golden/reference_outputs.npz holds OUTPUTS OF THE REFERENCE'S OWN CODE
    (common/vtrace.py, tests/vtrace_test.py ground truth, agents/r2d2/learner.py loss math) produced by
    tests/golden/make_golden.py; tests/test_golden_fixtures.py pins this oracle and the HIP kernels to them.
  * Conv2D / MaxPool2D / LSTMCell / Dense / Adam numerics live in TensorFlow
    2.4.1 + Keras (un-vendored; not importable here: no tensorflow).  They are
    restated from the published Keras semantics.  The
    reference tests pin only structure (39 trainable tensors,
    tests/agents_test.py:45; LSTM input width, atari/networks_test.py:105-117)
    => numerical parity of the network forward/backward and of Adam is
    "PARITY UNPINNED" (self-consistent HIP-vs-oracle only).
"""
