"""seed_rl_amd -- MI355X (gfx950) native SEED-RL learner hot path.

Host-side mirror (Python, like the reference) of the reference's learner
interfaces for the path named in BASELINE.json: `vtrace.from_importance_weights`,
`parametric_distribution`, `learner.compute_loss` / train step, the agents'
unroll -- all backed by hand-written HIP kernels behind the C ABI in
include/seedhip.h (libseedhip.so).  There is NO CPU fallback: importing the
kernels without the built library raises.
"""
__version__ = '0.1.0'
