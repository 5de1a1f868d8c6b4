"""The bench line's bookkeeping that needs no GPU: the whole-step roofline (VERDICT r3 task 4) and the kernel-source
digest that decides whether a committed PMC profile still describes this tree's kernels."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_step_roofline_sums_per_kernel_floors():
  bench = importlib.import_module('bench')
  kern = {
      # 100 MB at 6.3 TB/s = 15.9 us > 1 GF at 157.3 TF/s = 6.4 us: HBM-bound, launched twice
      'a': dict(calls=2, total_ms=0.08, avg_ms=0.04, flops=1e9, bytes=100e6, pipe='f32'),
      # 14.27 GF on the fp32 pipe = 90.7 us > 30 MB
      'b': dict(calls=1, total_ms=0.15, avg_ms=0.15, flops=14.27e9, bytes=30e6, pipe='f32'),
      # the same flops on bf16x6: 14.27 / 416.7 = 34.2 us
      'c': dict(calls=1, total_ms=0.09, avg_ms=0.09, flops=14.27e9, bytes=30e6, pipe='bf16x6'),
      # bf16x3: 35.2 GF / 833 TF/s = 42.3 us < 351 MB / 6.3 TB/s = 55.7 us
      'd': dict(calls=1, total_ms=0.13, avg_ms=0.13, flops=35.2e9, bytes=351e6, pipe='bf16x3'),
  }
  r = bench.step_roofline(kern, 0.5)
  want = 2 * 100e6 / 6.3e12 * 1e3 + 14.27e9 / 157.3e12 * 1e3 + 14.27e9 / (2500e12 / 6) * 1e3 + 351e6 / 6.3e12 * 1e3
  assert abs(r['floor_ms'] - want) < 1e-3, (r['floor_ms'], want)
  assert abs(r['frac'] - want / 0.5) < 1e-3
  assert r['kernels']['a']['bound'] == 'hbm' and r['kernels']['b']['bound'] == 'mfma' and r['kernels']['d']['bound'] == 'hbm'


def test_regions_with_calls_of_different_sizes_are_priced_per_call():
  """R2D2 runs each torso layer on 40 x 256 and on 81 x 256 frames under ONE region name; the r4 profiler kept
  the LAST call's flops and the median duration over ALL calls (cfg5's conv fraction came out 1.34x too high).  The
  aggregation now prices each group of equally sized calls at its own median and reports sum(flops) / sum(time)."""
  from seed_rl_amd import ops
  small, big = (54.3e9, 100e6), (110.0e9, 200e6)            # (flops, bytes) of the 40- and 81-step calls
  durations = {'conv': [0.40, 0.41, 0.80, 0.82, 9.9]}       # ms; the last one caught a page fault
  costs = {'conv': [small, small, big, big, big]}
  r = ops.aggregate(durations, costs, {'conv': 'bf16x6'})['conv']
  assert r['calls'] == 5 and r['pipe'] == 'bf16x6'
  want_ms = 2 * 0.405 + 3 * 0.82                           # medians per group: 0.405 and 0.82 (the outlier does not move it)
  assert abs(r['total_ms'] - want_ms) < 1e-9
  assert abs(r['flops_total'] - (2 * small[0] + 3 * big[0])) < 1 and abs(r['bytes_total'] - (2 * small[1] + 3 * big[1])) < 1
  # the rate every consumer computes, flops / avg_ms, is sum(flops) / sum(time)
  assert abs(r['flops'] / r['avg_ms'] - r['flops_total'] / r['total_ms']) < 1e-3
  last_call_over_global_median = big[0] / 0.80             # what r4 reported
  assert r['flops'] / r['avg_ms'] < 0.99 * last_call_over_global_median
  # and the whole-step floor sums the groups' own floors (here: all MFMA-bound on bf16x6)
  bench = importlib.import_module('bench')
  fl = bench.step_roofline({'conv': r}, 5.0)
  want = (2 * small[0] + 3 * big[0]) / (2500e12 / 6) * 1e3
  assert abs(fl['floor_ms'] - want) < 1e-3, (fl['floor_ms'], want)
  # a group can be HBM-bound while another of the same name is MFMA-bound: each takes its own maximum
  mixed = ops.aggregate({'k': [1.0, 1.0]}, {'k': [(1e9, 630e6), (157.3e9, 1e6)]})['k']
  fl = bench.step_roofline({'k': mixed}, 5.0)
  assert abs(fl['floor_ms'] - (0.1 + 1.0)) < 1e-3, fl['floor_ms']


def test_relu_mask_is_one_bit_per_element_in_the_algorithmic_bytes():
  """a data gradient's ReLU mask counts 1 bit per element in the floor whatever tensor the kernel reads."""
  import inspect
  from seed_rl_amd import ops
  src = inspect.getsource(ops.conv2d_bwd_data)
  assert 'elems // 8 if relu_mask is not None' in src and '4 * elems if add is not None' in src


def test_kernel_source_digest_tracks_the_sources(tmp_path, monkeypatch):
  from seed_rl_amd import build
  d0 = build.csrc_digest()
  assert len(d0) == 64 and d0 == build.csrc_digest()          # stable
  # a changed byte in any kernel source changes the digest (checked on a copy of the directory)
  import shutil
  src = os.path.join(ROOT, 'seed_rl_amd', 'csrc')
  dst = tmp_path / 'pkg' / 'csrc'
  shutil.copytree(src, dst, ignore=shutil.ignore_patterns('serve'))
  monkeypatch.setattr(build, '_HERE', str(tmp_path / 'pkg'))
  d1 = build.csrc_digest()
  with open(dst / 'adam.hip', 'a') as f:
    f.write('// touched\n')
  assert build.csrc_digest() != d1
